"""One-rank RCCL self-check on this box's GPU: torch.distributed with backend nccl (= RCCL on ROCm) is initialised,
the sweep's own message -- (C x C second moments + count) in float64, C = 512: 2 MiB -- goes through all_reduce and
a barrier, the group is destroyed.  With one rank nothing crosses xGMI: what this shows is that the RCCL library of the
image initialises a communicator on the hardware and runs the collective the path uses; the N-rank behaviour is covered
by the 2-rank gloo tests and measured by the driver's --gpus N runs.  Prints one JSON line."""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist


def main():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    c = 512
    packed = torch.arange(c * c + 1, dtype=torch.float64, device='cuda')
    want = packed.clone()
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)       # communicator creation happens here
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = 50
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(n):
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    dist.barrier()
    ok = bool(torch.equal(packed, want))
    out = dict(ok=ok, backend=dist.get_backend(), world_size=dist.get_world_size(), message_bytes=packed.numel() * 8,
               init_and_first_allreduce_s=round(t1 - t0, 3), allreduce_us=round((t3 - t2) / n * 1e6, 1),
               nccl_version='.'.join(str(v) for v in torch.cuda.nccl.version()),
               note='one rank: the communicator and the collective run on the GPU, nothing crosses xGMI')
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    sys.exit(main())
