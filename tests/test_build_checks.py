"""Static checks of the compiled kernels that need no GPU (hipcc cross-compiles gfx950 here)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='no hipcc')
def test_asm_lds_reads_reach_their_consumers_only_through_their_waits():
    """rw_upwino.hip reads its weight words from LDS in inline assembly (ds_read2_b64 of the same address into both
    halves of the MFMA operand).  The compiler takes the outputs of an asm for ready, so the generated code is checked:
    between each such read and the s_waitcnt that retires it nothing touches the destination registers, no scalar load
    sits in between (lgkmcnt counts those too, and they return out of order), and the wait's count is one the LDS's
    in-order returns make sufficient."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'check_asm_loads.py'),
                        os.path.join(ROOT, 'rewriting_amd', 'csrc', 'rw_upwino.hip')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '16 asm LDS reads checked, 0 violations' in r.stdout


SPLIT_PATH_SOURCES = ['rw_common.h', 'rw_bound.hip', 'rw_wino4.hip', 'rw_upwino.hip', 'rw_dconv.hip', 'rw_tconv.hip', 'rw_ops.hip']


def test_no_launch_hands_a_device_scalar_to_the_next_one():
    """Round 4's split-operand path carried max |x| bounds and the packed weights' 2^-eU from launch to launch in 4-byte
    device scalars (memset + filtered system-scope atomics + system-scope loads); a consumer occasionally read a stale one
    (GPUTEST_r04: 0.0415 on the image).  Round 5 removed the mechanism: per-wave slots stored plainly, one reduction
    launch, per-lane vector loads, weight scales by value.  Nothing of the old machinery may creep back into the sources
    of that path."""
    banned = ['__hip_atomic', 'atomicMax', 'atomicAdd', 'atomicCAS', 'hipMemsetAsync', 'hipMemset(']
    for name in SPLIT_PATH_SOURCES:
        text = open(os.path.join(ROOT, 'rewriting_amd', 'csrc', name)).read()
        for word in banned:
            assert word not in text, (name, word)
    header = open(os.path.join(ROOT, 'include', 'rewriting_hip.h')).read()
    assert 'rw_publish_scalar_f32(' not in header


def test_bound_sizes_and_weight_scales_on_the_host():
    """rw_bound_floats and rw_split_weight_scale are host functions of the library (no launch): the buffer a producer
    needs, and the power of two u_scale = 2^(15 - e), max |U| < 2^e, that is passed BY VALUE to the pack kernels and
    (inverted) to every convolution launch."""
    from rewriting_amd import _lib
    lib = _lib.load()
    assert lib.rw_bound_floats(0) == 64 + 2048 + 1
    assert lib.rw_bound_floats(64 * 32 * 1024 * 1024) == 64 + 2048 + 64 * 32 * 1024 + 1
    assert lib.rw_bound_floats(-1) == -1
    for amax, want in [(1.0, 2.0 ** 14), (0.999, 2.0 ** 15), (3.7, 2.0 ** 13), (2.0 ** -3, 2.0 ** 17), (0.0, 1.0),
                       (65504.0, 2.0 ** -1)]:
        su = lib.rw_split_weight_scale(amax)
        assert su == want, (amax, su)
        assert amax * su < 2.0 ** 15
        if amax:
            assert amax * su >= 2.0 ** 14


def test_a_bound_travels_on_its_tensor_and_only_inside_the_unhooked_forward():
    """models._amax_of: the producer's bound is an attribute of the feature-map TENSOR (not a bag key, not an address):
    ignored outside the un-hooked forward, ignored once the tensor was written in place, absent from slices."""
    import torch
    from rewriting_amd.utils.stylegan2 import models
    t = torch.zeros(2, 4, 8, 8)
    bound = torch.ones(64)
    t.rw_amax = (bound, t._version)
    assert models._amax_of(t) is None                       # not inside SeqStyleGAN2.forward
    models._rgb_branch.image_path = True
    try:
        assert models._amax_of(t) is bound
        assert models._amax_of(t[:1]) is None               # a slice is another tensor
        assert models._amax_of(None) is None
        t.add_(1.0)                                         # edited in place (a hook): the bound no longer describes it
        assert models._amax_of(t) is None
    finally:
        models._rgb_branch.image_path = False


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='no hipcc')
def test_side_stream_kernels_are_compiled_without_packed_fp32_fma():
    """rw_ops.hip carries `// hipcc-flags: -fno-slp-vectorize`: its kernels run on the side streams beside the convolutions'
    MFMAs, and to_rgb_kernel's v_pk_fma_f32 came back wrong (low half, lanes 48..63) while rw_tconv.hip's kernel ran on
    another stream (round 5: scripts/interference_repro.py, profiles/r05i / r05l).  The generated code of the RGB branch's
    kernels must not contain a packed fp32 FMA."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    # rw_ops.hip: every streaming kernel (ToRGB, upfirdn2d, equalised linear / modulation, demodulation factors, noise,
    # bias + activation ...) -- whatever the forward puts on the RGB / prefetch streams comes from this file;
    # rw_bound.hip: the bound reduction that follows every split-operand launch
    for name in ('rw_ops.hip', 'rw_bound.hip'):
        src = os.path.join(ROOT, 'rewriting_amd', 'csrc', name)
        first = open(src).readline()
        assert first.startswith('// hipcc-flags:') and '-fno-slp-vectorize' in first, name
        out = os.path.join('/tmp', '%s_check_%d.s' % (name, os.getpid()))
        flags = first[len('// hipcc-flags:'):].split()
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only'] + flags +
                           ['-o', out, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
        os.remove(out)
        if name == 'rw_ops.hip':
            for kernel in ('to_rgb_kernel', 'upfirdn2d_up2k4_kernel', 'to_rgb_scalar_kernel', 'equal_linear_mfma_kernel'):
                assert kernel in text, kernel
        # no packed fp32 arithmetic in ANY kernel of the file (the FMA is what was seen to fail; the others share its pipe)
        for insn in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32'):
            lines = [l for l in text.splitlines() if insn in l and not l.lstrip().startswith(';')]
            assert not lines, (name, insn, len(lines))


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'), reason='no llvm-objdump')
def test_the_shipped_library_contains_no_packed_fp32_instruction():
    """csrc/build.sh compiles every kernel but the solver's (rw_solve.hip) with the device feature -packed-fp32-ops (round 6): beside another wave's MFMAs a
    packed fp32 instruction costs more issue time than the two scalar ones it replaces (the forward is 4 - 5 % faster without
    them, profiles/r06af / r06ag), and a v_pk_fma_f32 with op_sel modifiers can return a wrong low half while another wave
    interleaves MFMAs with memory instructions (profiles/r06_interference_probe.md).  Checked on the code objects of the
    library that ships (rewriting_amd/librewriting_hip.so), not on a recompilation."""
    import tempfile
    from rewriting_amd import _lib
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    assert os.path.isfile(_lib.LIB_PATH), 'build the library first (rewriting_amd/csrc/build.sh)'
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(_lib.LIB_PATH, os.path.join(d, 'lib.so'))
        r = subprocess.run([objdump, '--offloading', 'lib.so'], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        objs = sorted(f for f in os.listdir(d) if 'hipv4-amdgcn' in f)
        assert len(objs) >= 10, objs                     # one code object per source file
        kernels = 0
        for f in objs:
            text = subprocess.run([objdump, '-d', '--mcpu=gfx950', f], cwd=d, capture_output=True, text=True).stdout
            kernels += text.count('v_mfma_f32') > 0
            if 'solve_persistent_kernel' in text:        # rw_solve.hip keeps them ("// hipcc-keep-packed-fp32": its header says why)
                continue
            for insn in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32'):
                assert insn not in text, (f, insn, text.count(insn))
        assert kernels >= 6                              # (the disassembly is real: the MFMA kernels are in it)
