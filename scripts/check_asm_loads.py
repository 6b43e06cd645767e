"""Static check of the kernels that read LDS from inline assembly (ds_read2_b64 with register outputs): the compiler
takes an asm's outputs for ready, so between such a read and the s_waitcnt that covers it nothing may touch the
destination registers.  Reads hipcc's assembly (-S) of a source file and checks every `ds_read2*` inside ASMSTART /
ASMEND against the instructions up to the wait that retires it (in-order LDS returns: the wait lgkmcnt(n) retires a read
once at most n LDS instructions were issued after it; scalar memory loads, which return out of order, must not occur
in between at all).

    python scripts/check_asm_loads.py rewriting_amd/csrc/rw_upwino.hip [more.hip]      (exit 1 on a violation)
"""
import re
import subprocess
import sys
import tempfile


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def check(path):
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
        flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-S', '--cuda-device-only']
        first = open(path).readline()
        if first.startswith('// hipcc-flags:'):
            flags += first.split(':', 1)[1].split()
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + [path, '-o', f.name], stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split('\n')
    bad = 0
    n_reads = 0
    in_asm = False
    for i, l in enumerate(lines):
        t = l.strip()
        if 'ASMSTART' in t:
            in_asm = True
            continue
        if 'ASMEND' in t:
            in_asm = False
            continue
        if not (in_asm and t.startswith('ds_read2')):
            continue
        n_reads += 1
        dst = regs(t.split()[1].rstrip(','))
        younger = 0
        for j in range(i + 1, len(lines)):
            u = lines[j].strip()
            if u.startswith('.LBB') or u.startswith('.Lfunc_end'):
                print('%s:%d: the read reaches a label before its wait' % (path, i + 1))
                bad += 1
                break
            if not u or u.startswith(';') or u.startswith('.'):
                continue
            m = re.search(r'lgkmcnt\((\d+)\)', u)
            if u.startswith('s_waitcnt') and m and int(m.group(1)) <= younger:
                break
            if u.startswith(('s_load', 's_buffer_load')):
                print('%s:%d: scalar load `%s` between the read at line %d and its wait' % (path, j + 1, u, i + 1))
                bad += 1
                break
            if u.startswith('ds_'):
                younger += 1
            toks = re.findall(r'v\[\d+:\d+\]|v\d+', u)
            hit = [x for x in toks if regs(x) & dst]
            if hit:
                print('%s:%d: `%s` touches %s before the wait of the read at line %d' % (path, j + 1, u, hit[0], i + 1))
                bad += 1
                break
    print('%s: %d asm LDS reads checked, %d violations' % (path, n_reads, bad))
    return bad


if __name__ == '__main__':
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
