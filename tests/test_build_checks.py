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
