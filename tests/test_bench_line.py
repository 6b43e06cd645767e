"""CPU tests of bench.py's host logic that need no launcher (the line's size and keys: tests/test_bench_launcher.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_bench_names_the_upsampling_kernel_the_launcher_picks(monkeypatch):
    """bench.py attributes a fused upsampling launch to the kernel rw_tconv_blur_f32 dispatches to (csrc/rw_tconv.hip, the
    launcher's automatic choice): persistent for 32 .. 128 input channels, one eight-wave workgroup above, the 32-out-channel
    form only for the range RW_TCONV_N32 names; RW_TCONV_PERSISTENT=2 means the pipelined persistent form."""
    import bench
    for name in ('RW_TCONV_N32', 'RW_TCONV_PERSISTENT', 'RW_TCONV_TY'):
        monkeypatch.delenv(name, raising=False)
    assert [bench.tconv_auto_form(i, o) for i, o in ((512, 512), (512, 256), (256, 128), (128, 64), (64, 32), (16, 16))] == \
        ['16', '16', '16', '0', '0', '16']
    monkeypatch.setenv('RW_TCONV_N32', '64:128')
    assert bench.tconv_auto_form(128, 64) == '32' and bench.tconv_auto_form(64, 32) == '32' and bench.tconv_auto_form(256, 128) == '16'
    assert bench.tconv_auto_form(64, 16) == '0'                     # 32 does not divide the out-channels
    monkeypatch.setenv('RW_TCONV_N32', 'nonsense')
    assert bench.tconv_auto_form(128, 64) == '0'
    monkeypatch.delenv('RW_TCONV_N32')
    monkeypatch.setenv('RW_TCONV_PERSISTENT', '2')
    assert bench.tconv_auto_form(64, 32) == '2' and bench.tconv_auto_form(256, 128) == '16'
