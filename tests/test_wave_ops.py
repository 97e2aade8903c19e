"""DPP wave primitives (ccs_amd/csrc/wave_ops.h) against scalar references, on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_wave_ops_on_device(tmp_path):
    exe = tmp_path / "test_wave_ops"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "ccs_amd", "csrc"),
                           os.path.join(ROOT, "tools", "dpp", "test_wave_ops.hip"), "-o", str(exe)], stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout
