"""The bf16 decode kernels issue v_dot2c_f32_bf16 through inline assembly, where the compiler
inserts no wait states after it.  Compile them to assembly and check that nothing but another
dot2c touches a dot result inside the window the hardware leaves unguarded (3 wait states for a
read, 4 for an overwrite).  Seen without the guard: results that changed from run to run."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_unguarded_dot2c_results(tmp_path):
    src = os.path.join(ROOT, "vllm-splitwise_amd", "csrc", "pa_bf16.hip")
    asm = str(tmp_path / "pa_bf16.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, capture_output=True, timeout=600)
    text = open(asm).read()
    assert text.count("v_dot2c_f32_bf16") > 1000          # the packed path is what was compiled
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dot_hazards.py"), asm],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
