"""The drop-in boundary is a C ABI: examples/capi_hover.c uses include/qd.h from plain C (gcc, C99) with nothing but the HIP
runtime -- no Python, no PyTorch in the process -- and flies 4096 drones with the on-device PID cascade."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_the_library_without_python(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    libdir = os.path.join(ROOT, "mujoco-drone_amd")
    assert os.path.exists(os.path.join(libdir, "libqd.so")), "libqd.so is not built"
    exe = str(tmp_path / "capi_hover")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
                           os.path.join(ROOT, "examples", "capi_hover.c"), "-o", exe, "-L", libdir, "-lqd", "-L", os.path.join(rocm, "lib"),
                           "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(rocm, "lib")])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(run.stdout)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "Action dimension mismatch" in run.stdout
