"""examples/score_poses.cc: a C++ host driving the C ABI directly (no Python between the caller and
libpgp.so).  Compiled here with g++ and run on the GPU; the program checks the exact properties
itself (identity scores 1.0 and wins, running best, registered ids) and exits non-zero otherwise."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_cpp_host_example(tmp_path):
    exe = str(tmp_path / "score_poses")
    lib = os.path.join(ROOT, "physimglobalpose_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "score_poses.cc"), "-L", lib, "-lpgp", f"-Wl,-rpath,{lib}",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, "1024"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
    print(out.stdout)
