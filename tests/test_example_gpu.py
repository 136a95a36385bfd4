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


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_cpp_host_resident_chain_and_multi_target_icp(tmp_path):
    """examples/refine_objects.cc: score -> top-k hand-off -> ONE multi-target ICP launch for three objects, all through the
    C ABI with device pointers (HIP runtime API only), equal to per-object host-pointer calls bit for bit."""
    exe = str(tmp_path / "refine_objects")
    lib = os.path.join(ROOT, "physimglobalpose_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "refine_objects.cc"), "-L", lib, "-lpgp",
                        "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, "1024"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
    print(out.stdout)


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_cpp_host_six_objects_over_the_device_group(tmp_path):
    """examples/six_objects.cc: BASELINE configs[3] from a C++ host -- six objects in one native group (eight logical
    members on this one GPU), the flat (object, hypothesis) space scored with one exchange, the best poses of every object
    refined with the poses sharded; equal to single-context calls bit for bit (the program checks)."""
    exe = str(tmp_path / "six_objects")
    lib = os.path.join(ROOT, "physimglobalpose_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "six_objects.cc"), "-L", lib, "-lpgp", f"-Wl,-rpath,{lib}",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, "65536", "32"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PGP_MULTI_EMULATE="8"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
    print(out.stdout)


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_cpp_host_streams_lists_through_the_device_group(tmp_path):
    """examples/stream_batches.cc: the streaming form of the device group (pgp_multi_upload_slot / _enqueue_slot / _collect) from a
    C++ host -- four logical members on this one GPU, then one member: streamed == synchronous == one context (the program checks)."""
    exe = str(tmp_path / "stream_batches")
    lib = os.path.join(ROOT, "physimglobalpose_amd")
    r = subprocess.run(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "stream_batches.cc"), "-L", lib, "-lpgp", f"-Wl,-rpath,{lib}",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for env in (dict(os.environ, PGP_MULTI_EMULATE="4"), {k: v for k, v in os.environ.items() if k != "PGP_MULTI_EMULATE"}):
        out = subprocess.run([exe, "1500", "5"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.strip().endswith("OK") and "streamed == synchronous == one context: yes" in out.stdout
        print(out.stdout)
