"""The host-side concurrency of the library and the drop-in, WITHOUT a GPU: `make -C shim tsan` builds
shim/test_concurrency.cc -- the per-object slots and their leases (shim/object_slots.h), the frame's kept worker pool
(shim/frame_pool.h), the device group's worker hand-offs and the per-device table (csrc/host_worker.h), the very headers the
products include, against stub contexts -- once under ThreadSanitizer and once under AddressSanitizer + UBSan; both must run
clean.  It stages the round-5 defect (one NEW object brought by two calls at once while every slot is leased: two slots, two
contexts) and proves it gone.  SURVEY section 5 "race detection": the reference itself has a live race (main.cpp:20-39)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "shim"), "tsan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return True


@pytest.mark.parametrize("flavour,env", [("tsan", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}),
                                         ("asan", {"ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "halt_on_error=1"})])
def test_host_side_concurrency_is_clean(built, flavour, env):
    exe = os.path.join(ROOT, "shim", f"test_concurrency_{flavour}")
    r = subprocess.run([exe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ALL OK" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr
    assert "runtime error" not in r.stderr
    for part in ("slots: 32 threads x 10000", "same new object from two calls", "frame pool: 8 callers", "workers: 8 callers",
                 "device table: 32 threads"):
        assert part in r.stdout, part
