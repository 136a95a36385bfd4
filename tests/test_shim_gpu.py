"""The C++-linkage drop-in (shim/libsuper4pcs.so exporting the reference's own
getProbableTransformsSuper4PCS symbol) driven through the file hand-off the node uses: three PLY
files as pcl::io::savePLYFile writes them, a 16-bit probability PNG, a PPFMap.txt, camera
intrinsics.  shim/test_shim is prebuilt in the build container (it needs Eigen headers, which the
GPU box does not have) and travels with the repository snapshot."""
import os
import subprocess

import numpy as np
import pytest

from physimglobalpose_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "shim", "test_shim")


from _dropin import make_dropin_case, ppf_map, write_ply  # noqa: F401


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
@pytest.mark.parametrize("binary", [False, True])
def test_drop_in_symbol_recovers_the_pose(tmp_path, binary):
    argv, case = make_dropin_case(tmp_path, binary)
    w, P, Qv, Qs, table = case["w"], case["P"], case["Qv"], case["Qs"], case["table"]
    seg, val, search, png, ppf, fx, fy, cx, cy = argv
    env = dict(os.environ, PGP_SHIM_SEED="12345", PGP_SHIM_VERBOSE="1")
    out = subprocess.run([BIN, seg, val, search, png, ppf, str(fx), str(fy), str(cx), str(cy)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # the in-memory overload (no PLY/PNG round trip) gives the same answer for the same seed
    out_mem = subprocess.run([BIN, seg, val, search, png, ppf, str(fx), str(fy), str(cx), str(cy)],
                             env=dict(env, SHIM_TEST_INMEMORY="1"), capture_output=True, text=True, timeout=600)
    assert out_mem.returncode == 0, out_mem.stderr[-2000:]
    def strip(t):   # wall-clock line aside, the two hand-offs print the same result
        return [l for l in t.splitlines() if not l.startswith("ELAPSED_MS")]
    assert strip(out_mem.stdout) == strip(out.stdout)
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l and l[0].isupper()}
    assert int(lines["PPFMAP"][0]) == len(table)
    score = float(lines["BEST_SCORE"][0])
    pose = np.array(lines["BEST_POSE"], float).reshape(4, 4)
    hyp_scores = [float(x) for x in lines["HYPOTHESES"][1:]]
    assert int(lines["HYPOTHESES"][0]) == len(hyp_scores) >= 1
    assert all(b > a for a, b in zip(hyp_scores, hyp_scores[1:]))          # running-best subsequence
    assert abs(hyp_scores[-1] - score) < 1e-6
    assert int(lines["REGISTERED"][0]) > 0
    # the recovered pose is close to the ground truth: rotation < 5 deg, translation < 1 cm
    G = w.T_gt_world
    dR = pose[:3, :3] @ G[:3, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
    # reference score of the ground-truth pose on the same (re-centred) clouds
    from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED
    Pc, Qsc, Qvc, cP, cQ = LcpScorer.center(P.astype(np.float32), Qs.astype(np.float32), Qv.astype(np.float32))
    sc = LcpScorer()
    sc.init(Pc, w.P_nrm, w.P_w, Qvc, w.Q_nrm, w.delta)
    A, B = np.eye(4), np.eye(4)
    A[:3, 3], B[:3, 3] = -cP, cQ
    s_gt = sc.score(synth.colmajor16(A @ G @ B)[None], PGP_MODE_WEIGHTED)[0][0]
    s_found = sc.score(synth.colmajor16(A @ pose @ B)[None], PGP_MODE_WEIGHTED)[0][0]
    print(out.stderr[-300:], "score", score, "gt", s_gt, "rot err", ang, "trans err", pose[:3, 3] - G[:3, 3])
    assert abs(s_found - score) < 1e-4          # the returned pose really has the returned score
    assert score > 0.6 * s_gt
    assert ang < 10.0 and np.linalg.norm(pose[:3, 3] - G[:3, 3]) < 0.02, (ang, pose[:3, 3] - G[:3, 3], score)


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
def test_drop_in_reads_the_probability_image_in_the_encoding_of_cv_imwrite(tmp_path):
    """The probability image as the node writes it (dense, Sub filter on every row, byte runs only, several IDAT chunks;
    also with noise in every pixel, which leaves the decoder nothing but literals) holds the same values as the sparse
    PIL-encoded one at every pixel a segment point falls on: the file hand-off returns the same poses and scores."""
    outs = {}
    for kind in ("pil", "opencv", "opencv_noisy"):
        d = tmp_path / kind
        d.mkdir()
        argv, _ = make_dropin_case(d, png=kind)
        env = dict(os.environ, PGP_SHIM_SEED="2024", SHIM_TEST_REPEAT="2")
        r = subprocess.run([BIN, *argv], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[kind] = [l for l in r.stdout.splitlines() if not l.startswith("ELAPSED_MS")]
        assert any(l.startswith("BEST_SCORE") and float(l.split()[1]) > 0 for l in outs[kind])
    assert outs["opencv"] == outs["pil"]
    assert outs["opencv_noisy"] == outs["pil"]


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
@pytest.mark.parametrize("n_objects", [1, 3, 10])
def test_objects_of_one_frame_side_by_side_equal_the_single_calls(tmp_path, n_objects):
    """getProbableTransformsSuper4PCSFrame (the node's object loop, SceneCfg.cpp:379-402, as ONE call: every object on a
    thread and a context of its own, device work overlapping): every job of every frame returns what the single call
    returns -- best pose and score, the list's scores, the registered points, bit for bit -- for one object (the caller's
    thread), three, and ten (more jobs than worker threads); the serial form of the same entry point too."""
    argv, _ = make_dropin_case(tmp_path)
    for extra in ({}, {"PGP_SHIM_FRAME_SERIAL": "1"}):
        env = dict(os.environ, PGP_SHIM_SEED="777", PGP_SHIM_PRIVATE_RAND="1", SHIM_TEST_FRAME=str(n_objects), SHIM_TEST_REPEAT="6", **extra)
        r = subprocess.run([BIN, *argv], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l and l[0].isupper()}
        assert lines["FRAME_SAME"] == [str(6 * n_objects), "of", str(6 * n_objects)], r.stdout
        assert float(lines["BEST_SCORE"][0]) > 0


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
@pytest.mark.parametrize("extra", [{"SHIM_TEST_FRAME_THREADS": "1"}, {"SHIM_TEST_FRAME_THREADS": "files"},
                                   {"SHIM_TEST_FRAME_THREADS": "1", "SHIM_TEST_FRAME_ONE_OBJECT": "1"},
                                   {"SHIM_TEST_FRAME_ONE_OBJECT": "1"}])
def test_single_calls_from_fresh_threads_find_their_objects(tmp_path, extra):
    """The drop-in's state belongs to the process: a fresh std::thread per object around the single call -- the form the
    reference's authors left commented out at SceneCfg.cpp:377,402-403 -- finds each object's context and table from the
    frame before (no 13 ms re-upload per call) and returns the single call's result; calls for ONE object from several
    threads (or several jobs of one frame) take turns on its context, same result."""
    argv, _ = make_dropin_case(tmp_path)
    env = dict(os.environ, PGP_SHIM_SEED="4321", PGP_SHIM_PRIVATE_RAND="1", SHIM_TEST_FRAME="4", SHIM_TEST_REPEAT="8", **extra)
    r = subprocess.run([BIN, *argv], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l and l[0].isupper()}
    assert lines["FRAME_SAME"] == ["32", "of", "32"], r.stdout
    ms = [float(x) for x in lines["FRAME_MS"]]
    assert np.median(ms[2:]) < 8.0, ms          # four objects per frame: nowhere near a table upload (13 ms) per call


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
def test_drop_in_takes_the_reference_tie_rule_for_a_segment_with_duplicated_points(tmp_path):
    """A segment that holds duplicated points makes exact distance ties an every-query event; the drop-in notices
    (a hash of the coordinates) and switches pgp_set_exact_ties on for that object: same output as forcing it, and a
    verbose line says so.  A duplicate-free segment keeps the default."""
    argv, case = make_dropin_case(tmp_path, False)
    seg, val, search, png, ppf, fx, fy, cx, cy = argv
    # rewrite the segment with a tenth of its points twice (pcl::io::savePLYFile layout: write_ply)
    P, w = case["P"], case["w"]
    rng = np.random.default_rng(5)
    extra = rng.choice(len(P), len(P) // 10, replace=False)
    Pd = np.concatenate([P, P[extra]])
    Nd = np.concatenate([w.P_nrm, synth._unit(rng.standard_normal((len(extra), 3))).astype(np.float32)])
    seg_dup = os.path.join(str(tmp_path), "segment_dup.ply")
    write_ply(seg_dup, Pd, Nd, False)
    base_env = dict(os.environ, PGP_SHIM_SEED="777", PGP_SHIM_VERBOSE="1", SHIM_TEST_REPEAT="1")
    base_env.pop("PGP_SHIM_EXACT_TIES", None)

    def run(segment, **extra_env):
        out = subprocess.run([BIN, segment, val, search, png, ppf, str(fx), str(fy), str(cx), str(cy)],
                             env=dict(base_env, **extra_env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return [l for l in out.stdout.splitlines() if not l.startswith("ELAPSED_MS")], out.stderr

    auto, err_auto = run(seg_dup)
    forced, _ = run(seg_dup, PGP_SHIM_EXACT_TIES="1")
    assert auto == forced and "exact ties: on" in err_auto
    _, err_plain = run(seg)
    assert "exact ties: off" in err_plain


@pytest.mark.skipif(not os.path.exists(BIN), reason="shim/test_shim not built (needs Eigen: make -C shim)")
def test_drop_in_over_a_device_group_returns_the_single_device_result(tmp_path):
    """PGP_SHIM_DEVICES=2 (hypotheses sharded over a native group, pgp_multi_*) with the group's N > 1 branches run on one
    device (PGP_MULTI_EMULATE=2): best pose, score, list and registered count of the single-device call, for the same seed."""
    argv, _ = make_dropin_case(tmp_path)
    cmd = [BIN] + [str(a) for a in argv]
    env = dict(os.environ, PGP_SHIM_SEED="4242", SHIM_TEST_INMEMORY="1", SHIM_TEST_REPEAT="2")
    one = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    two = subprocess.run(cmd, env=dict(env, PGP_SHIM_DEVICES="2", PGP_MULTI_EMULATE="2"), capture_output=True, text=True, timeout=600)
    assert one.returncode == 0 and two.returncode == 0, one.stderr[-1000:] + two.stderr[-1000:]

    def strip(t):
        return [l for l in t.splitlines() if not l.startswith("ELAPSED_MS")]
    assert strip(one.stdout) == strip(two.stdout)
    assert any(l.startswith("BEST_SCORE") and float(l.split()[1]) > 0.1 for l in one.stdout.splitlines())

