#!/usr/bin/env python3
"""Runs every kernel family of the library a few times at representative sizes, for
`rocprofv3 --kernel-trace --stats -- python3 tools/profile_rows.py` (summary committed under
profiles/) and for the PMC passes of tools/collect_rows_pmc.sh.  Sizes follow bench.py's other_rows."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth  # noqa: E402

REPS = int(os.environ.get("PGP_PROFILE_REPS", "5"))
rng = np.random.default_rng(0)
w = synth.make_workload(50000, 5000, 4096, config_id=2)
sc = LcpScorer(0)
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)

# ICP: 64 poses, 2500-point segment vs 5000-point model, trimmed (UCTState form), 10 iterations
seg = w.Q_xyz[rng.choice(len(w.Q_xyz), 2500, replace=False)]
R = synth._rot_axis_angle([0.2, 0.5, -0.4], 0.8)
S = (seg @ R.T + np.array([0.1, 0.0, 0.7])).astype(np.float32)
Tinv = np.linalg.inv(synth._se3(R, np.array([0.1, 0.0, 0.7])))
G = np.stack([synth.colmajor16(Tinv @ synth._se3(synth._random_rot(rng, np.deg2rad(5)), 0.005 * rng.standard_normal(3)))
              for _ in range(64)])
for _ in range(REPS):
    sc.icp_refine(S, w.Q_xyz, G, trim=0.9, max_iterations=10)
# point-to-plane and the capped / grid form on scene-sized clouds (SceneCfg.cpp:135-141)
sc.icp_refine_ex(S, w.Q_xyz, G[:8], tgt_nrm=w.Q_nrm, max_iterations=10, energy_ratio=0.0, error_metric=1)
big_src = (w.P_xyz[rng.choice(len(w.P_xyz), 30000, replace=False)] + 0.002).astype(np.float32)
for _ in range(REPS):
    sc.icp_refine_ex(big_src, w.P_xyz, synth.colmajor16(np.eye(4))[None], max_iterations=8, max_corr_dist=0.01,
                     energy_ratio=0.0, transformation_epsilon=1e-9, nn_search=2)

# congruent sets: single base (pair extraction + quads), then the batched drop-in path
w2 = synth.make_workload(4000, 2000, 4, config_id=3, n_search=1000)
sc2 = LcpScorer(0)
sc2.set_search_model(w2.Qs_xyz)
T = w2.T_gt.reshape(4, 4).T
ids = rng.choice(1000, 4, replace=False)
base = (w2.Qs_xyz[ids] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
d1 = float(np.linalg.norm(base[0] - base[1]))
d6 = float(np.linalg.norm(base[2] - base[3]))
for _ in range(REPS):
    p1 = sc2.extract_pairs(d1, w.delta, cap=1 << 20)
    p6 = sc2.extract_pairs(d6, w.delta, cap=1 << 20)
    sc2.find_congruent(base, 0.4, 0.6, w.delta, p1, p6, cap=1 << 20)
from _dropin import make_dropin_case  # noqa: E402
with tempfile.TemporaryDirectory() as d:
    _, case = make_dropin_case(d)
cw, table = case["w"], case["table"]
keys = np.array(list(table.keys()), np.int32)
counts = np.array([len(table[tuple(k)]) for k in keys.tolist()], np.int32)
pairs = np.concatenate([np.array(table[tuple(k)], np.int32).reshape(-1, 2) for k in keys.tolist()])
sc3 = LcpScorer(0)
sc3.init(cw.P_xyz, cw.P_nrm, cw.P_w, cw.Q_xyz, cw.Q_nrm, cw.delta)
sc3.set_search_model(cw.Qs_xyz)
sc3.set_ppf_map(keys, counts, pairs)
for _ in range(REPS):
    bids, binv, st = sc3.select_bases(rng.random((128, 4)))
    ok = st == 1
    nq = sc3.find_congruent_batch(bids[ok], cw.P_xyz[bids[ok]], binv[ok], cw.delta)
    picks = np.array([(b, j) for b in range(int(ok.sum())) for j in range(min(int(nq[b]), 100))], np.int32).reshape(-1, 2)
    if len(picks):
        Tf, pose, status, rms = sc3.congruent_batch_fit(picks, bids[ok], cw.centroid_P, cw.centroid_Q)
        sc3.score(Tf[status == 1], PGP_MODE_WEIGHTED)

# rigid fits, clustering, depth cost, back-projection, radius filter
sc.set_search_model(w.Qs_xyz)
b = rng.integers(0, len(w.P_xyz), (10000, 4)).astype(np.int32)
qd = rng.integers(0, len(w.Qs_xyz), (10000, 4)).astype(np.int32)
sw, _, _, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
obs = rng.uniform(0.4, 1.2, (480, 640)).astype(np.float32)
ren = (obs[None] + rng.normal(0, 0.02, (16, 480, 640))).astype(np.float32)
raw = rng.integers(2000, 60000, (480, 640)).astype(np.uint16)
msk = (rng.random((480, 640)) < 0.5).astype(np.uint8)
Kc = np.array([[614.0, 0, 322.5], [0, 614.0, 239.7], [0, 0, 1]], np.float32)
for _ in range(REPS):
    sc.rigid_from_congruent(b, qd, w.centroid_P, w.centroid_Q)
    sc.cluster_poses(w.T, sw + np.float32(1e-6), bs, accept_fraction=0.0)
    sc.depth_cost(obs, ren, 0.01)
    sc.backproject_depth(raw, Kc, msk)
    seg = sc.voxel_grid(w.P_xyz[:20000], 0.01)
    sc.mls_normals(seg, 0.02)
flt = LcpScorer(0)
for _ in range(REPS):
    flt.radius_outlier_filter(cw.P_xyz, cw.P_nrm, 0.03, 10)

# round 3: ICP checker paths, leaf-state rendering + cost in HBM, Verify's early termination, unexplained segment
os.environ["PGP_ICP_NN"] = "scan"
sc.icp_refine(S, w.Q_xyz, G, trim=0.9, max_iterations=10)                 # icp_nn_split + icp_refine<true>
os.environ["PGP_ICP_NN"] = "index"
os.environ["PGP_ICP_PERSIST"] = "0"
sc.icp_refine(S, w.Q_xyz, G, trim=0.9, max_iterations=10)                 # icp_nn_index + icp_refine<true>
os.environ.pop("PGP_ICP_NN")
os.environ.pop("PGP_ICP_PERSIST")
import torch  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_render_gpu import icosphere  # noqa: E402
mv, mf = icosphere(5, 0.1)
cam = sc.camera(Kc, 480, 640, 0.1, 1.0)
Tl = np.stack([synth.colmajor16(synth._se3(synth._random_rot(rng), [rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1),
                                                                   rng.uniform(0.5, 0.9)])) for _ in range(64)])
d_v, d_f, d_Tl = torch.from_numpy(mv).cuda(), torch.from_numpy(mf).cuda(), torch.from_numpy(Tl).cuda()
d_par, d_ob = torch.from_numpy(obs).cuda().clamp(max=0.95), torch.from_numpy(obs).cuda()
d_pts = torch.from_numpy(w.Q_xyz.astype(np.float32)).cuda()
for _ in range(REPS):
    d_img = sc.render_depth_device(d_v, d_f, d_Tl, cam, d_parent=d_par)
    sc.depth_cost_device(d_ob, d_img, 0.01)
    sc.render_depth_device(d_pts, None, d_Tl, cam, d_parent=d_par, d_depth=d_img)
torch.cuda.synchronize()
sc.set_verify_early_out(True)
for _ in range(REPS):
    sc.score(w.T, 0)
sc.set_verify_early_out(False)
models = [w.Q_xyz, w.Q_xyz[:1500]]
poses = np.stack([w.T_gt, w.T[7]])
for _ in range(REPS):
    sc.unexplained_segment(w.P_xyz[:3000], models, poses, 0.008)
# round 3, second half: sparse form of the scene index (forced on the C2 scene), the round-by-round clustering pass
os.environ["PGP_INDEX"] = "sparse"
sp = LcpScorer(0)
sp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
os.environ.pop("PGP_INDEX")
for _ in range(REPS):
    sp.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
    sp.score(w.T, 0)
for _ in range(REPS):
    sc.cluster_poses(w.T, sw, bs)                      # accept_fraction 0.5: few clusters -> cluster_rounds
print("profile_rows done")
