"""The ICP call of bench.py's config2_object row (64 best poses of 16 384 scored hypotheses, scene segment vs the
5000-point model, 30 iterations): host-pointer call, device-pointer call (HIP events), and -- with the diagnostic
library (PGP_LIB=tools/ab/libpgp_icpstamps.so) -- the phases of the slowest pose."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from physimglobalpose_amd import LcpScorer, PGP_MODE_WEIGHTED, synth

def inv16(T16):
    return synth.colmajor16(np.linalg.inv(np.asarray(T16, np.float64).reshape(4, 4).T))

w = synth.make_workload(50000, 5000, 16384, config_id=210)
sc = LcpScorer()
sc.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
s, c, bi, bs = sc.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg)
top = np.argsort(-s, kind="stable")[:64]
seg = np.ascontiguousarray(w.P_xyz[w.P_w == 1.0])
G = np.stack([inv16(w.T[h]) for h in top])
stamps = "icpstamps" in os.environ.get("PGP_LIB", "")
if stamps:
    os.environ["PGP_ICP_DBG_POSE"] = "0"
T, e, it = sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
print(f"segment {len(seg)} points, iterations: total {it.sum()} min {it.min()} max {it.max()}  histogram {np.bincount(it)[1:]}")
if stamps:
    e = e.astype(np.float64); ni = it[0]
    us = e[1:6] / 100.0 / ni; dbg = e[8:16]
    print(f"pose 0, {ni} iterations: per iteration nn {us[0]:.1f} select {us[1]:.1f} sums {us[2]:.1f} solve {us[3]:.1f} stop {us[4]:.1f} us; bounds {dbg[5]/100/ni:.1f} sort {dbg[6]/100/ni:.1f} search {dbg[7]/100/ni:.1f}; unresolved per iteration {dbg[3]/ni:.0f}")
    whole = e[16:] / 100.0
    print(f"whole time in the kernel of poses 16..: min {whole.min():.0f} mean {whole.mean():.0f} max {whole.max():.0f} us; their iterations {it[16:]}")
else:
    for reps in (3, 20):
        t0 = time.perf_counter()
        for _ in range(reps): sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
        dt = (time.perf_counter() - t0) / reps
        print(f"host-pointer call: {dt*1e3:.3f} ms ({reps} reps)")
    d_src = torch.zeros(len(seg), 4, device="cuda"); d_src[:, :3] = torch.from_numpy(seg).cuda()
    d_tgt = torch.zeros(len(w.Q_xyz), 4, device="cuda"); d_tgt[:, :3] = torch.from_numpy(w.Q_xyz).cuda()
    d_G = torch.from_numpy(G).cuda()
    d_T = d_G.clone()
    d_it = torch.zeros(64, dtype=torch.int32, device="cuda")
    for label, tok in (("device call, index rebuilt every call", None), ("device call, resident index (token)", 77)):
        for _ in range(2):
            d_T.copy_(d_G); sc.icp_refine_device(d_src, d_tgt, d_T, None, d_it, trim=0.9, max_iterations=30, target_token=tok)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        tot = 0.0
        for _ in range(20):
            d_T.copy_(d_G)
            ev[0].record(); sc.icp_refine_device(d_src, d_tgt, d_T, None, d_it, trim=0.9, max_iterations=30, target_token=tok); ev[1].record()
            torch.cuda.synchronize(); tot += ev[0].elapsed_time(ev[1])
        print(f"{label}: {tot/20:.3f} ms (events), same transforms as the host call: {np.array_equal(d_T.cpu().numpy(), T)}")
    for n in (8, 16, 32, 64):
        t0 = time.perf_counter()
        for _ in range(10): sc.icp_refine(seg, w.Q_xyz, G[:n], trim=0.9, max_iterations=30)
        print(f"host-pointer call, {n:3d} poses: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
    for wgs in ("1", "2", "4"):
        os.environ["PGP_ICP_WGS"] = wgs
        sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
        t0 = time.perf_counter()
        for _ in range(10): sc.icp_refine(seg, w.Q_xyz, G, trim=0.9, max_iterations=30)
        print(f"host-pointer call, 64 poses, PGP_ICP_WGS={wgs}: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
