"""Two PROCESSES on one GPU, each refining 64 poses (the clustered, cooperative ICP launch) 200 times: no call may stall
(a lost meeting would cost its 2 s clock bound and a retry) and every call returns the bits of the first.
Measured: 200 calls in 0.67 s per process, slowest call 14 ms."""
import sys, os, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
if len(sys.argv) > 1:
    import numpy as np
    from physimglobalpose_amd import LcpScorer
    from test_icp_index_gpu import _problem, FORMS
    S, M, N, G = _problem(50 + int(sys.argv[1]), 5000, 2500, 64, rot_deg=5.0, trans=0.005)
    sc = LcpScorer()
    ref = sc.icp_refine_ex(S, M, G, **FORMS["trimmed"])
    t0 = time.time(); worst = 0
    for _ in range(200):
        t1 = time.time()
        out = sc.icp_refine_ex(S, M, G, **FORMS["trimmed"])
        worst = max(worst, time.time() - t1)
        assert all(np.array_equal(x, y) for x, y in zip(ref, out))
    print(f"proc {sys.argv[1]}: 200 calls {time.time()-t0:.2f} s, slowest call {worst*1e3:.1f} ms", flush=True)
else:
    ps = [subprocess.Popen([sys.executable, __file__, str(k)]) for k in range(2)]
    print([p.wait() for p in ps])
