"""CPU restatement (numpy float32, operation for operation) of pgp_unexplained_segment -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  It restates the
pre-filter of UCTState::performTrICP (PPE/hypothesis_verification/mcts/UCTState.cpp:142-174): every placed
object's model is moved to its pose, and a segment point is removed when any such point lies strictly within
pointRemovalThreshold (0.008, UCTState.cpp:9) of it.  PARITY UNPINNED against PCL / FLANN (not vendored in
the reference, not installed here): pcl::transformPointCloud's and FLANN's float evaluation orders are taken
as rows ((r0 x + r1 y) + r2 z) + t and (dx^2 + dy^2) + dz^2; what is pinned is HIP path == this file."""
import numpy as np

F = np.float32


def unexplained_segment(seg, models, poses, radius=0.008):
    seg = np.asarray(seg, F)
    keep = np.ones(len(seg), bool)
    r2 = F(radius) * F(radius)
    for m, G in zip(models, np.asarray(poses, F).reshape(-1, 16)):
        m = np.asarray(m, F)
        px = ((G[0] * m[:, 0] + G[4] * m[:, 1]) + G[8] * m[:, 2]) + G[12]
        py = ((G[1] * m[:, 0] + G[5] * m[:, 1]) + G[9] * m[:, 2]) + G[13]
        pz = ((G[2] * m[:, 0] + G[6] * m[:, 1]) + G[10] * m[:, 2]) + G[14]
        for i0 in range(0, len(seg), 512):
            s = seg[i0:i0 + 512]
            dx = s[:, None, 0] - px[None, :]
            dy = s[:, None, 1] - py[None, :]
            dz = s[:, None, 2] - pz[None, :]
            d2 = (dx * dx + dy * dy) + dz * dz
            keep[i0:i0 + 512] &= ~(d2 < r2).any(axis=1)
    return keep
