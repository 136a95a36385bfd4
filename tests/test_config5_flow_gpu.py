"""BASELINE.json configs[4] in miniature: the scene-level search of HypothesisSelection /
UCTSearch (PPE/hypothesis_verification/mcts/UCTSearch.cpp) with every data-parallel step on the GPU
through the C ABI and the tree itself on the host, as north_star prescribes:

  depth image --pgp_backproject_depth--> per-object segment --pgp_set_scene--> index
  per-object hypothesis list --pgp_score_lcp--> LCP scores (child h-values, UCTSearch.cpp:20-26)
  expansion / LCP roll-out (UCTSearch::LCPPolicy :75-135) on the host
  leaf: render (host stand-in) --pgp_depth_cost--> renderScore (UCTState::computeCost :93-116)

Stated substitutions: Bullet (correctPhysics) is skipped and the OpenGL depth renderer
(depth_sim/renderScene.cpp) is replaced by a host point-splat z-buffer -- neither exists in this
image; the SAME splatter produces the observed image, so the costs are consistent.  The test
asserts that the search ends on the ground-truth combination of hypotheses."""
import math

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer, PGP_MODE_PLAIN, synth

pytestmark = pytest.mark.gpu

ROWS, COLS = 240, 320
K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1]], np.float32)


def splat(points_cam, depth=None):
    """z-buffer of camera-frame points (stand-in for depth_sim): min z per pixel, 0 = empty."""
    if depth is None:
        depth = np.full((ROWS, COLS), np.inf, np.float32)
    z = points_cam[:, 2]
    u = np.round(K[0, 0] * points_cam[:, 0] / z + K[0, 2]).astype(int)
    v = np.round(K[1, 1] * points_cam[:, 1] / z + K[1, 2]).astype(int)
    ok = (z > 0.1) & (u >= 0) & (u < COLS) & (v >= 0) & (v < ROWS)
    np.minimum.at(depth, (v[ok], u[ok]), z[ok].astype(np.float32))
    return depth


def finish(depth):
    d = depth.copy()
    d[~np.isfinite(d)] = 0.0
    return d


def pose(rng, t, max_deg):
    return synth._se3(synth._random_rot(rng, np.deg2rad(max_deg)), t)


def test_scene_level_search_finds_the_ground_truth_combination():
    rng = np.random.default_rng(2026)
    n_obj, n_hyp = 3, 10
    model, _ = synth.make_model(rng, 6000)
    model = model.astype(np.float64)
    centers = [np.array([-0.22, 0.02, 0.85]), np.array([0.05, -0.03, 0.8]), np.array([0.27, 0.04, 0.9])]
    gt = [pose(rng, c, 180) for c in centers]
    cams = [model @ G[:3, :3].T + G[:3, 3] for G in gt]
    observed = np.full((ROWS, COLS), np.inf, np.float32)
    label = np.zeros((ROWS, COLS), np.uint8)
    for k, pc in enumerate(cams):
        before = observed.copy()
        splat(pc, observed)
        label[observed < before] = k + 1
    observed = finish(observed)

    sc = LcpScorer(0)
    hyps, scores, gt_index = [], [], []
    for k in range(n_obj):
        # segment of object k from the image, as the node builds pclSegment
        seg = sc.backproject_depth(observed, K, (label == k + 1).astype(np.uint8))
        assert len(seg) > 500
        # hypotheses: the ground truth hidden among perturbed poses (rotation up to 25 deg, 3 cm)
        H = [gt[k] @ pose(rng, 0.03 * rng.standard_normal(3), 25) for _ in range(n_hyp)]
        j = int(rng.integers(n_hyp))
        H[j] = gt[k] @ pose(rng, 0.0005 * rng.standard_normal(3), 0.3)
        gt_index.append(j)
        hyps.append(H)
        obj = LcpScorer(0)
        obj.set_scene(seg, None, None, 0.005)
        obj.set_model(model.astype(np.float32))
        s, _, bi, _ = obj.score(np.stack([synth.colmajor16(T) for T in H]), PGP_MODE_PLAIN)
        scores.append(s)
        assert bi == j                                   # the LCP scorer already ranks it first

    # ---- host UCT over (object 0 hypothesis, object 1 hypothesis, ...) --------------------------
    class Node:
        def __init__(self, depth, parent):
            self.depth, self.parent, self.children, self.n, self.q = depth, parent, {}, 0, 0.0

    def render_cost(states):
        """renderScore of several complete states in ONE GPU call (batched computeCost)."""
        imgs = []
        for st in states:
            d = None
            for k, h in enumerate(st):
                T = hyps[k][h]
                d = splat(model @ T[:3, :3].T + T[:3, 3], d)
            imgs.append(finish(d))
        score, _ = sc.depth_cost(observed, np.stack(imgs), 0.01)
        return score

    root, best, best_cost, expansions = Node(0, None), None, math.inf, 0
    norm = float((observed > 0).sum()) * 2
    for it in range(40):
        node, state = root, []
        while node.depth < n_obj and len(node.children) == n_hyp:      # selection (UCB1)
            h = max(node.children, key=lambda c: node.children[c].q / node.children[c].n +
                    1.0 * math.sqrt(math.log(node.n) / node.children[c].n))
            node, state = node.children[h], state + [h]
        if node.depth < n_obj:                                           # expansion by LCP order
            order = np.argsort(-scores[node.depth], kind="stable")
            h = int(next(c for c in order if c not in node.children))
            node.children[h] = Node(node.depth + 1, node)
            node, state = node.children[h], state + [h]
            expansions += 1
        rollout = state + [int(np.argmax(scores[k])) for k in range(len(state), n_obj)]   # LCPPolicy
        cost = float(render_cost([rollout])[0])
        if cost < best_cost:
            best, best_cost = rollout, cost
        reward = 1.0 - cost / norm
        while node:                                                      # backupReward (:61-67)
            node.n, node.q, node = node.n + 1, node.q + reward, node.parent
    assert best == gt_index, (best, gt_index, best_cost)
    # the batched leaf cost ranks the ground-truth state below single-object swaps
    alts = [list(gt_index) for _ in range(n_obj)]
    for k in range(n_obj):
        alts[k][k] = (gt_index[k] + 1) % n_hyp
    costs = render_cost([gt_index] + alts)
    assert costs[0] < costs[1:].min()
    assert expansions > 0
