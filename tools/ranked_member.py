#!/usr/bin/env python3
"""ONE rank of a device group that spans processes (pgp_multi_create_ranked), for tests/test_multi_ranked_emulated_gpu.py: with
PGP_MULTI_EMULATE_RANKED=1 the ranks share one device and exchange through shared memory, so the launcher form's own logic -- slices
by GLOBAL rank, every process holding all transforms, every process taking the arg-max over the complete vector -- runs with
several ranks on a 1-GPU box.  Every rank checks what it gets against a single context of its own and prints one line.
usage: python tools/ranked_member.py rank world id_hex [device]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from physimglobalpose_amd import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED, synth  # noqa: E402


def same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def main():
    rank, world, uid = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3])
    dev = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    w = synth.make_workload(20000, 2000, 4 * 500, config_id=47)
    lists = [w.T[k * 500:(k + 1) * 500] for k in range(4)] + [w.T[:world - 1 if world > 1 else 1], w.T[:0]]
    one = LcpScorer(dev)
    one.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    grp = MultiGpuScorer.ranked([dev], rank, world, uid)
    inf = grp.info()
    assert inf["world"] == world and inf["rank0"] == rank and inf["n_local"] == 1, inf
    grp.init(w.P_xyz, w.P_nrm, w.P_w, w.Q_xyz, w.Q_nrm, w.delta)
    checks = 0
    for mode in (PGP_MODE_PLAIN, PGP_MODE_WEIGHTED):
        want = [one.score(T, mode, w.gate_deg) for T in lists]
        for k, T in enumerate(lists):                      # the synchronous call: every rank gets the complete arrays
            assert same(grp.score(T, mode, w.gate_deg), want[k]), ("sync", mode, k)
            checks += 1
        for k, T in enumerate(lists):
            grp.upload_slot(k, T)
        for order in ([0], [1, 2], [3, 2, 1, 0, 4], [2, 5], [5, 1], [0, 1, 2, 3, 0, 1]):   # the streaming form
            for s in order:
                grp.enqueue_slot(s, mode, w.gate_deg)
            assert same(grp.collect(), want[order[-1]]), ("stream", mode, order)
            checks += 1
    # the running-best list over the COMPLETE vector, decided on every process (exact records on member 0's context)
    one.set_exact_records(True)
    grp.set_exact_records(True)
    assert same(grp.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg), one.score(w.T, PGP_MODE_WEIGHTED, w.gate_deg))
    # calls that gather without a collective refuse a group that spans processes
    try:
        grp.icp_refine([(w.Q_xyz[:100], w.Q_xyz, np.eye(4, dtype=np.float32).T.reshape(1, 16))])
        refused = world == 1
    except Exception as e:
        refused = "spans several processes" in str(e)
    assert refused
    ex = grp.info()["exchanges"]
    grp.close()
    one.close()
    print(f"RANK_OK {rank} of {world}: {checks} comparisons, {ex} exchanges, emulated {inf['emulated']}", flush=True)


if __name__ == "__main__":
    main()
