"""Clustered ICP launches (several workgroups per pose, cooperative) from several host threads and contexts of one
process at once: the library chains them per device (csrc/icp.hip: CoopChain), so none is ever half resident while
another spins -- every call returns the bits of the single-threaded run, none reports a lost meeting."""
import threading

import numpy as np
import pytest

from physimglobalpose_amd import LcpScorer
from test_icp_index_gpu import _problem, FORMS

pytestmark = pytest.mark.gpu


def test_three_threads_three_contexts_clustered_icp():
    probs = [_problem(40 + k, 3000, 1800, 24, rot_deg=6.0, trans=0.006, outliers=0.05) for k in range(3)]
    scs = [LcpScorer() for _ in range(3)]
    ref = [sc.icp_refine_ex(S, M, G, **FORMS["trimmed"]) for sc, (S, M, N, G) in zip(scs, probs)]
    out = [[None] * 6 for _ in range(3)]
    err = []

    def work(k):
        try:
            S, M, N, G = probs[k]
            for rep in range(6):
                out[k][rep] = scs[k].icp_refine_ex(S, M, G, **FORMS["trimmed"])
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for k in range(3):
        for rep in range(6):
            for x, y in zip(ref[k], out[k][rep]):
                assert np.array_equal(x, y), (k, rep)
            assert (out[k][rep][2] >= 1).all()
