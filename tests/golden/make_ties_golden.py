#!/usr/bin/env python3
"""tests/golden/lattice_ties.npz: exact distance ties on a lattice scene, answered by the REFERENCE-backed harness
(oracle/_ref/libpgp_ref.so: the reference's own kd-tree, kdtree.h, compiled from /root/reference -- build container
only: `make -C oracle ref && python tests/golden/make_ties_golden.py`).  Inputs + the harness's outputs, no source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _checkers import Ref  # noqa: E402
from test_exact_ties_gpu import _lattice_case  # noqa: E402

P, Pn, Pw, Q, Qn, delta, T = _lattice_case(21)
ref = Ref(P, Pn, Pw, Q, Qn)
nH, nQ = len(T), len(Q)
hits = np.zeros((nH, nQ), np.int32)
counts = np.zeros(nH, np.int32)
scores = np.zeros(nH, np.float32)
wscores = np.zeros(nH, np.float32)
reg_flat, reg_off = [], [0]
for h in range(nH):
    s, g, hit = ref.verify(T[h], delta)
    scores[h], counts[h], hits[h] = s, g, hit
    ws, reg = ref.weighted_verify(T[h], delta)
    wscores[h] = ws
    reg_flat.append(reg)
    reg_off.append(reg_off[-1] + len(reg))
np.savez_compressed(os.path.join(HERE, "lattice_ties.npz"), P=P, Pn=ref.normals(0), Pw=Pw, Q=Q, Qn=ref.normals(1), T=T,
                    delta=np.float32(delta), hits=hits, counts=counts, scores=scores, wscores=wscores,
                    reg_flat=np.concatenate(reg_flat).astype(np.int32), reg_off=np.array(reg_off, np.int32))
print("lattice_ties.npz:", len(P), "scene points,", nQ, "model points,", nH, "transforms; registered", reg_off[-1])
