"""oracle/preprocess_oracle.py (the restatement of UCTState::performTrICP's pre-filter, UCTState.cpp:142-174)
on a case small enough to check by hand."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import preprocess_oracle as po  # noqa: E402


def test_points_within_the_threshold_of_a_posed_model_are_explained():
    model = np.array([[0, 0, 0], [0.1, 0, 0]], np.float32)
    # pose: translate by (1, 2, 3), column-major
    G = np.eye(4, dtype=np.float32)
    G[:3, 3] = [1, 2, 3]
    T = G.T.reshape(1, 16)
    seg = np.array([[1.0, 2.0, 3.005],      # 5 mm from the first model point: explained
                    [1.1, 2.0, 3.0],        # on the second model point: explained
                    [1.05, 2.0, 3.0],       # 5 cm from both: stays
                    [0.0, 0.0, 0.0]], np.float32)   # at the UNPOSED model: stays
    keep = po.unexplained_segment(seg, [model], T, 0.008)
    assert keep.tolist() == [False, False, True, True]
    assert po.unexplained_segment(seg, [], np.zeros((0, 16)), 0.008).all()
    two = po.unexplained_segment(seg, [model, np.array([[0.05, 0, 0]], np.float32)], np.concatenate([T, T]), 0.008)
    assert two.tolist() == [False, False, False, True]
