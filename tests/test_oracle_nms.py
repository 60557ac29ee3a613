"""CPU: SURVEY 8a row G2 -- `non_max_suppression` of plugins/track/strong_sort/sort/preprocessing.py:6-73 (dead code in the reference): the C
restatement (oracle/src/nms.c) against vectors produced by the function itself (tests/golden/make_golden.py gen_nms)."""
import os

import numpy as np

from conftest import GOLDEN


def nms_cases():
    g = np.load(os.path.join(GOLDEN, "deepsort_nms.npz"))
    for ci in range(int(g["n_cases"])):
        yield g[f"c{ci}_boxes"], float(g[f"c{ci}_thr"]), (g[f"c{ci}_scores"] if f"c{ci}_scores" in g else None), g[f"c{ci}_pick"].tolist()


def test_oracle_nms_matches_the_reference_function(orc):
    n = 0
    for boxes, thr, scores, pick in nms_cases():
        assert orc.deepsort_nms(boxes, thr, scores) == pick
        n += 1
    assert n == 6 and orc.deepsort_nms(np.zeros((0, 4)), 0.5) == []


def test_oracle_nms_properties(orc):
    rng = np.random.default_rng(0)
    boxes = np.concatenate([rng.uniform(0, 500, (80, 2)), rng.uniform(30, 90, (80, 2))], 1)
    scores = rng.permutation(80) / 80.0
    keep = orc.deepsort_nms(boxes, 0.4, scores)
    assert keep[0] == int(np.argmax(scores)) and len(set(keep)) == len(keep)           # best score first, no duplicates
    assert list(scores[keep]) == sorted(scores[keep], reverse=True)                   # picked in descending score order
    assert sorted(orc.deepsort_nms(boxes, 1e9, scores)) == list(range(80))            # nothing overlaps that much: all kept
    assert orc.deepsort_nms(np.tile(boxes[:1], (5, 1)), 0.5, np.arange(5.0)) == [4]    # identical boxes: the best one survives
