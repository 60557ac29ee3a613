"""LSA oracle: scipy-identical assignments (including tie-breaks) on the golden matrices, the
reference's vendored py-motmetrics known-answer tests, and live scipy on fresh matrices."""
import json
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from conftest import GOLDEN


def test_lsa_motmetrics_known_answers(orc):
    kat = json.load(open(os.path.join(GOLDEN, "lsa_kat.json")))
    for case in kat["cases"]:
        r, c = orc.lsa(np.array(case["cost"], dtype=float))
        assert list(r) == case["rows"] and list(c) == case["cols"]


def test_lsa_golden_cases(orc):
    g = np.load(os.path.join(GOLDEN, "lsa_cases.npz"))
    for i in range(int(g["n_cases"])):
        r, c = orc.lsa(g[f"c{i}_cost"])
        np.testing.assert_array_equal(r, g[f"c{i}_rows"], err_msg=f"case {i}")
        np.testing.assert_array_equal(c, g[f"c{i}_cols"], err_msg=f"case {i}")


def test_lsa_vs_live_scipy(orc):
    rng = np.random.default_rng(123)
    for it in range(300):
        n, m = rng.integers(1, 40, 2)
        kind = it % 4
        c = rng.uniform(0, 1, (n, m))
        if kind == 1:
            c = np.round(c * 3)
        elif kind == 2:
            c[c > 0.4] = 0.4 + 1e-5
        elif kind == 3:
            c = -c
        r, cc = orc.lsa(c)
        er, ec = linear_sum_assignment(c)
        np.testing.assert_array_equal(r, er)
        np.testing.assert_array_equal(cc, ec)


def test_lsa_empty_and_invalid(orc):
    r, c = orc.lsa(np.zeros((0, 5)))
    assert len(r) == 0 and len(c) == 0
    with pytest.raises(ValueError):
        orc.lsa(np.array([[np.nan, 1.0]]))
    with pytest.raises(ValueError):
        orc.lsa(np.array([[np.inf, np.inf]]))
