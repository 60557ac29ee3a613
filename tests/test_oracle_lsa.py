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


def test_lapjv_cost_limit_semantics(orc):
    """ByteTrack's lap.lapjv(cost, extend_cost=True, cost_limit=L): brute force over all partial matchings on small problems
    (objective = sum of matched costs + L/2 per unmatched row and column), and scipy on the documented embedding."""
    import itertools
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(4)
    for nr, nc in [(1, 1), (2, 3), (3, 2), (4, 4), (5, 3), (3, 6)]:
        for L in (0.3, 0.8, 1.7):
            c = rng.uniform(0, 1.2, (nr, nc))
            x, y = orc.lapjv_limit(c, L)
            best, best_pairs = None, None
            for k in range(min(nr, nc) + 1):
                for rows in itertools.combinations(range(nr), k):
                    for cols in itertools.permutations(range(nc), k):
                        v = sum(c[r, cc] for r, cc in zip(rows, cols)) + (nr - k + nc - k) * L / 2
                        if best is None or v < best - 1e-12:
                            best, best_pairs = v, set(zip(rows, cols))
            got = {(i, int(x[i])) for i in range(nr) if x[i] >= 0}
            assert got == best_pairs, (nr, nc, L)
            assert all(y[j] == i for i, j in got) and sum(y >= 0) == len(got)
            assert all(c[i, j] < L for i, j in got)
    c = rng.uniform(0, 1, (40, 55))
    x, y = orc.lapjv_limit(c, 0.25)
    n = 95
    ext = np.full((n, n), 0.125); ext[40:, 55:] = 0; ext[:40, :55] = c
    r, cc = linear_sum_assignment(ext)
    assert {(i, j) for i, j in zip(r, cc) if i < 40 and j < 55} == {(i, int(x[i])) for i in range(40) if x[i] >= 0}
    x0, y0 = orc.lapjv_limit(np.zeros((0, 5)), 0.5)
    assert len(x0) == 0 and (y0 == -1).all()
