"""tracklab_amd.hota vs the TrackEval copy vendored by the reference (golden: tests/golden/hota_cases.npz)."""
import os

import numpy as np

from conftest import GOLDEN
from tracklab_amd import hota

FIELDS = ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP")


def _seq(g, si):
    n = int(g[f"s{si}_n_frames"])
    gt = [(g[f"s{si}_f{f}_gt_ids"], g[f"s{si}_f{f}_gt_boxes"]) for f in range(n)]
    tr = [(g[f"s{si}_f{f}_tr_ids"], g[f"s{si}_f{f}_tr_boxes"]) for f in range(n)]
    return hota.hota_sequence(*hota.sequence_from_rows(gt, tr))


def test_hota_matches_vendored_trackeval():
    g = np.load(os.path.join(GOLDEN, "hota_cases.npz"))
    packs = []
    for si in range(2):
        res = _seq(g, si)
        packs.append(hota.pack(res))
        fin = hota.finalize(packs[-1])
        for k in FIELDS:
            np.testing.assert_allclose(fin[k], g[f"s{si}_{k}"], rtol=1e-12, atol=1e-12, err_msg=f"seq {si} {k}")
    comb = hota.finalize(packs[0] + packs[1])          # combine_sequences == SUM of the packed statistics
    for k in FIELDS:
        np.testing.assert_allclose(comb[k], g[f"comb_{k}"], rtol=1e-12, atol=1e-12, err_msg=f"combined {k}")


def test_hota_edge_cases():
    e = np.zeros(0, dtype=int)
    r = hota.hota_sequence([e, e], [e, e], [np.zeros((0, 0))] * 2)
    assert r["HOTA_TP"].sum() == 0
    r = hota.hota_sequence([np.array([0, 1])], [e], [np.zeros((2, 0))])
    assert (r["HOTA_FN"] == 2).all()
    b = np.array([[0, 0, 10, 10.0], [20, 20, 30, 30]])
    gi, ti, s = hota.sequence_from_rows([(np.array([5, 9]), b)] * 3, [(np.array([2, 7]), b)] * 3)
    fin = hota.finalize(hota.pack(hota.hota_sequence(gi, ti, s)))
    assert abs(fin["summary"]["HOTA"] - 1.0) < 1e-12
