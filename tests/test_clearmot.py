"""CPU: tracklab_amd.clearmot (CLEAR-MOT + ID measures from in-memory tables) against the py-motmetrics copy the reference vendors
(tests/golden/make_golden.py gen_clearmot: four synthetic sequences, per sequence and OVERALL)."""
import os

import numpy as np

from conftest import GOLDEN


def _replay(g, s, clearmot):
    acc = clearmot.MOTAccumulator()
    og, oh = g[f"s{s}_offsets_gt"], g[f"s{s}_offsets_hyp"]
    for f in range(len(og) - 1):
        gi, gb = g[f"s{s}_gt_ids"][og[f]:og[f + 1]], g[f"s{s}_gt_ltwh"][og[f]:og[f + 1]]
        hi, hb = g[f"s{s}_hyp_ids"][oh[f]:oh[f + 1]], g[f"s{s}_hyp_ltwh"][oh[f]:oh[f + 1]]
        if f == 10:
            np.testing.assert_allclose(clearmot.iou_distance_matrix(gb, hb, max_iou=0.5), g[f"s{s}_dist_f10"], rtol=0, atol=1e-15, equal_nan=True)
        acc.update_boxes(gi, gb, hi, hb, max_iou=0.5)
    return acc


def test_clearmot_matches_vendored_motmetrics_per_sequence_and_overall():
    from tracklab_amd import clearmot
    g = np.load(os.path.join(GOLDEN, "clearmot.npz"))
    names, rows = [str(n) for n in g["metric_names"]], [str(r) for r in g["row_names"]]
    accs = [_replay(g, s, clearmot) for s in range(len(rows) - 1)]
    for s, acc in enumerate(accs):
        m = acc.metrics()
        for k, exp in zip(names, g["summary"][s]):
            np.testing.assert_allclose(m[k], exp, rtol=1e-12, atol=1e-12, err_msg=f"{rows[s]} {k}")
    assert rows[-1] == "OVERALL"
    tot = clearmot.merge([a.counts() for a in accs])
    for k, exp in zip(names, g["summary"][-1]):
        np.testing.assert_allclose(tot[k], exp, rtol=1e-12, atol=1e-12, err_msg=f"OVERALL {k}")
    # the SUM all-reduce form: pack -> add -> unpack -> finalize gives the same numbers
    v = sum(clearmot.pack(a.counts()) for a in accs)
    red = clearmot.finalize(clearmot.unpack(v))
    assert red["mota"] == tot["mota"] and red["idf1"] == tot["idf1"] and red["num_switches"] == tot["num_switches"]


def test_clearmot_edge_cases():
    from tracklab_amd import clearmot
    acc = clearmot.MOTAccumulator()
    acc.update([], [], np.empty((0, 0)))                      # empty frame still counts as a frame
    acc.update([1, 2], [], np.empty((2, 0)))                  # only ground truth: two misses
    acc.update([], [7], np.empty((0, 1)))                     # only hypotheses: one false positive
    acc.update([1], [7], [[np.nan]])                          # do-not-pair: miss + false positive
    m = acc.metrics()
    assert (m["num_frames"], m["num_misses"], m["num_false_positives"], m["num_matches"]) == (4, 3, 2, 0)
    assert m["mota"] == 1.0 - 5 / 3 and np.isnan(m["motp"]) and m["idf1"] == 0.0
    assert clearmot.iou_distance_matrix(np.zeros((0, 4)), np.zeros((3, 4))).shape == (0, 0)
