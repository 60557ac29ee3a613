"""tracklab_amd.trackeval_metrics: TrackEval's CLEAR / Identity restated (pip trackeval is absent: parity unpinned against it).  Pinned by
  * brute force: every frame's kept assignment is an optimum of TrackEval's score matrix, the Identity assignment is the optimum over all id maps;
  * agreement with the py-motmetrics restatement (pinned against the reference's vendored copy) where the two definitions coincide;
  * a hand-made sequence where they do NOT coincide (the documented difference);
  * the result layout tracklab's process_trackeval_results hooks read."""
import itertools

import numpy as np
import pytest

from tracklab_amd import clearmot, evaluate, hota
from tracklab_amd import trackeval_metrics as te


def _stream(seed, n_obj=6, n_frames=30, miss=0.15, swap=0.05, noise=3.0):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, 600, (n_obj, 2)); vel = rng.uniform(-4, 4, (n_obj, 2))
    ids = np.arange(1, n_obj + 1)
    tid = ids.copy() + 100
    gt_fr, pr_fr = [], []
    for f in range(n_frames):
        pos = pos + vel
        box = np.column_stack([pos, pos + [40, 90]])
        keep = rng.random(n_obj) > miss
        if rng.random() < swap and n_obj > 1:
            i, j = rng.choice(n_obj, 2, replace=False); tid[i], tid[j] = tid[j], tid[i]
        pb = box[keep] + rng.normal(0, noise, (int(keep.sum()), 4))
        extra = rng.random() < 0.2
        pi = tid[keep]
        if extra:
            pb = np.vstack([pb, rng.uniform(0, 600, (1, 4)) + [0, 0, 700, 700]]); pi = np.append(pi, 999)
        gt_fr.append((ids.copy(), box)); pr_fr.append((pi.copy(), pb))
    return gt_fr, pr_fr


def test_every_frame_assignment_is_an_optimum_and_identity_is_the_global_optimum():
    gt_fr, pr_fr = _stream(1, n_obj=4, n_frames=12)
    g, t, sims = hota.sequence_from_rows(gt_fr, pr_fr)
    res = te.identity_eval_sequence(g, t, sims)
    ng, nt = 4, int(max(x.max() for x in t if len(x))) + 1
    pot = np.zeros((ng, nt)); gc = np.zeros(ng); tc = np.zeros(nt)
    for gi, ti, s in zip(g, t, sims):
        a, b = np.nonzero(s >= 0.5); pot[gi[a], ti[b]] += 1; gc[gi] += 1; tc[ti] += 1
    best = None
    for k in range(0, min(ng, nt) + 1):                       # brute force over every partial one-to-one id map
        for rows in itertools.combinations(range(ng), k):
            for cols in itertools.permutations(range(nt), k):
                tp = sum(pot[r, c] for r, c in zip(rows, cols))
                cost = (gc.sum() - tp) + (tc.sum() - tp)
                best = cost if best is None else min(best, cost)
    assert res["IDFN"] + res["IDFP"] == best and res["IDTP"] == gc.sum() - res["IDFN"]


@pytest.mark.parametrize("seed", range(6))
def test_agrees_with_motmetrics_where_the_definitions_coincide(seed):
    """no id swaps inside the tracker and well separated objects: continuity never conflicts with the optimum -> same TP / FP / FN / IDSW / MOTP"""
    gt_fr, pr_fr = _stream(seed, swap=0.0, noise=1.0)
    g, t, sims = hota.sequence_from_rows(gt_fr, pr_fr)
    c = te.clear_final(te.clear_eval_sequence(g, t, sims))
    acc = clearmot.MOTAccumulator()
    to_ltwh = lambda b: np.column_stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]]).reshape(-1, 4)      # noqa: E731
    for (gi, gb), (pi, pb) in zip(gt_fr, pr_fr):
        acc.update_boxes(gi, to_ltwh(gb), pi, to_ltwh(pb), max_iou=0.5)
    m = clearmot.finalize(acc.counts())
    assert c["CLR_FP"] == m["num_false_positives"] and c["CLR_FN"] == m["num_misses"] and c["IDSW"] == m["num_switches"]
    assert c["MOTA"] == pytest.approx(m["mota"], abs=1e-12)
    i = te.identity_final(te.identity_eval_sequence(g, t, sims))
    assert (i["IDTP"], i["IDFP"], i["IDFN"]) == (m["idtp"], m["idfp"], m["idfn"]) and i["IDF1"] == pytest.approx(m["idf1"], abs=1e-12)


def test_id_switch_is_scored_against_the_last_id_ever_matched():
    """gt 0 tracked by A, lost for a frame, then tracked by B: TrackEval counts the switch (prev_tracker_id survives the gap)"""
    box = np.array([[0., 0., 10., 10.]])
    far = np.array([[500., 500., 510., 510.]])
    g = [np.array([0])] * 4
    t = [np.array([0]), np.array([0]), np.array([1]), np.array([1])]
    sims = [hota.box_iou_matrix(box, box), hota.box_iou_matrix(box, far), hota.box_iou_matrix(box, box), hota.box_iou_matrix(box, box)]
    r = te.clear_eval_sequence(g, t, sims)
    assert (r["CLR_TP"], r["CLR_FN"], r["CLR_FP"], r["IDSW"], r["Frag"]) == (3, 1, 1, 1, 1)
    assert (r["MT"], r["PT"], r["ML"]) == (0, 1, 0)            # 3 of 4 frames = 0.75: not > 0.8


def test_mt_threshold_is_strict():
    box = np.array([[0., 0., 10., 10.]])
    far = np.array([[500., 500., 510., 510.]])
    g = [np.array([0])] * 5
    t = [np.array([0])] * 5
    sims = [hota.box_iou_matrix(box, box)] * 4 + [hota.box_iou_matrix(box, far)]
    r = te.clear_eval_sequence(g, t, sims)
    assert (r["MT"], r["PT"]) == (0, 1)                        # exactly 0.8: mostly tracked for py-motmetrics (>=), not for TrackEval (>)


def test_degenerate_sequences():
    e = np.zeros(0, np.int64)
    r = te.clear_eval_sequence([np.array([0, 1])], [e], [np.zeros((2, 0))])
    assert (r["CLR_FN"], r["ML"], r["CLR_TP"]) == (2, 2, 0)
    r = te.clear_eval_sequence([e], [np.array([0])], [np.zeros((0, 1))])
    assert (r["CLR_FP"], r["CLR_FN"]) == (1, 0)
    assert te.identity_eval_sequence([e], [np.array([0])], [np.zeros((0, 1))]) == {"IDTP": 0, "IDFN": 0, "IDFP": 1}


def test_layout_is_what_process_trackeval_results_reads():
    seqs = {}
    for name, seed in (("seqA", 3), ("seqB", 4)):
        gt_fr, pr_fr = _stream(seed)
        g, t, sims = hota.sequence_from_rows(gt_fr, pr_fr)
        seqs[name] = dict(te.evaluate_sequence_frames(gt_fr, pr_fr), hota=hota.pack(hota.hota_sequence(g, t, sims), frames=len(gt_fr)))
    res = te.trackeval_layout(seqs)
    assert set(res) == {"seqA", "seqB", "COMBINED_SEQ", "SUMMARIES"}
    ped = res["COMBINED_SEQ"]["pedestrian"]
    assert set(ped) == {"HOTA", "CLEAR", "Identity"} and ped["HOTA"]["HOTA"].shape == (19,)
    assert ped["CLEAR"]["CLR_TP"] == sum(res[s]["pedestrian"]["CLEAR"]["CLR_TP"] for s in ("seqA", "seqB"))
    assert ped["CLEAR"]["MOTA"] == pytest.approx((ped["CLEAR"]["CLR_TP"] - ped["CLEAR"]["CLR_FP"] - ped["CLEAR"]["IDSW"]) / (ped["CLEAR"]["CLR_TP"] + ped["CLEAR"]["CLR_FN"]))
    # mot_like/common.py:243-249 parses every summary value with float() / int()
    for fam, fields in res["SUMMARIES"]["pedestrian"].items():
        for k, v in fields.items():
            assert isinstance(v, str) and (float(v) if "." in v else int(float(v))) is not None, (fam, k, v)
    assert float(res["SUMMARIES"]["pedestrian"]["HOTA"]["HOTA"]) == pytest.approx(100 * float(np.mean(ped["HOTA"]["HOTA"])), rel=1e-4)


def test_summaries_carry_trackevals_summary_fields_only():
    """TrackEval's CLEAR.summary_fields are the main float + main integer fields; CLR_F1, FP_per_frame, MOTAL, MOTP_sum and CLR_Frames are
    per-sequence detail and do not reach the SUMMARIES block (which process_trackeval_results forwards to the loggers).  CLR_Frames is set on
    the full path of eval_sequence only: TrackEval's two early returns leave it at 0."""
    gt_fr, pr_fr = _stream(5)
    g, t, sims = hota.sequence_from_rows(gt_fr, pr_fr)
    seqs = {"s": dict(te.evaluate_sequence_frames(gt_fr, pr_fr), hota=hota.pack(hota.hota_sequence(g, t, sims), frames=len(gt_fr)))}
    summ = te.trackeval_layout(seqs)["SUMMARIES"]["pedestrian"]
    assert list(summ["CLEAR"]) == ["MOTA", "MOTP", "MODA", "CLR_Re", "CLR_Pr", "MTR", "PTR", "MLR", "sMOTA", "CLR_TP", "CLR_FN", "CLR_FP", "IDSW", "MT", "PT",
                                   "ML", "Frag"]
    assert list(summ["Identity"]) == ["IDF1", "IDR", "IDP", "IDTP", "IDFN", "IDFP"]
    assert list(summ["HOTA"]) == ["HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "OWTA", "HOTA(0)", "LocA(0)", "HOTALocA(0)"]
    e = np.zeros(0, np.int64)
    assert te.clear_eval_sequence([np.array([0, 1])] * 3, [e] * 3, [np.zeros((2, 0))] * 3)["CLR_Frames"] == 0
    assert te.clear_eval_sequence([e] * 3, [np.array([0])] * 3, [np.zeros((0, 1))] * 3)["CLR_Frames"] == 0
    assert seqs["s"]["CLEAR"]["CLR_Frames"] == len(gt_fr)
