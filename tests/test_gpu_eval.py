"""-m gpu: HOTA on the device (tlk_hota_sequence_f64) against the reference-made TrackEval fixtures (tests/golden/hota_cases.npz) and, on a
600-frame 100-object stream tracked by the oracle, against the numpy restatement that those fixtures pin (tracklab_amd/hota.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FIELDS = ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP")


def test_gpu_hota_matches_vendored_trackeval():
    from tracklab_amd import hota
    g = np.load(os.path.join(GOLDEN, "hota_cases.npz"))
    packs = []
    for si in range(2):
        n = int(g[f"s{si}_n_frames"])
        gt = [(g[f"s{si}_f{f}_gt_ids"], g[f"s{si}_f{f}_gt_boxes"]) for f in range(n)]
        tr = [(g[f"s{si}_f{f}_tr_ids"], g[f"s{si}_f{f}_tr_boxes"]) for f in range(n)]
        packs.append(hota.pack(hota.hota_sequence_gpu(gt, tr)))
        fin = hota.finalize(packs[-1])
        for k in FIELDS:
            np.testing.assert_allclose(fin[k], g[f"s{si}_{k}"], rtol=1e-12, atol=1e-12, err_msg=f"seq {si} {k}")
    comb = hota.finalize(packs[0] + packs[1])
    for k in FIELDS:
        np.testing.assert_allclose(comb[k], g[f"comb_{k}"], rtol=1e-12, atol=1e-12, err_msg=f"combined {k}")


def test_gpu_hota_on_a_full_stream_equals_the_numpy_restatement(orc):
    from tracklab_amd import hota
    from tracklab_amd.synth import SyntheticStream
    hyper = dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115, iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False)
    trk = orc.OCSort(**hyper)
    gt, tr = [], []
    for fr in SyntheticStream(3, 100, 600, miss_prob=0.08):
        out = orc.ocsort_wrapper_step(trk, fr["dets"], 0.4)
        gt.append((fr["gt_all_ids"], fr["gt_boxes"]))
        tr.append((out[:, 4].astype(int), out[:, :4]))
    cpu = hota.hota_sequence(*hota.sequence_from_rows(gt, tr))
    gpu = hota.hota_sequence_gpu(gt, tr)
    for k in ("HOTA_TP", "HOTA_FN", "HOTA_FP"):
        np.testing.assert_array_equal(gpu[k], cpu[k])
    for k in ("LocA_sum", "AssA", "AssRe", "AssPr"):
        np.testing.assert_allclose(gpu[k], cpu[k], rtol=1e-12, atol=1e-12)
    assert 0.5 < hota.finalize(hota.pack(gpu))["summary"]["HOTA"] < 1.0


def test_gpu_hota_edge_cases():
    from tracklab_amd import hota
    e, z = np.zeros(0, dtype=int), np.zeros((0, 4))
    assert hota.hota_sequence_gpu([(e, z)] * 2, [(e, z)] * 2)["HOTA_TP"].sum() == 0
    b = np.array([[0, 0, 10, 10.0], [20, 20, 30, 30]])
    assert (hota.hota_sequence_gpu([(np.array([0, 1]), b)], [(e, z)])["HOTA_FN"] == 2).all()
    assert (hota.hota_sequence_gpu([(e, z)], [(np.array([4, 6]), b)])["HOTA_FP"] == 2).all()
    gt = [(np.array([5, 9]), b), (e, z), (np.array([5, 9]), b)]
    tr = [(np.array([2, 7]), b), (np.array([2]), b[:1]), (np.array([2, 7]), b)]        # a frame without ground truth: its tracker box is a false positive
    r, c = hota.hota_sequence_gpu(gt, tr), hota.hota_sequence(*hota.sequence_from_rows(gt, tr))
    assert (r["HOTA_TP"] == 4).all() and (r["HOTA_FP"] == 1).all()
    for k in r:
        np.testing.assert_allclose(r[k], c[k], rtol=1e-14, atol=1e-14)


def _clear_frames(g, s):
    og, oh = g[f"s{s}_offsets_gt"], g[f"s{s}_offsets_hyp"]
    return [(g[f"s{s}_gt_ids"][og[f]:og[f + 1]], g[f"s{s}_gt_ltwh"][og[f]:og[f + 1]], g[f"s{s}_hyp_ids"][oh[f]:oh[f + 1]], g[f"s{s}_hyp_ltwh"][oh[f]:oh[f + 1]])
            for f in range(len(og) - 1)]


def test_gpu_clearmot_matches_vendored_motmetrics():
    """tlk_clear_sequence_f64 on the four sequences of the py-motmetrics fixture: every measure per sequence and OVERALL as the vendored
    motmetrics computed them (<= 1e-12), and count for count equal to the host accumulator."""
    from conftest import GOLDEN
    from tracklab_amd import clearmot
    g = np.load(os.path.join(GOLDEN, "clearmot.npz"))
    names, rows = [str(n) for n in g["metric_names"]], [str(r) for r in g["row_names"]]
    counts = []
    for s in range(len(rows) - 1):
        frames = _clear_frames(g, s)
        c = clearmot.sequence_counts_gpu(frames, max_iou=0.5)
        acc = clearmot.MOTAccumulator()
        for fr in frames:
            acc.update_boxes(*fr, max_iou=0.5)
        ref = acc.counts()
        for k in clearmot.SUM_FIELDS:
            assert c[k] == ref[k] or abs(c[k] - ref[k]) <= 1e-12 * max(1.0, abs(ref[k])), (rows[s], k, c[k], ref[k])
        assert c["sum_distance"] == ref["sum_distance"]                         # same additions in the same order
        m = clearmot.finalize(c)
        for k, exp in zip(names, g["summary"][s]):
            np.testing.assert_allclose(m[k], exp, rtol=1e-12, atol=1e-12, err_msg=f"{rows[s]} {k}")
        counts.append(c)
    tot = clearmot.merge(counts)
    for k, exp in zip(names, g["summary"][-1]):
        np.testing.assert_allclose(tot[k], exp, rtol=1e-12, atol=1e-12, err_msg=f"OVERALL {k}")


def test_gpu_clearmot_on_a_tracked_stream_and_edge_cases(orc):
    """A 100-object stream tracked by the C oracle (id switches, misses, late births) and the degenerate sequences."""
    from tracklab_amd import clearmot
    from tracklab_amd.synth import SyntheticStream
    hyper = dict(asso_func="iou", delta_t=3, det_thresh=0.3, inertia=0.2, iou_threshold=0.3, max_age=5, min_hits=1, use_byte=False)
    trk = orc.OCSort(**hyper)
    frames = []
    for fr in SyntheticStream(17, 100, 120, miss_prob=0.15, churn_period=30):
        rows = orc.ocsort_wrapper_step(trk, fr["dets"], 0.4)
        gt = fr["gt_boxes"]
        g_ltwh = np.column_stack([gt[:, 0], gt[:, 1], gt[:, 2] - gt[:, 0], gt[:, 3] - gt[:, 1]])
        h_ltwh = np.column_stack([rows[:, 0], rows[:, 1], rows[:, 2] - rows[:, 0], rows[:, 3] - rows[:, 1]]) if len(rows) else np.zeros((0, 4))
        frames.append((fr["gt_all_ids"], g_ltwh, rows[:, 4].astype(np.int64) if len(rows) else np.zeros(0, np.int64), h_ltwh))
    c = clearmot.sequence_counts_gpu(frames)
    acc = clearmot.MOTAccumulator()
    for fr in frames:
        acc.update_boxes(*fr, max_iou=0.5)
    ref = acc.counts()
    assert ref["num_switches"] > 0 and ref["num_misses"] > 100 and ref["num_fragmentations"] > 0
    for k in clearmot.SUM_FIELDS:
        assert c[k] == ref[k], (k, c[k], ref[k])
    # degenerate sequences
    assert clearmot.sequence_counts_gpu([])["num_frames"] == 0
    e = clearmot.sequence_counts_gpu([([], np.zeros((0, 4)), [], np.zeros((0, 4))), ([1, 2], np.array([[0., 0, 10, 10], [50, 50, 10, 10]]), [], np.zeros((0, 4))),
                                      ([], np.zeros((0, 4)), [7], np.array([[0., 0, 10, 10]])), ([1], np.array([[0., 0, 10, 10]]), [7], np.array([[100., 100, 10, 10]]))])
    assert (e["num_frames"], e["num_misses"], e["num_false_positives"], e["num_matches"]) == (4, 3, 2, 0)
    assert clearmot.finalize(e)["idf1"] == 0.0


def test_evaluate_folders_on_the_device_equals_the_host_path(tmp_path):
    """python -m tracklab_amd.evaluate --gpu: MOT files -> HOTA + CLEAR-MOT / ID measures with both evaluators on the device, per sequence and
    combined, equal to the host path (counts exactly, HOTA's sums to 1e-12)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    from test_evaluate import _write
    from tracklab_amd import evaluate
    gt_dir, pr_dir = tmp_path / "gt", tmp_path / "pred"
    gt_dir.mkdir(); pr_dir.mkdir()
    rng = np.random.default_rng(1)
    for name, nobj, nfr in (("a", 12, 40), ("b", 30, 25)):
        base = rng.uniform(100, 1500, (nobj, 2))
        frames = [[(i + 1, (base[i, 0] + 4 * f, base[i, 1] + 2 * f, 60.0, 140.0)) for i in range(nobj)] for f in range(nfr)]
        _write(gt_dir / f"{name}.txt", frames)
        pred = [[(tid + (100 if (f >= nfr // 2 and tid % 5 == 0) else 0), (l + rng.normal(0, 6), t + rng.normal(0, 6), w, h)) for tid, (l, t, w, h) in rows if rng.random() > 0.1]
                for f, rows in enumerate(frames)]
        _write(pr_dir / f"{name}.txt", pred)
    cpu = evaluate.evaluate_folders(str(gt_dir), str(pr_dir))
    gpu = evaluate.evaluate_folders(str(gt_dir), str(pr_dir), device="gpu")
    assert cpu["combined"]["num_switches"] > 0 and cpu["combined"]["num_misses"] > 0
    for scope in ("a", "b"):
        for k, v in cpu["sequences"][scope].items():
            assert abs(gpu["sequences"][scope][k] - v) <= 1e-12 * max(1.0, abs(v)) or (np.isnan(v) and np.isnan(gpu["sequences"][scope][k])), (scope, k)
    for k, v in cpu["combined"].items():
        assert abs(gpu["combined"][k] - v) <= 1e-12 * max(1.0, abs(v)), k
    out = subprocess.run([sys.executable, "-m", "tracklab_amd.evaluate", str(gt_dir), str(pr_dir), "--gpu"], capture_output=True, text=True, cwd=REPO,
                         env=dict(os.environ, PYTHONPATH=REPO))
    assert out.returncode == 0, out.stderr[-2000:]
    assert abs(json.loads(out.stdout)["combined"]["HOTA"] - cpu["combined"]["HOTA"]) < 1e-12


def test_evaluator_plugin_on_the_device_equals_its_host_path():
    """HipTrackEvalEvaluator (the plugin in the place of tracklab.wrappers.TrackEvalEvaluator) with cfg.device gpu (its default) and cpu on the same
    tracker state: every count equal, HOTA's sums to 1e-12."""
    from types import SimpleNamespace as NS

    from test_eval_plugin import _state
    from tracklab_amd.wrappers import HipTrackEvalEvaluator
    st = _state(5)
    gpu = HipTrackEvalEvaluator(NS(device="gpu"), "val", False, "unused", None).run(st)
    cpu = HipTrackEvalEvaluator(NS(device="cpu"), "val", False, "unused", None).run(st)
    assert set(gpu["sequences"]) == set(cpu["sequences"])
    for name, m in cpu["sequences"].items():
        for k, v in m.items():
            assert gpu["sequences"][name][k] == pytest.approx(v, rel=1e-12, abs=1e-12, nan_ok=True), (name, k)      # (ratios of the video without boxes are NaN on both sides)
    for k, v in cpu["combined"].items():
        assert gpu["combined"][k] == pytest.approx(v, rel=1e-12, abs=1e-12, nan_ok=True), k
