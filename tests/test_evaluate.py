"""CPU: tracklab_amd.evaluate -- MOT files -> HOTA + CLEAR-MOT / ID measures, per sequence and combined."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import REPO


def _write(path, frames):
    with open(path, "w") as f:
        for fr, rows in enumerate(frames, start=1):
            for tid, (l, t, w, h) in rows:
                f.write(f"{fr},{tid},{l},{t},{w},{h},1.0,-1,-1,-1\n")


def test_evaluate_folders_perfect_shifted_and_combined(tmp_path):
    from tracklab_amd import clearmot, evaluate, hota, mot_io
    gt_dir, pr_dir = tmp_path / "gt", tmp_path / "pred"
    gt_dir.mkdir(); pr_dir.mkdir()
    rng = np.random.default_rng(0)
    seqs = {}
    for name, nobj, nfr in (("a", 4, 20), ("b", 6, 15)):
        base = rng.uniform(100, 800, (nobj, 2))
        frames = [[(i + 1, (base[i, 0] + 5 * f, base[i, 1] + 2 * f, 60.0, 140.0)) for i in range(nobj)] for f in range(nfr)]
        seqs[name] = frames
        _write(gt_dir / f"{name}.txt", frames)
    _write(pr_dir / "a.txt", seqs["a"])                                        # perfect
    worse = [[(tid + 10 if (f >= 8 and tid == 1) else tid, (l + 12.0, t, w, h)) for tid, (l, t, w, h) in rows[:-1]]      # shifted, one id switch, one miss
             for f, rows in enumerate(seqs["b"])]
    _write(pr_dir / "b.txt", worse)
    res = evaluate.evaluate_folders(str(gt_dir), str(pr_dir))
    a, b, c = res["sequences"]["a"], res["sequences"]["b"], res["combined"]
    assert abs(a["HOTA"] - 1) < 1e-9 and a["MOTA"] == 1.0 and a["IDF1"] == 1.0 and a["num_switches"] == 0
    assert 0.3 < b["HOTA"] < 1 and b["num_misses"] == 15 and b["num_switches"] == 1 and b["num_false_positives"] == 0 and b["IDF1"] < 1
    assert b["HOTA"] < c["HOTA"] < a["HOTA"] and c["num_frames"] == 35 and c["num_objects"] == 4 * 20 + 6 * 15
    # the combination is the one the two metric modules define on summed statistics
    rb = evaluate.evaluate_sequence(mot_io.load_mot(str(gt_dir / "b.txt")), mot_io.load_mot(str(pr_dir / "b.txt")))
    ra = evaluate.evaluate_sequence(mot_io.load_mot(str(gt_dir / "a.txt")), mot_io.load_mot(str(pr_dir / "a.txt")))
    assert abs(hota.finalize(ra["hota"] + rb["hota"])["summary"]["HOTA"] - c["HOTA"]) < 1e-12
    assert clearmot.merge([ra["clear"], rb["clear"]])["mota"] == c["MOTA"]
    out = subprocess.run([sys.executable, "-m", "tracklab_amd.evaluate", str(gt_dir), str(pr_dir)], capture_output=True, text=True, cwd=REPO,
                         env=dict(os.environ, PYTHONPATH=REPO))
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout)["combined"]["MOTA"] == c["MOTA"]


def test_evaluate_sequence_with_empty_prediction_and_gaps():
    from tracklab_amd import evaluate
    gt = {"frame": np.array([1, 1, 3]), "track_id": np.array([1, 2, 1]), "ltwh": np.array([[0, 0, 10, 10], [50, 50, 10, 10], [1, 0, 10, 10.0]])}
    none = {"frame": np.zeros(0, np.int64), "track_id": np.zeros(0, np.int64), "ltwh": np.zeros((0, 4))}
    r = evaluate.combine({"s": evaluate.evaluate_sequence(gt, none)})["combined"]
    assert r["num_frames"] == 3 and r["num_misses"] == 3 and r["MOTA"] == 0.0 and r["HOTA"] == 0.0
