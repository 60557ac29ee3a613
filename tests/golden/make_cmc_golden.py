#!/usr/bin/env python3
"""Fixture generator for the camera-motion estimators -- runs ONLY where OpenCV (cv2) is installed (it is not in the build container,
which is why oracle/src/cmc.c and tlk_cmc.hip are "parity unpinned"). Writes tests/golden/cmc_opencv.npz:

  * two seeded synthetic frames related by a known similarity (the generator of tests/test_oracle_cmc.py),
  * every intermediate of GMC.applySparseOptFlow (plugins/track/bot_sort/gmc.py:239-303) computed by OpenCV itself: grey image,
    downscaled image, corners of goodFeaturesToTrack, points / status of calcOpticalFlowPyrLK, matrix / inliers of
    estimateAffinePartial2D, and -- when /root/reference is importable -- the matrix GMC(method="sparseOptFlow").apply returns,
  * a frame pair with its 0.1-scaled grey images and cv2.findTransformECC's warp (full run and after 1 / 2 / 5 iterations): Track.ECC.

tests/test_oracle_cmc.py::test_oracle_against_opencv_fixture (skipped while the file is absent) then pins oracle/src/cmc.c stage by
stage; the GPU tests pin tlk_cmc.hip on the oracle. Usage: python tests/golden/make_cmc_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import cv2
    from test_oracle_cmc import _textured, _warp
    H_, W_ = 360, 640
    rng = np.random.default_rng(11)
    base = _textured(rng, H_ + 80, W_ + 80)
    f0 = np.clip(np.rint(base[40:40 + H_, 40:40 + W_]), 0, 255).astype(np.uint8)
    a, b, tx, ty = 1.01 * np.cos(-0.006), 1.01 * np.sin(-0.006), 3.0, 8.0
    f1 = _warp(base[40:, 40:], a, b, tx, ty, H_, W_)
    out = {"f0": f0, "f1": f1, "truth": np.array([[a, -b, tx], [b, a, ty]]), "cv2_version": np.array(cv2.__version__)}
    feature_params = dict(maxCorners=1000, qualityLevel=0.01, minDistance=1, blockSize=3, useHarrisDetector=False, k=0.04)
    small = []
    for k, f in enumerate((f0, f1)):
        g = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)
        s = cv2.resize(g, (W_ // 2, H_ // 2))
        out[f"gray{k}"], out[f"small{k}"] = g, s
        out[f"eig{k}"] = cv2.cornerMinEigenVal(s, 3, ksize=3)
        out[f"corners{k}"] = cv2.goodFeaturesToTrack(s, mask=None, **feature_params).reshape(-1, 2)
        small.append(s)
    nxt, st, err = cv2.calcOpticalFlowPyrLK(small[0], small[1], out["corners0"].reshape(-1, 1, 2), None)
    out["lk_next"], out["lk_status"] = nxt.reshape(-1, 2), st.reshape(-1).astype(np.uint8)
    ok = out["lk_status"].astype(bool)
    M, inl = cv2.estimateAffinePartial2D(out["corners0"][ok], out["lk_next"][ok], cv2.RANSAC)
    out["affine"], out["inliers"] = M, inl.reshape(-1).astype(np.uint8)
    try:
        sys.path.insert(0, "/root/reference/plugins/track")
        from bot_sort.gmc import GMC
        g = GMC(method="sparseOptFlow", downscale=2)
        g.apply(f0)
        out["gmc_apply"] = g.apply(f1)
    except Exception as e:                                   # the reference tree is optional here
        print("reference GMC not importable:", e)
    # StrongSORT's estimator: Track.ECC of plugins/track/strong_sort/sort/track.py:129-211 (oracle/src/ecc.c, tlk_ecc.hip)
    from test_oracle_ecc import ecc_pair
    e0, e1 = ecc_pair(5, 0.004, 13.0, -8.0, 540, 960)
    s0 = cv2.resize(cv2.cvtColor(e0, cv2.COLOR_BGR2GRAY), (0, 0), fx=0.1, fy=0.1, interpolation=cv2.INTER_LINEAR)
    s1 = cv2.resize(cv2.cvtColor(e1, cv2.COLOR_BGR2GRAY), (0, 0), fx=0.1, fy=0.1, interpolation=cv2.INTER_LINEAR)
    crit = (cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 100, 1e-5)
    rho, wm = cv2.findTransformECC(s0, s1, np.eye(2, 3, dtype=np.float32), cv2.MOTION_EUCLIDEAN, crit, None, 1)
    out["ecc_f0"], out["ecc_f1"], out["ecc_small0"], out["ecc_small1"], out["ecc_warp_small"], out["ecc_rho"] = e0, e1, s0, s1, wm.copy(), np.array(rho)
    wm[0, 2] = wm[0, 2] / 0.1; wm[1, 2] = wm[1, 2] / 0.1
    out["ecc_warp"] = wm
    for iters in (1, 2, 5):                                   # early iterates: pin the per-iteration arithmetic, not only the fixed point
        _, wi = cv2.findTransformECC(s0, s1, np.eye(2, 3, dtype=np.float32), cv2.MOTION_EUCLIDEAN, (cv2.TERM_CRITERIA_COUNT, iters, -1), None, 1)
        out[f"ecc_warp_small_it{iters}"] = wi
    np.savez_compressed(os.path.join(HERE, "cmc_opencv.npz"), **out)
    print("wrote cmc_opencv.npz with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
