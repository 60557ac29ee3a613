#!/usr/bin/env python3
"""Fixture generator for the IMAGE side of the hot path -- runs ONLY where OpenCV (cv2) is installed; rtmlib and albumentations are used when they
are importable too.  Neither is in the container this repo is built in, which is why rows W3 / N1 / N2 of DESIGN.md ("detector / ReID / pose
adapters", letterbox, decode + NMS) are "parity unpinned": oracle/src/image.c and pose.c restate the published arithmetic, but nothing produced by
the libraries themselves pins them.  This script writes tests/golden/image_opencv.npz with outputs of the libraries for seeded inputs:

  lb{i}_*      detector pre-processing of five frame sizes: rtmlib `YOLOX.preprocess` (tracklab/wrappers/bbox_detector/rtmlib_api.py:27-46 ->
               rtmlib 0.0.13 tools/object_detection/yolox.py: ratio, cv2.resize INTER_LINEAR, 114 padding) -- padded uint8 image + ratio
  rz{k}_*      `cv2.resize(crop, (w, h), interpolation=cv2.INTER_LINEAR)` of ReID crops (384x128 and 256x128 targets; clipped boxes, 1-4 px
               boxes, up- and down-scaling) = albumentations `Resize` inside the ReID transform (tracklab/wrappers/reid/kpreid_api.py:115-144),
               plus albumentations `Normalize` itself when installed
  wa{k}_*      `cv2.warpAffine(img, M, (192, 256), flags=cv2.INTER_LINEAR)` of pose crops with rtmlib's warp matrices; with rtmlib installed also
               `RTMPose.preprocess` (centre, scale, normalised crop) and `RTMPose.postprocess` (SimCC decode) on seeded SimCC maps
               (tracklab/wrappers/pose_estimator/rtmlib_api.py:27-33)
  yx_*         rtmlib `YOLOX.postprocess` (decode, multiclass NMS 0.45 / 0.7, > 0.3, class 0) on a seeded synthetic head

Every section records where its numbers came from (`*_source`: "rtmlib", "albumentations" or "cv2" = the library's few lines restated around the cv2
call).  The consumers skip while the file is absent: tests/test_oracle_image_golden.py (oracle vs fixture, CPU) and
tests/test_gpu_image_golden.py (HIP kernels vs fixture).  Usage: python tests/golden/make_image_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

FRAME_SIZES = [(540, 960), (720, 1280), (480, 854), (1000, 600), (333, 517)]
REID_TARGETS = [(384, 128), (256, 128)]
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
RTMPOSE_MEAN, RTMPOSE_STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)


def textured(rng, h, w):
    """A frame with gradients, edges and noise (every bilinear weight is exercised), seeded."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], axis=-1).astype(np.int32)
    img += rng.integers(-20, 20, img.shape)
    for _ in range(12):
        t, l = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
        img[t:t + int(rng.integers(4, h // 3)), l:l + int(rng.integers(4, w // 4))] = rng.integers(0, 256, 3)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    import cv2
    try:
        from rtmlib.tools.object_detection.yolox import YOLOX
        from rtmlib.tools.pose_estimation.rtmpose import RTMPose
        import rtmlib
        have_rtmlib = True
    except Exception as e:                                   # noqa: BLE001
        print("rtmlib not importable, its few lines around cv2 are restated:", e)
        have_rtmlib = False
    try:
        import albumentations as A
        have_alb = True
    except Exception as e:                                   # noqa: BLE001
        print("albumentations not importable, Normalize is skipped:", e)
        have_alb = False
    rng = np.random.default_rng(2024)
    out = {"cv2_version": np.array(cv2.__version__), "rtmlib_version": np.array(getattr(rtmlib, "__version__", "?") if have_rtmlib else "absent")}

    # ---- detector pre-processing
    det = SimpleNamespace(model_input_size=(640, 640))
    for i, (h, w) in enumerate(FRAME_SIZES):
        img = textured(rng, h, w)
        if have_rtmlib:
            padded, ratio = YOLOX.preprocess(det, img)
            src = "rtmlib"
        else:                                                # rtmlib 0.0.13 yolox.py preprocess, restated around the cv2 call
            padded = np.ones((640, 640, 3), dtype=np.uint8) * 114
            ratio = min(640 / img.shape[0], 640 / img.shape[1])
            resized = cv2.resize(img, (int(img.shape[1] * ratio), int(img.shape[0] * ratio)), interpolation=cv2.INTER_LINEAR).astype(np.uint8)
            padded[:int(img.shape[0] * ratio), :int(img.shape[1] * ratio)] = resized
            src = "cv2"
        out[f"lb{i}_img"], out[f"lb{i}_padded"], out[f"lb{i}_ratio"], out[f"lb{i}_source"] = img, np.asarray(padded), np.float64(ratio), np.array(src)
    out["lb_n"] = np.int64(len(FRAME_SIZES))

    # ---- ReID crops: cv2.resize (albumentations Resize) + Normalize
    frame = textured(rng, 540, 960)
    out["rz_frame"] = frame
    boxes = [(100, 50, 180, 300), (0, 0, 60, 200), (900, 400, 960, 540), (300, 100, 301, 104), (500, 200, 504, 201), (10, 10, 400, 500),
             (640, 30, 700, 90), (200, 300, 203, 303)]                                                   # l, t, r, b (already clipped + rounded)
    k = 0
    for (l, t, r, b) in boxes:
        crop = np.ascontiguousarray(frame[t:b, l:r])
        for (oh, ow) in REID_TARGETS:
            res = cv2.resize(crop, (ow, oh), interpolation=cv2.INTER_LINEAR)
            out[f"rz{k}_ltrb"], out[f"rz{k}_hw"], out[f"rz{k}_resized"] = np.array([l, t, r, b], np.int32), np.array([oh, ow], np.int32), res
            if have_alb:
                tf = A.Compose([A.Resize(oh, ow), A.Normalize(mean=IMAGENET_MEAN, std=IMAGENET_STD)])
                out[f"rz{k}_normalized"] = tf(image=crop)["image"].astype(np.float32)
            k += 1
    out["rz_n"] = np.int64(k)
    out["rz_source"] = np.array("albumentations" if have_alb else "cv2")

    # ---- pose crops: warpAffine with rtmlib's matrices (+ RTMPose.preprocess / postprocess themselves)
    import oracle                                            # only for the warp matrix when rtmlib is absent (the matrix is an INPUT here)
    oracle.build()
    pose = SimpleNamespace(model_input_size=(192, 256), mean=RTMPOSE_MEAN, std=RTMPOSE_STD)
    pframe = textured(rng, 720, 1280)
    out["wa_frame"] = pframe
    pboxes = [(100, 50, 300, 600), (0, 0, 200, 300), (1100, 500, 1280, 720), (600, 300, 640, 420), (20, 600, 500, 700), (640.4, 100.7, 700.2, 333.3)]
    for k, bb in enumerate(pboxes):
        if have_rtmlib:
            crop, center, scale = RTMPose.preprocess(pose, pframe, list(bb))
            out[f"wa{k}_center"], out[f"wa{k}_scale"], out[f"wa{k}_crop"] = np.asarray(center, np.float64), np.asarray(scale, np.float64), np.asarray(crop)
            try:
                from rtmlib.tools.pose_estimation.pre_processings import get_warp_matrix
                M = get_warp_matrix(np.asarray(center), np.asarray(scale), 0, output_size=(192, 256))
            except Exception as e:                           # noqa: BLE001  (module layout differs between rtmlib versions)
                print("rtmlib get_warp_matrix not importable, using the restated matrix as INPUT:", e)
                M = oracle.rtmpose_warp_matrix(center, scale)
        else:
            _, center, scale = oracle.rtmpose_preprocess(pframe, bb)
            M = oracle.rtmpose_warp_matrix(center, scale)
        out[f"wa{k}_box"], out[f"wa{k}_M"] = np.asarray(bb, np.float64), np.asarray(M, np.float64)
        out[f"wa{k}_warped"] = cv2.warpAffine(pframe, np.asarray(M, np.float64), (192, 256), flags=cv2.INTER_LINEAR)
        if have_rtmlib:
            sx = rng.normal(0, 1, (1, 17, 384)).astype(np.float32)
            sy = rng.normal(0, 1, (1, 17, 512)).astype(np.float32)
            sx[0, 3] = -np.abs(sx[0, 3])                      # a keypoint whose score is not positive: location -1
            kp, sc = RTMPose.postprocess(pose, (sx, sy), center, scale)
            out[f"wa{k}_simcc_x"], out[f"wa{k}_simcc_y"] = sx[0], sy[0]
            out[f"wa{k}_kpts"], out[f"wa{k}_scores"] = np.asarray(kp, np.float64)[0], np.asarray(sc, np.float32)[0]
    out["wa_n"] = np.int64(len(pboxes))
    out["wa_source"] = np.array("rtmlib" if have_rtmlib else "cv2")

    # ---- detector post-processing
    if have_rtmlib:
        from tracklab_amd.synth import SyntheticStream, synth_yolox_head
        fr = SyntheticStream(3, 60, 1).step()
        ratio = min(640 / 1080, 640 / 1920)
        head = synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio)
        yx = SimpleNamespace(model_input_size=(640, 640), nms_thr=0.45, score_thr=0.7)
        boxes_out = YOLOX.postprocess(yx, head[None].copy(), ratio)
        out["yx_head"], out["yx_ratio"], out["yx_boxes"] = head, np.float64(ratio), np.asarray(boxes_out, np.float32).reshape(-1, 4)
    path = os.path.join(HERE, "image_opencv.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({os.path.getsize(path) / 1e6:.1f} MB; rtmlib {'yes' if have_rtmlib else 'no'}, albumentations {'yes' if have_alb else 'no'})")


if __name__ == "__main__":
    main()
