"""world_size-2 gloo run of the multi-GPU logic on CPU: stream partitioning, barrier/max-time, and the per-epoch
SUM all-reduce of the HOTA sufficient statistics and the CLEAR-MOT / ID counts == the single-process result over all streams."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import REPO

N_STREAMS = 4
NVEC = 19 * 7 + 2 + 19          # HOTA statistics + clearmot.SUM_FIELDS


def _stream_stats(s):
    import oracle
    from tracklab_amd import hota
    from tracklab_amd.synth import SyntheticStream
    hyper = dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115,
                 iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False)
    from tracklab_amd import clearmot
    trk = oracle.OCSort(**hyper)
    acc = clearmot.MOTAccumulator()
    gt, tr = [], []
    ltwh = lambda b: np.column_stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]]).reshape(-1, 4)
    for fr in SyntheticStream(s, 12, 40, miss_prob=0.1):
        out = oracle.ocsort_wrapper_step(trk, fr["dets"], 0.4)
        gt.append((fr["gt_all_ids"], fr["gt_boxes"]))
        tr.append((out[:, 4].astype(int), out[:, :4]))
        acc.update_boxes(fr["gt_all_ids"], ltwh(fr["gt_boxes"]), out[:, 4].astype(int), ltwh(out[:, :4]), max_iou=0.5)
    return np.concatenate([hota.pack(hota.hota_sequence(*hota.sequence_from_rows(gt, tr)), frames=40, seconds=0.0), clearmot.pack(acc.counts())])


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tracklab_amd import dist as D
    dist = D.init("gloo")
    mine = D.streams_for_rank(N_STREAMS, rank, world)
    vec = sum((_stream_stats(s) for s in mine), np.zeros(NVEC))
    dist.barrier()
    tmax = D.allreduce_max(float(rank + 1), dist)
    total = D.allreduce_sum(vec, dist)
    q.put((rank, mine, tmax, total))
    dist.destroy_process_group()


def test_two_rank_metric_allreduce_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]              # stream s -> rank s mod world
    assert res[0][2] == res[1][2] == 2.0                           # max over ranks
    expected = sum((_stream_stats(s) for s in range(N_STREAMS)), np.zeros(NVEC))
    for _, _, _, total in res:
        np.testing.assert_allclose(total, expected, rtol=1e-13)
    from tracklab_amd import hota
    from tracklab_amd import clearmot
    fin = hota.finalize(res[0][3][:19 * 7 + 2])
    assert fin["frames"] == 40 * N_STREAMS and 0.3 < fin["summary"]["HOTA"] <= 1.0
    cm = clearmot.finalize(clearmot.unpack(res[0][3][19 * 7 + 2:]))
    assert cm["num_frames"] == 40 * N_STREAMS and 0.5 < cm["mota"] <= 1.0 and 0.5 < cm["idf1"] <= 1.0 and cm["num_objects"] > 1000
