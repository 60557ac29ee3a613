"""-m gpu: the bench entry point under the driver's torchrun command line ON the GPU with the RCCL backend (backend "nccl"): process-group
creation bound to the rank's device, barrier, max-time all-reduce, all-gather of the per-rank rates and the SUM all-reduce of hota.pack() run
through RCCL -- at world size 1, the only size a one-GPU box offers (two ranks cannot share a device). The N > 1 logic (spawn, partition,
reductions) is covered with gloo in tests/test_bench_launcher.py; this test covers what gloo cannot: RCCL itself with libtlk as the tracker."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_under_torchrun_with_rccl_collectives():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=REPO, TLK_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--workload", "config2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency-leg",
           "--check-frames", "32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == [[0, 0]] and len(j["per_rank_fps"]) == 1 and j["value"] > 100
    assert j["hota_allreduce"]["frames"] > 0 and 0.0 < j["hota_allreduce"]["HOTA"] <= 1.0
    assert j["parity"]["track_ids_equal_oracle"] is True
    assert str(j.get("collectives")).startswith("nccl")
