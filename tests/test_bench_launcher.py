"""CPU: the bench ENTRY POINT's multi-GPU launch path with world size 2 on gloo -- `python bench.py --gpus 2` re-executes itself
under torch.distributed.run, and the driver's own torchrun command line reaches the same code. `--dry-run` replaces the GPU step with a
stand-in (ground truth as tracker output): what is under test is the spawn, the 127.0.0.1 rendezvous, stream s -> rank s mod world, the
barrier / max-time reduction, the all-gather of per-rank rates and the SUM all-reduce of the HOTA statistics, and the JSON contract."""
import json
import os
import socket
import subprocess
import sys

from conftest import REPO

ARGS = ["--dry-run", "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "config2", "--frames-per-step", "4", "--objects", "8"]


def _check(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                      # exactly ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert sorted(r for r, _ in j["ranks_seen"]) == [0, 1] and len(j["per_rank_fps"]) == 2 and all(f > 0 for f in j["per_rank_fps"])
    assert j["streams_of_rank0"] == [0]                                        # stream s -> rank s mod world, one stream per rank
    assert j["hota_allreduce"]["frames"] == 2 * 3 * 4                          # both ranks' frames arrived through the all-reduce
    assert abs(j["hota_allreduce"]["HOTA"] - 1.0) < 1e-12                      # the stand-in tracker is perfect
    assert j["config"]["parallelism"] == "stream-parallel x2" and j["dry_run"] is True and j["value"] is None


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = REPO
    return env


def test_bench_gpus_flag_spawns_the_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + ARGS, capture_output=True, text=True, timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(r.stdout)


def test_bench_under_the_drivers_torchrun_command_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py")] + ARGS
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(r.stdout)


def test_bench_refuses_more_gpus_than_visible():
    """Without --dry-run the launcher must not silently run on fewer devices than asked for."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=REPO)
    assert r.returncode != 0 and "--gpus 64" in (r.stderr + r.stdout)


def test_eight_ranks_like_the_drivers_scaling_run():
    """world size 8 on gloo through the bench entry point: one stream per rank, all eight rates and placements gathered, the HOTA
    statistics of all eight streams summed, every rank on its own slice of the host CPUs"""
    args = [a if a != "2" or i != 2 else "8" for i, a in enumerate(ARGS)]
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and sorted(x for x, _ in j["ranks_seen"]) == list(range(8)) and len(j["per_rank_fps"]) == 8
    assert j["hota_allreduce"]["frames"] == 8 * 3 * 4 and abs(j["hota_allreduce"]["HOTA"] - 1.0) < 1e-12
    assert j["config"]["parallelism"] == "stream-parallel x8" and j["streams_of_rank0"] == [0]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert all(p[2] == ncpu // 8 for p in j["rank_placement"]), j["rank_placement"]      # [rank, node, cpus]: eight equal, disjoint slices


def test_cpu_slices_are_disjoint_and_numa_local():
    from tracklab_amd import dist as tdist
    cpus = list(range(64))
    sl = [tdist.even_cpu_slice(r, 8, cpus) for r in range(8)]
    assert all(len(s_) == 8 for s_ in sl) and sorted(sum(sl, [])) == cpus
    assert tdist.even_cpu_slice(0, 1, cpus) == cpus and tdist.even_cpu_slice(3, 8, [0, 1]) == [0, 1]
    node1 = list(range(32, 64))                                # GPUs 4..7 hang off NUMA node 1
    got = [tdist.numa_cpu_slice(node1, [4, 5, 6, 7], d, allowed=range(64)) for d in (4, 5, 6, 7)]
    assert got[0] == list(range(32, 40)) and got[3] == list(range(56, 64)) and sorted(sum(got, [])) == node1


def test_warmup_sections_of_the_ranks_do_not_overlap():
    """r05: every rank's warm-up (graph capture, library tuning, pinned allocations in a real run) passes a
    node-wide lock one at a time; the dry run exercises the same lock with a stand-in section and reports that the eight windows are disjoint,
    and that every rank sits on a non-empty CPU slice that can be disjoint from the others'."""
    args = [a if a != "2" or i != 2 else "8" for i, a in enumerate(ARGS)]
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["warmup_serialized"] is True and j["rank_placement_sound"] is True


def test_serialized_lock_excludes_across_processes(tmp_path):
    import multiprocessing as mp
    import time
    from tracklab_amd import dist as tdist
    os.environ["TLK_LOCK_DIR"] = str(tmp_path)
    try:
        def worker(q):
            order = []
            with tdist.serialized("t", order=order):
                time.sleep(0.15)
            q.put((order[0][1], order[1][1]))
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(q,)) for _ in range(3)]
        for p_ in ps:
            p_.start()
        wins = sorted(q.get(timeout=30) for _ in ps)
        for p_ in ps:
            p_.join(timeout=30)
        assert all(a[1] <= b[0] + 1e-6 for a, b in zip(wins, wins[1:])), wins
        assert tdist.placement_is_sound([[0, -1, 8], [1, -1, 8]], 2, 16) and not tdist.placement_is_sound([[0, -1, 0], [1, -1, 8]], 2, 16)
        assert not tdist.placement_is_sound([[0, -1, 16], [1, -1, 16]], 2, 16)
    finally:
        os.environ.pop("TLK_LOCK_DIR", None)


def test_serialized_lock_survives_an_unusable_lock_directory(tmp_path):
    """a lock file that cannot be opened (directory gone, a file another user left behind) must not cost the job its warm-up: the
    section runs unlocked and says so; the file name carries the uid so that another user's file is never touched."""
    from tracklab_amd import dist as tdist
    os.environ["TLK_LOCK_DIR"] = str(tmp_path / "does" / "not" / "exist")
    try:
        order = []
        with tdist.serialized("t", order=order) as s:
            assert s.locked is False
        assert [e for e, _ in order] == ["enter", "exit"]
        assert f"_{os.getuid()}_" in os.path.basename(s.path)
    finally:
        os.environ.pop("TLK_LOCK_DIR", None)


def test_single_rank_process_group_runs_the_collectives_on_a_real_backend():
    """the N = 1 bench line initialises a one-rank process group so that barrier / all-reduce / all-gather run on the job's
    backend (RCCL on the GPU box; gloo here) instead of being skipped."""
    code = ("import numpy as np, torch\n"
            "from tracklab_amd import dist as tdist\n"
            "d = tdist.init_single('gloo')\n"
            "assert d.is_initialized() and d.get_world_size() == 1\n"
            "d.barrier()\n"
            "assert tdist.allreduce_max(3.5, d, torch.device('cpu')) == 3.5\n"
            "assert np.array_equal(tdist.allreduce_sum(np.arange(4.0), d, torch.device('cpu')), np.arange(4.0))\n"
            "d.destroy_process_group()\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
