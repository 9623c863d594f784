"""bench.py's launcher contract, checked without a GPU: `--gpus N` must never degrade to a silent single-GPU run
(round-1 review: it printed n_gpus 1 and exited 0)."""
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_flag_must_match_world_size():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr + out.stdout
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0


def test_launcher_starts_the_ranks_and_fails_when_they_fail():
    """No GPU in the CPU suite: every spawned rank exits with the 'needs a GPU' error, and the parent -- which must
    have started exactly two children without importing torch itself -- reports the failure with a non-zero code."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    import torch
    if torch.cuda.is_available():
        return  # on a GPU box the 2-rank run is exercised by tests/test_dist_gpu.py
    assert out.returncode != 0
    assert "rank exit codes [1, 1]" in out.stderr
    assert out.stderr.count("needs a GPU") == 2


def test_launcher_kills_the_other_ranks_when_one_fails_or_the_deadline_passes():
    """a rank stuck in a collective its peers never joined must not hold the GPUs.  Rank 1 hangs (a selftest
    hook in bench.py), rank 0 fails ('needs a GPU' here; any non-zero exit on a GPU box): the launcher kills rank 1 and
    returns 1 within seconds.  With every rank hanging, the overall deadline ends the run."""
    import time
    import torch
    if torch.cuda.is_available():
        return  # rank 0 would run the benchmark
    t0 = time.time()
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"GSPLAT_BENCH_SELFTEST": "hang:1"})
    assert out.returncode == 1 and time.time() - t0 < 120
    assert "rank(s) [0] failed" in out.stderr and "-9]" in out.stderr
    t0 = time.time()
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"GSPLAT_BENCH_SELFTEST": "hang:all", "GSPLAT_BENCH_DEADLINE_S": "2"})
    assert out.returncode == 1 and time.time() - t0 < 60
    assert "deadline passed" in out.stderr and "[-9, -9]" in out.stderr


def test_headline_guard_prints_the_line_exactly_once():
    """bench.py's HeadlineGuard (r05): whatever the optional legs behind the timed region do, rank 0's ONE line gets out.
    (a) finish() prints the line with the extras merged, once; (b) a main thread that never comes back (a hung
    collective) -- the timer prints the line as it stands and leaves with exit code 0."""
    import io
    import json
    sys.path.insert(0, ROOT)
    import bench
    buf = io.StringIO()
    g = bench.HeadlineGuard({"n_gpus": 2, "value": 3.0}, 60, out=buf).arm()
    assert g.finish({"exchange_ms_per_step": {"split": 1.0}}) is True
    assert g.finish({"exchange_ms_per_step": "again"}) is False
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "value": 3.0, "exchange_ms_per_step": {"split": 1.0}}
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.HeadlineGuard({'n_gpus': 2, 'value': 3.0}, 0.5, on_timeout={'exchange_ms_per_step': {'status': 'late'}}).arm(); "
            "time.sleep(120)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["exchange_ms_per_step"] == {"status": "late"} and json.loads(lines[0])["value"] == 3.0


def test_payload_sweep_reports_failures_and_stops_on_asymmetric_ones():
    """bench.sweep_payloads on two in-process ranks (CPU tensors): a payload that raises on EVERY rank is reported and
    left out, the others are timed (MAX over the ranks); one that raises on ONE rank only ends the sweep on all ranks
    (SweepAbort) -- the caller prints the headline regardless."""
    import importlib
    import torch
    sys.path.insert(0, ROOT)
    import bench
    gdist = importlib.import_module("3dgs_amd.dist")
    dev = torch.device("cpu")

    def everywhere(comm):
        def time_mode(mode):
            if mode == "b":
                raise RuntimeError("backend says no")
            return 1.0 + comm.rank
        return bench.sweep_payloads(comm, ("a", "b", "c"), time_mode, torch, dev)

    res = gdist.ThreadGroup(2).run(everywhere)
    assert res[0] == res[1] == {"a": 2.0, "b": "failed: RuntimeError: backend says no", "c": 2.0}

    def one_rank_only(comm):
        def time_mode(mode):
            if mode == "b" and comm.rank == 1:
                raise RuntimeError("only here")
            return 1.0
        try:
            bench.sweep_payloads(comm, ("a", "b", "c"), time_mode, torch, dev)
        except bench.SweepAbort as e:
            return str(e)
        return "no abort"

    res = gdist.ThreadGroup(2).run(one_rank_only)
    assert all("some ranks only" in r for r in res), res


def test_launcher_deadline_is_below_the_drivers_timeout():
    src = open(BENCH).read()
    assert 'os.environ.get("GSPLAT_BENCH_DEADLINE_S", "540")' in src  # the driver allows a run 600 s


def test_bench_line_describes_the_collectives_without_a_process_group():
    """r06: the line's `collectives` entry exists at N = 1 too (null where there is nothing to measure), so that the first
    multi-GPU record and a single-GPU one have the same keys."""
    sys.path.insert(0, ROOT)
    from conftest import pkg
    env = pkg("dist").collective_environment(None)
    assert env["rccl_world"] is None and env["backend"] is None
    for k in ("nccl_algo", "nccl_proto", "nccl_min_nchannels", "rccl_msccl_enable"):
        assert k in env
    src = open(BENCH).read()
    assert '"collectives": dict(gdist.collective_environment(' in src and "all_reduce_common_ms=None" in src
    assert 'extra["config2"] = config2_workload(' in src  # BASELINE configs[1] is in the line
