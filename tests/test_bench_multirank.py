"""bench.py's multi-rank branch executed for real - the driver's own launch line (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), two CPU ranks over gloo,
with `--stub-engine` standing in for the purification engine: rank / world from the launcher's environment, per-rank
shards keyed by global sample index, the all_gather inside the timed region, barrier + MAX-over-ranks timing, ONE JSON line
from rank 0.  (No 8-GPU node is available to the builder; the RCCL path differs from this one in the backend name only.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [1, 2])
def test_bench_launch_gather_and_timing_harness(n):
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--stub-engine"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
               "--stub-engine"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == n * d["config"]["per_gpu_batch"]
    assert d["gather_ok"] is True
    assert abs(d["value"] - d["config"]["global_batch"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]


def test_bench_harness_at_world_8_takes_the_max_over_ranks():
    """The driver's 8-GPU launch line over gloo with the stub engine: eight ranks, shards in rank order, and the reported time is
    the MAX over ranks - rank 5 is made to sleep 0.25 s per call, so every step must take at least that long although seven of
    the eight ranks finish at once.  (No 8-GPU node is available to the builder: NO scaling curve has been measured.)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--stub-engine", "--stub-slow-rank", "5", "--stub-sleep", "0.25", "--batch", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == 8 and d["config"]["global_batch"] == 24 and d["gather_ok"] is True
    assert d["ms_per_step"] >= 250.0, d["ms_per_step"]
    # round 5: the line explains itself - every rank's own ms per step (the headline is their MAX), its held clock (None on CPU ranks),
    # the time it spent inside the all-gather and its engine build time
    r = d["ranks"]
    assert all(len(r[k]) == 8 for k in ("ms_per_step_per_rank", "sclk_mhz_median_per_rank", "all_gather_ms_per_step_per_rank", "engine_build_s_per_rank"))
    assert abs(max(r["ms_per_step_per_rank"]) - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    assert all(v is not None and v >= 0.0 for v in r["all_gather_ms_per_step_per_rank"])
    # the slow rank arrives last: it waits least inside the collective, the others wait for it there
    ag = r["all_gather_ms_per_step_per_rank"]
    assert ag[5] == min(ag) and max(ag) >= 200.0, ag
    assert d["collectives"]["all_gather_ms"] == max(ag)
    assert abs(d["value"] - 24 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]


def test_bench_refuses_a_world_size_that_does_not_match_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
