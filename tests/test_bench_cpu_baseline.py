"""bench.py's `cpu_baseline` leg on CPU: persistent pinned workers, the sweep over concurrent workers, the early stop, and that a
failure never raises (the GPU line it belongs to is already measured).  Small: the CIFAR-10 score network at batch 4, 2 threads per worker."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("workload", ["cifar32_ncsnpp", "cifar32_ncsnpp_adjoint"])
def test_cpu_baseline_sweep_over_pinned_workers(workload, monkeypatch):
    if len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs 4 host cores")
    monkeypatch.setenv("DIFFPURE_CPU_WORKER_THREADS", "2")
    monkeypatch.setenv("DIFFPURE_CPU_WORKERS", "2")
    b = _bench()
    out = b.cpu_baseline(workload, 100, 100, 1234, budget_s=0.5, start_timeout=300.0)
    assert out["value"] is not None and out["value"] > 0, out
    assert out["unit"] == "images/s" and out["kind"] in ("reference", "port")
    from oracle import ref_loader
    assert out["kind"] == ("reference" if ref_loader.available() else "port")       # the reference's own modules whenever oracle/_ref is there
    sweep = out["host"]["sweep"]
    assert [p["workers"] for p in sweep] == [1, 2] and out["host"]["workers_started"] == 2
    assert all(len(p["per_worker"]) == p["workers"] and p["calls"] >= 1 for p in sweep)
    assert out["value"] == max(p["value"] for p in sweep)
    assert out["cores"] == 2 * max(sweep, key=lambda p: p["value"])["workers"]
    if workload.endswith("_adjoint"):
        assert all(p["calls_fb"] >= 1 for p in sweep) and "adjoint steps" in out["sample"]


def test_cpu_baseline_never_raises(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "executable", "/nonexistent/python")       # no worker can start
    out = b.cpu_baseline("cifar32_ncsnpp", 100, 100, 1, budget_s=0.1, start_timeout=5.0)
    assert out["value"] is None and "cpu_baseline failed" in out["sample"]


def test_cpu_sweep_stops_after_two_declining_points():
    """the rule itself, on the closing measurement's numbers (guided UNet: 1 / 2 / 4 / 8 / 16 workers)"""
    vals = [0.0053, 0.0062, 0.0048, 0.0049, 0.0033]
    sweep, best = [], None
    for v in vals:
        sweep.append(v)
        best = v if best is None else max(best, v)
        if len(sweep) >= 3 and max(sweep[-1], sweep[-2]) < best:
            break
    assert sweep == vals[:4]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'len(sweep) >= 3 and max(sweep[-1]["value"], sweep[-2]["value"]) < best["value"]' in src
