#!/bin/bash
# Development check on the GPU box (gpurun): the whole -m gpu suite, smoke, the two CIFAR bench lines.  Writes gpurun_out/dev_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/dev_tests.log 2>&1
echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/dev_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/dev_smoke.log; lap smoke
timeout 100 python -m pytest tests/test_gpu_models.py -q -s -k "lean" 2>&1 | grep -a "lean vs\|passed\|failed" > gpurun_out/dev_lean.log; lap lean
timeout 200 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 > gpurun_out/dev_bench_cifar_b256.json 2> gpurun_out/dev_bench_cifar.err; lap cifar
timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 2 --warmup 1 > gpurun_out/dev_bench_cifar_adjoint_b128.json 2> gpurun_out/dev_bench_adjoint.err; lap adjoint
tail -4 gpurun_out/dev_tests.log | cut -c1-300; tail -2 gpurun_out/dev_smoke.log; cat gpurun_out/dev_lean.log
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/dev_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "sclk", (r.get("sclk_mhz") or {}).get("median"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat gpurun_out/dev_timeline.log
