#!/bin/bash
# Short development check on the GPU box (gpurun): kernel tests of the convolution variants, smoke, the two CIFAR bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 400 python -m pytest tests/test_gpu_ops.py -q -k "conv2d" > gpurun_out/dev_tests.log 2>&1
echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/dev_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/dev_smoke.log; lap smoke
timeout 200 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_cifar.json 2> gpurun_out/dev_bench_cifar.err; lap cifar
DIFFPURE_LEAN=0 timeout 200 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_cifar_nolean.json 2> gpurun_out/dev_bench_cifar_nolean.err; lap cifar_nolean
timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_adjoint.json 2> gpurun_out/dev_bench_adjoint.err; lap adjoint
tail -3 gpurun_out/dev_tests.log; tail -2 gpurun_out/dev_smoke.log
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/dev_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "sclk", (r.get("sclk_mhz") or {}).get("median"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat gpurun_out/dev_timeline.log
