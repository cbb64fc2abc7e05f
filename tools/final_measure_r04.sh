#!/bin/bash
# Round-4 closing measurements on the GPU box (gpurun): [the whole -m gpu suite, smoke,] the bench lines, rocprofv3 kernel stats of the
# default command, the PMC passes (traffic of the bench workload; counters of the dominant kernel) - ALL AT HEAD.  Everything lands
# under gpurun_out/final/ (copied into profiles/r04/ afterwards).
#   bash tools/final_measure_r04.sh          everything
#   bash tools/final_measure_r04.sh quick    default bench line + rocprofv3 kernel stats only
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O="$R/gpurun_out/final"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> "$O/timeline.log"; }
: > "$O/timeline.log"
QUICK=${1:-}
if [ -z "$QUICK" ]; then
  timeout 900 python -m pytest tests -m gpu -q -s > "$O/gpu_tests.log" 2>&1; echo "rc=$?" >> "$O/gpu_tests.log"; lap gpu_tests
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "rc=$?" >> "$O/smoke.log"; lap smoke
fi
timeout 400 python bench.py --steps 3 --warmup 1 > "$O/bench_default_f16sr_b64.json" 2> "$O/bench_default.err"; lap bench_default
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof" -o run --output-format csv -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err" ); lap rocprof_stats
find "$O/prof" -name "*kernel_trace.csv" -delete; find "$O/prof" -name "*agent_info.csv" -delete
if [ -z "$QUICK" ]; then
  timeout 400 bash tools/pmc_traffic.sh --precision f16sr > "$O/pmc.log" 2>&1; lap pmc_traffic
  find "$R/gpurun_out/pmc_traffic" -name "*counter_collection.csv" -delete
  PMC_GROUPS="sq1 sq2 sq3 tcc1 grbm" timeout 500 bash tools/pmc_conv.sh dw_r04 --dw 1 --res16 --f16out > "$O/pmc_conv.log" 2>&1; lap pmc_conv
  for P in f16sr f16x3; do
    timeout 200 python bench.py --workload cifar32_ncsnpp --precision $P --steps 3 --warmup 1 $([ $P = f16x3 ] && echo --no-cpu-baseline) > "$O/bench_cifar_b256_$P.json" 2> "$O/bench_cifar_$P.err"; lap bench_cifar_$P
    timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --precision $P --steps 2 --warmup 1 $([ $P = f16x3 ] && echo --no-cpu-baseline) > "$O/bench_cifar_adjoint_b128_$P.json" 2> "$O/bench_adjoint_$P.err"; lap bench_adjoint_$P
  done
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$O/prof_adjoint" -o run --output-format csv -- python "$R/bench.py" --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0 --no-cpu-baseline > "$O/bench_adjoint_t10_under_rocprof.json" 2> "$O/rocprof_adjoint.err" ); lap rocprof_adjoint
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$O/prof_cifar" -o run --output-format csv -- python "$R/bench.py" --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0 --no-cpu-baseline > "$O/bench_cifar_t10_under_rocprof.json" 2> "$O/rocprof_cifar.err" ); lap rocprof_cifar
  find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
  timeout 200 python bench.py --t 150 --dt 1.5e-3 --steps 1 --warmup 1 --no-cpu-baseline > "$O/bench_t150_dt1.5e-3_100step.json" 2> "$O/bench_t150a.err"; lap bench_t150_100
  timeout 200 python bench.py --t 150 --steps 1 --warmup 1 --no-cpu-baseline > "$O/bench_t150_150step.json" 2> "$O/bench_t150b.err"; lap bench_t150_150
  timeout 100 python tests/probes/dw8_lifetime.py > "$O/dw8_lifetime.log" 2>&1; lap lifetime
  timeout 150 python tests/probes/dw8_timeline.py > "$O/dw8_timeline.log" 2>&1; lap timeline
  timeout 100 python tests/probes/conv_epi_probe.py > "$O/conv_epi_probe.log" 2>&1; lap epi_probe
  timeout 100 python tests/probes/gn_bwd_one_pass_probe.py > "$O/gn_bwd_one_pass_probe.log" 2>&1; lap gn_bwd_probe
  grep -E "passed|failed" "$O/gpu_tests.log" | tail -2; grep -E "^FAILED|^ERROR" "$O/gpu_tests.log" | head; tail -2 "$O/smoke.log"
fi
cat "$O/timeline.log"
python - <<'P'
import json, glob, os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/final/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        g = r.get("second_kernel") or {}
        print(os.path.basename(f), "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "frac", r.get("frac") and round(r["frac"], 3),
              "held", r.get("frac_at_held_clock") and round(r["frac_at_held_clock"], 3), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3),
              "sclk", (r.get("sclk_mhz") or {}).get("median"), "traffic", r.get("traffic"), "GN GB/s", g.get("achieved") and round(g["achieved"]),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "cores", (d.get("cpu_baseline") or {}).get("cores"))
    except Exception as e:
        print(f, "unreadable", e)
P
head -18 "$O/prof/run_kernel_stats.csv" | cut -c1-150
