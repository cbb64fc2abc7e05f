cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x > $O/dev_tests_all.log 2>&1; echo "rc=$?" >> $O/dev_tests_all.log
grep -E "passed|failed" $O/dev_tests_all.log | tail -2; grep -E "^FAILED|^ERROR" $O/dev_tests_all.log | head -10 | cut -c1-200
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_cifar" -o run --output-format csv -- python "$R/bench.py" --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0 --no-cpu-baseline > "$R/$O/bench_cifar_t10_under_rocprof.json" 2> "$R/$O/rocprof_cifar.err" )
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/$O/prof_adjoint" -o run --output-format csv -- python "$R/bench.py" --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0 --no-cpu-baseline > "$R/$O/bench_adjoint_t10_under_rocprof.json" 2> "$R/$O/rocprof_adjoint.err" )
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in prof_cifar prof_adjoint; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -24 "$f" | cut -c1-150; done
