#!/bin/bash
# HBM traffic of every kernel of the bench workload from rocprofv3 PMC counters, collected exactly as
# /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (they do not fit one pass), --kernel-trace only (no other trace domains).  Run on the GPU box from the repo root:
#     bash tools/pmc_traffic.sh [bench.py arguments ...]         e.g.  --precision f16x2 --batch 64
# The run is two reverse-SDE steps (--t 2: two UNet calls, every layer shape twice); results land under
# gpurun_out/pmc_traffic/ and are folded into profiles/pmc_traffic.json by tools/pmc_traffic_table.py.
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmc_traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/$C"
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/$C" -o run --output-format csv -- \
      python "$REPO/bench.py" --t 2 --steps 1 --warmup 0 --no-cpu-baseline --no-conv-profile --no-resident-call "$@" > "$OUT/$C.log" 2>&1
  echo "$C rc=$?"
  find "$OUT/$C" -name "*agent_info.csv" -delete
done
python "$REPO/tools/pmc_traffic_table.py" "$OUT" "$@"
