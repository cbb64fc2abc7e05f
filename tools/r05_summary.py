#!/usr/bin/env python
"""Print the numbers DESIGN.md / README.md quote from a tools/r05_measure.sh output directory:  python tools/r05_summary.py gpurun_out/r05/closing"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    r = j["roofline"]
    g = r.get("second_kernel") or {}
    cb = j.get("cpu_baseline") or {}
    print(f"{os.path.basename(f)}: value {j['value']:.3f} {j['unit']}; ms/step {j['ms_per_step']:.0f}; resident {((j.get('input') or {}).get('value_resident_batch_engine_call'))}; "
          f"conv {r.get('achieved')} TF frac {r.get('frac')} held {r.get('frac_at_held_clock')} share {r.get('time_share_of_step')} avg_launch_ms {r.get('avg_launch_ms')}; "
          f"sclk {(r.get('sclk_mhz') or {}).get('median')}; GN {g.get('achieved')} GB/s share {g.get('time_share_of_step')}; traffic {r.get('traffic')}; "
          f"unet TF {r.get('end_to_end_unet_tflops_per_gpu')}; peak GiB {j.get('peak_device_memory_gib')}; h2d {(j.get('input') or {}).get('h2d_ms_per_batch_pinned')}; "
          f"cpu {cb.get('value')} {cb.get('kind')} cores {cb.get('cores')} sweep {[(p.get('workers'), p.get('value')) for p in (cb.get('host') or {}).get('sweep', [])]}")
for f in sorted(glob.glob(os.path.join(d, "*_kernel_stats.csv"))):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {os.path.basename(f)}: {tot / 1e6:.1f} ms of kernels, {sum(int(r['Calls']) for r in rows)} launches")
    for r in rows[:16]:
        print(f"   {r['Name'][:100]:100s} n={r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):6.2f} %")
    for r in rows:
        if "FillFunctor" in r["Name"] or "copyBuffer" in r["Name"] or "elementwise_kernel" in r["Name"]:
            print(f"   [ATen/runtime] {r['Name'][:90]:90s} n={r['Calls']:>6s} {float(r['Percentage']):6.3f} %")
