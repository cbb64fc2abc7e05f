"""Fold the passes of tools/pmc_conv.sh: per convolution kernel, the per-launch average of every counter collected
(launch 0 of each kernel is dropped as warm-up) plus the launch duration under the profiler.
    python tools/pmc_conv_table.py <dir>"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"conv_igemm_[a-z0-9_]+", name)
    return m.group(0) if m else None


def main():
    d = sys.argv[1]
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True)):
        seen = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            first = seen.setdefault(k, r["Dispatch_Id"])
            if r["Dispatch_Id"] == first:
                continue                                  # warm-up launch
            e = out.setdefault(k, {})
            c = e.setdefault(r["Counter_Name"], [0.0, set()])
            c[0] += float(r["Counter_Value"])
            c[1].add(r["Dispatch_Id"])
            t = e.setdefault("_ns_" + os.path.relpath(f, d).split(os.sep)[0], [0.0, set()])
            if r["Dispatch_Id"] not in t[1]:
                t[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                t[1].add(r["Dispatch_Id"])
    res = {}
    for k, e in out.items():
        res[k] = {}
        ns = [v[0] / len(v[1]) for n, v in e.items() if n.startswith("_ns_")]
        res[k]["avg_launch_us_under_profiler"] = round(sum(ns) / len(ns) / 1e3, 2)
        for n, v in sorted(e.items()):
            if not n.startswith("_ns_"):
                res[k][n] = v[0] / len(v[1])
        res[k]["launches_averaged"] = min(len(v[1]) for n, v in e.items() if not n.startswith("_ns_"))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
