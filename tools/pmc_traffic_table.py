"""Fold the rocprofv3 --pmc passes of tools/pmc_traffic.sh into per-kernel HBM bytes per launch.

    python tools/pmc_traffic_table.py <dir with FETCH_SIZE/ and WRITE_SIZE/> [bench.py arguments used]

Counter units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests of wide (16 B / lane) streaming reads - global_load and LDS-DMA
alike - at 64 bytes, i.e. reports ONE HALF: doubled here for the kernels whose loads are of that kind (every kernel
of this engine reads with 16-byte lanes).  WRITE_SIZE is taken as reported (round 1 calibrated it on the
convolution output: exactly the tensor's bytes).  Writes <dir>/summary.json and prints the row for
profiles/pmc_traffic.json (dominant kernel: conv_igemm_h2_pp 3x3 launches).
"""
import argparse
import csv
import glob
import json
import os
import re
import sys


def short(name):
    # round 6: one row per kernel NAME (round 5 folded the 256-wide-tile kernels into one row, whose per-launch average over 232 launches
    # of three kernels could not be compared with the algorithmic bytes of the dominant kernel's own launches)
    m = re.search(r"(conv_igemm_dw|conv_igemm_dh|conv_igemm_sw|conv_igemm_h2_pp|conv_igemm_nn|conv_stem_h16)", name)
    if m:
        return m.group(1)
    m = re.search(r"(conv_igemm_h2|conv_igemm_f32|splitk_epilogue|gn_apply_h16|gn_apply_h2q|gn_apply|round_weights|gn_finalize_cols|gn_stats|"
                  r"gn_finalize|attn_flash|attn_pack|em_step|ddpm_step|temb|softmax_rows|gemm_strided|philox|axpby|silu|pack_h2)", name)
    return m.group(1) if m else name[:60]


def load(d, counter):
    rows = {}
    for f in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            e = rows.setdefault(k, dict(launches=0, kb=0.0, ns=0))
            e["launches"] += 1
            e["kb"] += float(r["Counter_Value"])
            e["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows


def main():
    d = sys.argv[1]
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--workload", default="imagenet256_guided")
    a, _ = ap.parse_known_args(sys.argv[2:])
    fetch, write = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, {}).get("kb", 0) + write.get(k, {}).get("kb", 0))):
        f, w = fetch.get(k), write.get(k)
        n = (f or w)["launches"]
        fb = 2.0 * 1024 * f["kb"] / f["launches"] if f else None          # gfx950: FETCH_SIZE reports one half
        wb = 1024 * w["kb"] / w["launches"] if w else None
        ms = (f or w)["ns"] / (f or w)["launches"] * 1e-6
        out[k] = dict(launches=n, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                      hbm_bytes_per_launch=(fb or 0) + (wb or 0), avg_ms_under_profiler=ms,
                      hbm_tb_per_s_under_profiler=((fb or 0) + (wb or 0)) / (ms * 1e-3) / 1e12 if ms else None)
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
    dom = out.get("conv_igemm_dw") or out.get("conv_igemm_sw") or out.get("conv_igemm_h2_pp")
    B = a.batch or (64 if a.workload == "imagenet256_guided" else (128 if a.workload.endswith("_adjoint") else 256))
    if dom:
        keep = ("conv_igemm_dw", "conv_igemm_dh", "conv_igemm_sw", "conv_igemm_h2_pp", "conv_igemm_h2", "conv_igemm_nn", "conv_stem_h16",
                "gn_apply_h16", "gn_finalize_cols", "round_weights", "attn_flash", "splitk_epilogue")
        row = dict(workload=a.workload, per_gpu_batch=B, precision=a.precision,
                   kernel="conv_igemm_dw (the dominant kernel: every launch of it in two UNet calls, 3x3 and 1x1); per_kernel holds one row per kernel name",
                   launches_profiled=dom["launches"], unet_calls_profiled=2, fetch_bytes_per_launch=dom["fetch_bytes_per_launch"],
                   write_bytes_per_launch=dom["write_bytes_per_launch"], hbm_bytes_per_launch=dom["hbm_bytes_per_launch"],
                   per_kernel={k: out[k] for k in keep if k in out},
                   corrections="FETCH_SIZE x2 (gfx950 reports one half for 16 B/lane streaming reads), KB -> bytes; WRITE_SIZE as reported",
                   source="tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over bench.py --t 2")
        print(json.dumps(row))
        json.dump(row, open(os.path.join(d, "row.json"), "w"), indent=1)
        # fold it into profiles/pmc_traffic.json when asked to (the row of the same workload / batch / precision moves to `history`)
        if os.environ.get("PMC_TRAFFIC_UPDATE"):
            tab_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
            tab = json.load(open(tab_path))
            key = lambda r: (r.get("workload"), r.get("per_gpu_batch"), r.get("precision"))
            old = [r for r in tab["rows"] if key(r) == key(row)]
            tab["rows"] = [r for r in tab["rows"] if key(r) != key(row)] + [dict(row, note=os.environ["PMC_TRAFFIC_UPDATE"])]
            tab.setdefault("history", []).extend(old)
            json.dump(tab, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    gn = out.get("gn_apply_h16")
    if gn:          # the second kernel of the step
        json.dump(dict(kernel="gn_apply_h16", **gn), open(os.path.join(d, "row_gn.json"), "w"), indent=1)
    for k, v in list(out.items())[:12]:
        print(f"{k:22s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch'] or 0:.3e} write {v['write_bytes_per_launch'] or 0:.3e} "
              f"B/launch, {v['avg_ms_under_profiler']:.3f} ms, {v['hbm_tb_per_s_under_profiler'] or 0:.2f} TB/s")


if __name__ == "__main__":
    main()
