"""Wall time of one UNet step, eager launches vs the HIP-graph step (DIFFPURE_GRAPH), at small batches (tuning aid).
    python tests/probes/graph_vs_eager.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for wl, batches in (("cifar32_ncsnpp", (4, 16, 64)), ("imagenet256_guided", (1, 4))):
    for b in batches:
        row = []
        for g in ("0", "1"):
            env = dict(os.environ, DIFFPURE_GRAPH=g)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--batch", str(b), "--dt", "1e-2",
                                  "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-conv-profile"],
                                 env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            row.append(json.loads(out)["ms_per_step"] / 10)
        print(f"{wl} B={b}: eager {row[0]:.2f} ms/step, graph {row[1]:.2f} ms/step ({row[0] / row[1]:.2f}x)", flush=True)
