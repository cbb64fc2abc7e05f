"""Build libdiffpure_hip_ablate.so: the library compiled with -DDP_ABLATE, i.e. WITH the timing-ablation modes of the
convolution kernels (DP_H2_SW_MODE / DP_H2_PP_MODE / DP_H2_DW_MODE: loops with their loads, waits or stores removed -
WRONG RESULTS, they exist to attribute time).  The product library never contains them.
    python tests/probes/build_ablate.py        (hipcc cross-compiles without a GPU; the .so travels with gpurun)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diffpure_amd import build  # noqa: E402

OUT = os.path.join(build.CSRC, "libdiffpure_hip_ablate.so")


def main():
    cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DDP_ABLATE",
           "-o", OUT] + [os.path.join(build.CSRC, s) for s in build.SOURCES]
    subprocess.run(cmd, check=True, cwd=build.CSRC)
    print(OUT)


if __name__ == "__main__":
    main()
