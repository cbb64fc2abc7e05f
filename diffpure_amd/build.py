"""Build libdiffpure_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["igemm.hip", "igemm_h2.hip", "igemm_h2_pp.hip", "igemm_h2_sw.hip", "igemm_h2_dw.hip", "igemm_h2_dh.hip", "igemm_h2_nn.hip", "norm.hip", "norm_bwd.hip", "elementwise.hip", "resize.hip", "attention.hip", "gemm_h16.hip", "stem.hip", "boundary.hip"]
OUT = os.path.join(CSRC, "libdiffpure_hip.so")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the HIP kernels")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]   # torch_binding.cpp: see build_torch_binding
    deps.append(os.path.join(HERE, "..", "include", "diffpure_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if force or needs_build():
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    build_torch_binding(force=force, verbose=verbose)
    return OUT


TORCH_SRC = os.path.join(CSRC, "torch_binding.cpp")
TORCH_OUT = os.path.join(CSRC, "libdiffpure_torch.so")


def build_torch_binding(force=False, verbose=False):
    """csrc/torch_binding.cpp (TORCH_LIBRARY registration of the hot operators; host C++ only) -> libdiffpure_torch.so,
    linked against libdiffpure_hip.so (found through $ORIGIN) and the torch libraries of THIS interpreter."""
    deps = [TORCH_SRC, OUT, os.path.join(HERE, "..", "include", "diffpure_hip.h")]
    if not force and os.path.exists(TORCH_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_OUT) for d in deps):
        return TORCH_OUT
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: cannot build the torch operator registration")
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include", "-o", TORCH_OUT, TORCH_SRC, f"-L{CSRC}", "-ldiffpure_hip",
                                                     f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
                                                     "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return TORCH_OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
