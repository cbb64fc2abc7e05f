"""ctypes binding of libdiffpure_hip.so (include/diffpure_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not resident on a GPU, the
call raises.  The shared object is built in-tree by `diffpure_amd.build.build()` (hipcc,
--offload-arch=gfx950) so that it travels with the repo snapshot.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdiffpure_hip.so")

_f = C.c_float
_i = C.c_int
_ll = C.c_longlong
_ull = C.c_ulonglong
_p = C.c_void_p

# name -> argtypes (restype is always int unless listed in _RESTYPE)
SIGNATURES = {
    "dp_abi_version": [],
    "dp_last_error": [],
    "dp_set_tuning": [C.c_char_p, _i],
    "dp_get_tuning": [C.c_char_p, _p],
    "dp_prof_enable": [_i],
    "dp_prof_collect": [_p, _p, _p, _p, _p],
    "dp_conv2d_nhwc": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _i, _p, _i, _f, _p, _i, _i, _p, _p, _p],
    "dp_conv2d_stem_ok": [_i, _i, _i, _i, _i],
    "dp_conv2d_stem": [_p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _p, _p],
    "dp_gn_finalize_cols": [_p, _i, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p],
    "dp_gemm_strided": [_p, _i, _ll, _ll, _i, _p, _i, _ll, _ll, _i, _p, _i, _ll, _ll, _i, _i, _i, _i, _i, _f, _p],
    "dp_gemm_strided_h16": [_p, _i, _i, _ll, _ll, _i, _p, _i, _i, _ll, _ll, _i, _p, _i, _ll, _ll, _i, _i, _i, _i, _i, _f, _p],
    "dp_gemm_strided_h16_ok": [_i, _i, _i],
    "dp_gn_bwd_stats": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p, _p],
    "dp_gn_bwd_apply": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p],
    "dp_gn_bwd_fused_ok": [_i, _i, _i, _i, _i, _i],
    "dp_gn_bwd_fused": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _f, _p],
    "dp_resample_bwd": [_p, _i, _i, _i, _i, _i, _p, _p, _p],
    "dp_softmax_bwd_rows": [_p, _p, _ll, _i, _p],
    "dp_add": [_p, _p, _p, _ll, _p],
    "dp_attention_fused": [_p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p],
    "dp_resize_affine": [_p, _i, _i, _i, _i, _i, _f, _f, _p, _i, _i, _i, _p],
    "dp_resize_affine_bwd": [_p, _i, _i, _i, _i, _i, _f, _p, _i, _i, _i, _p],
    "dp_softmax_rows": [_p, _ll, _i, _p],
    "dp_gn_stats": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p],
    "dp_gn_finalize": [_p, _i, _i, _i, _ll, _f, _p, _p],
    "dp_gn_apply": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p],
    "dp_gn_apply_h16": [_p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p],
    "dp_conv2d_nhwc_h2": [_p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _f, _p, _i, _p, _p, _p, _ll, _i, _i, _i, _i, _i,
                          _p, _i, _p, _i, _p],
    "dp_conv2d_nhwc_h2_takes_segments": [_i, _i, _i, _i, _i, _i, _i],
    "dp_conv2d_nhwc_h2_splits_by_shape": [_i, _i, _i, _i, _i],
    "dp_conv2d_nhwc_h2_partials": [_p, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _i, _i, _i, _p, _i, _p, _i, _p, _p],
    "dp_splitk_epilogue": [_p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _f, _p, _i, _p, _p, _p],
    "dp_splitk_gn_ok": [_i, _i, _i, _i, _i],
    "dp_splitk_gn": [_p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _f, _p, _i, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p],
    "dp_round_weights": [_p, _p, _ll, _i, _ull, _ll, _p],
    "dp_conv2d_nhwc_h2_workspace": [_i, _i, _i, _i, _i, _i],
    "dp_pack_h2": [_p, _ll, _i, _i, _p, _p],
    "dp_silu": [_p, _p, _ll, _p],
    "dp_axpby": [_p, _f, _p, _f, _p, _ll, _p],
    "dp_timestep_embedding": [_p, _i, _p, _i, _i, _p, _p],
    "dp_em_step": [_p, _p, _i, _i, _i, _i, _f, _f, _f, _i, _f, _f, _f, _p, _ull, _ll, _i, _p, _p],
    "dp_philox_normal": [_p, _i, _ll, _ull, _ll, _i, _p],
    "dp_ddpm_step": [_p, _p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _p, _ull, _ll, _i, _p, _p],
}
_RESTYPE = {"dp_last_error": C.c_char_p, "dp_conv2d_nhwc_h2_workspace": C.c_longlong}

_lib = None


class DiffpureHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle. Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffpureHipError(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback for the purification engine."
        )
    # The library's HIP calls must bind to the SAME HIP runtime torch uses (streams and device
    # pointers are shared).  PyTorch-ROCm bundles its own libamdhip64; load it first so that the
    # dynamic loader reuses it for our NEEDED entry instead of pulling a second copy from /opt/rocm
    # (two runtimes in one process => "no ROCm-capable device is detected" on the first launch).
    import torch

    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise DiffpureHipError(f"{name} failed (rc={rc}): {lib.dp_last_error().decode()}")
