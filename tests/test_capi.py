"""The C-ABI library loads and exports every symbol include/diffpure_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "diffpure_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from diffpure_amd import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/diffpure_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.dp_abi_version() == 8


def test_ctypes_table_matches_the_header_prototypes():
    """Every prototype of include/diffpure_hip.h against diffpure_amd/_lib.py::SIGNATURES: the same number of parameters, and
    per parameter the same class (pointer / float / 32-bit int / 64-bit int) - the ABI is bound by hand on the Python side, an
    added or re-ordered parameter must not go unnoticed until a kernel reads garbage."""
    import ctypes as C
    from diffpure_amd import _lib
    text = open(os.path.join(ROOT, "include", "diffpure_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text))
    assert sorted(protos) == sorted(_lib.SIGNATURES)

    def klass_c(param):
        param = " ".join(param.split())
        if param in ("void", ""):
            return None
        if "*" in param:
            return "ptr"
        base = param.rsplit(" ", 1)[0] if " " in param else param
        if base in ("float",):
            return "f32"
        if base in ("double",):
            return "f64"
        if base in ("long long", "unsigned long long", "int64_t", "uint64_t", "size_t"):
            return "i64"
        if base in ("int", "unsigned", "unsigned int", "int32_t", "uint32_t"):
            return "i32"
        raise AssertionError(f"unclassified C parameter {param!r}")

    klass_py = {C.c_void_p: "ptr", C.c_char_p: "ptr", C.c_float: "f32", C.c_double: "f64", C.c_int: "i32", C.c_uint: "i32",
                C.c_longlong: "i64", C.c_ulonglong: "i64"}
    for name, plist in protos.items():
        want = [k for k in (klass_c(q) for q in plist.split(",")) if k is not None]
        got = [klass_py[t] for t in _lib.SIGNATURES[name]]
        assert got == want, f"{name}: header {want} vs ctypes {got}"


def test_every_ctypes_call_site_passes_the_declared_number_of_arguments():
    """The GPU operators are stood in for on the CPU (tests/refops.py), so a call site of the C ABI that was not updated with a
    changed prototype would only fail on the GPU box (round 3: two dp_gn_apply call sites, found there).  Count the arguments of
    every `_lib.call("dp_...", ...)` in the package against the ctypes table (call sites that splat a tuple are counted with the
    tuple's length where it is a literal in the same function)."""
    import ast
    from diffpure_amd import _lib
    checked = 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, "diffpure_amd")):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
                tuples = {}       # name -> set of literal tuple lengths assigned in this function
                for n in ast.walk(fn):
                    if isinstance(n, ast.Assign):
                        tgts, val = n.targets, n.value
                        if len(tgts) == 1 and isinstance(tgts[0], ast.Tuple) and isinstance(val, ast.Tuple) and len(tgts[0].elts) == len(val.elts):
                            pairs = zip(tgts[0].elts, val.elts)
                        else:
                            pairs = [(t, val) for t in tgts]
                        for t, v in pairs:
                            if isinstance(t, ast.Name) and isinstance(v, ast.Tuple):
                                tuples.setdefault(t.id, set()).add(len(v.elts))
                for n in ast.walk(fn):
                    if not (isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "call"
                            and isinstance(n.func.value, ast.Name) and n.func.value.id == "_lib" and n.args
                            and isinstance(n.args[0], ast.Constant) and isinstance(n.args[0].value, str)):
                        continue
                    name, want = n.args[0].value, len(_lib.SIGNATURES[n.args[0].value])
                    fixed = sum(1 for a in n.args[1:] if not isinstance(a, ast.Starred))
                    star = [a for a in n.args[1:] if isinstance(a, ast.Starred)]
                    if not star:
                        assert fixed == want, f"{f}:{n.lineno}: {name} gets {fixed} arguments, the prototype has {want}"
                        checked += 1
                    elif all(isinstance(a.value, ast.Name) and len(tuples.get(a.value.id, ())) == 1 for a in star):
                        got = fixed + sum(next(iter(tuples[a.value.id])) for a in star)
                        assert got == want, f"{f}:{n.lineno}: {name} gets {got} arguments, the prototype has {want}"
                        checked += 1
    assert checked >= 30, checked


def test_product_library_carries_no_timing_ablation_modes():
    """The DP_H2_*_MODE timing ablations (kernels with their loads, waits or stores removed: wrong results, they exist to attribute
    time) are compiled only with -DDP_ABLATE into libdiffpure_hip_ablate.so (tests/probes/build_ablate.py); round 2 shipped them
    in the product library behind environment variables.  Neither the variable names nor getenv-per-call survive in the product."""
    from diffpure_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"_MODE" not in blob
    for f in os.listdir(os.path.join(ROOT, "diffpure_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cpp")):
            src = open(os.path.join(ROOT, "diffpure_amd", "csrc", f)).read()
            for m in re.finditer(r"getenv\(", src):
                head = src[:m.start()]
                in_ablate = head.rfind("#ifdef DP_ABLATE") > head.rfind("#endif")
                assert in_ablate or f == "elementwise.hip", f"{f}: getenv outside the once-only tuning table / a DP_ABLATE block"


def test_argument_validation_reports_errors_without_a_gpu():
    from diffpure_amd import _lib
    lib = _lib.load()
    # null pointers are rejected before any launch
    rc = lib.dp_silu(None, None, 4, None)
    assert rc != 0 and b"dp_silu" in lib.dp_last_error()
    rc = lib.dp_conv2d_nhwc(None, 4, None, 0, 1, 1, 1, 3, 3, None, 4, 4, None, None, 0, None, 0, 1.0, None, 4, 0, None, None, None)
    assert rc != 0 and b"null" in lib.dp_last_error()


def test_product_does_not_import_oracle_or_reference():
    bad = re.compile(r"^\s*(from|import)\s+(oracle|refops)\b|/root/reference/|sys\.path.*reference", re.M)
    for pkg in ("diffpure_amd", "runners"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("#", '"""')))
                    m = re.search(r"^\s*(from|import)\s+(oracle|refops|tests)\b", code, re.M)
                    assert m is None, f"{pkg}/{f} imports test infrastructure: {m.group(0)}"


def test_hot_kernels_do_not_spill_registers(tmp_path):
    """The ping-pong convolution runs at exactly 256 VGPRs per lane; an innocent edit of its epilogue once made the
    register allocator spill ~110 values there (-5 % on the whole kernel, invisible in any functional test).  Compile the
    hot kernels to assembly and read the spill counts from the kernel metadata."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "diffpure_amd", "csrc")
    for src, must in (("igemm_h2_pp.hip", ("conv_igemm_h2_ppILi256ELi256ELi0E", "conv_igemm_h2_ppILi512ELi128ELi0E")),
                      ("igemm_h2.hip", ("conv_igemm_h2ILi128ELi128ELi32ELi0E", "conv_igemm_h2ILi64ELi64ELi32ELi0E")),
                      ("igemm_h2_sw.hip", ("conv_igemm_swILi0ELi256E", "conv_igemm_swILi0ELi128E")),
                      ("igemm_h2_dw.hip", ("conv_igemm_dwILi0E",)),
                      ("igemm_h2_dh.hip", ("conv_igemm_dh",)),
                      ("attention.hip", ("attn_flash_kernelILi4E", "attn_flash_kernelILi2E"))):
        out = tmp_path / (src + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                        "-o", str(out), os.path.join(csrc, src)], check=True, capture_output=True)
        text = out.read_text()
        for name in must:       # EVERY instantiation of the kernel (operand formats, pass counts)
            found = re.findall(r"\.name:\s+(\S*" + re.escape(name) + r"\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
            assert found, f"{name} not found in the metadata of {src}"
            for full, spills in found:
                if src == "igemm_h2_sw.hip":
                    # the 512-register kernel (256 accumulators): since the fp16-residual branch joined its epilogue (round 4) the
                    # allocator parks ONE accumulator tile (16 registers) in scratch across the epilogue's head - stored once, read
                    # once, after the k-loop.  Tolerated there, and only there: nothing before the last barrier may touch scratch.
                    assert int(spills) <= 16, f"{full} spills {spills} VGPRs"
                    body = text[text.index(full + ":"):]
                    body = body[:body.index("s_endpgm")]
                    assert "scratch_" not in body[:body.rindex("s_barrier")], f"{full} touches scratch inside its k-loop"
                else:
                    assert int(spills) == 0, f"{full} spills {spills} VGPRs"


def test_elementwise_and_norm_kernels_use_no_scratch(tmp_path):
    """The GroupNorm forward / backward kernels carry the rarely used upfirdn2d stencils of `fir: True` networks.  Compiled into
    the common instantiations they once cost every network 196 bytes of scratch per lane in gn_bwd_apply (2x its time) and 45
    registers in the apply kernels; they are template-instantiated apart now, and NO kernel of these files may touch scratch."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "diffpure_amd", "csrc")
    for src in ("norm.hip", "norm_bwd.hip", "elementwise.hip"):
        out = tmp_path / (src + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                        "-o", str(out), os.path.join(csrc, src)], check=True, capture_output=True)
        found = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", out.read_text())
        assert found, src
        for name, scratch, vgprs in found:
            assert int(scratch) == 0, f"{name} uses {scratch} bytes of scratch per lane"
            if "gn_apply_h2q_kernelILb1ELi2ELb0E" in name:      # round 3's headline GroupNorm-apply: five resident waves per SIMD
                assert int(vgprs) <= 102, (name, vgprs)
            if "gn_apply_h16_kernel" in name:                   # round 4's (fp16 residual stream): at least four
                assert int(vgprs) <= 128, (name, vgprs)


def test_torch_dispatcher_registration():
    """diffpure_amd.torch_ops loads csrc/libdiffpure_torch.so, whose TORCH_LIBRARY block registers the hot operators as
    torch.ops.diffpure_hip.* for the CUDA (= HIP) key only: they resolve, carry schemas, and refuse CPU tensors (no CPU
    kernel exists behind them)."""
    import torch
    from diffpure_amd import torch_ops
    for name in torch_ops.OPERATORS:
        op = getattr(torch.ops.diffpure_hip, name)
        assert "diffpure_hip::" + name in str(op.default._schema)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.diffpure_hip.attention(torch.zeros(1, 64, 3 * 64), 1, True)


def test_split_k_factor_rules():
    """dp_conv2d_nhwc_h2_workspace is a pure host function; the split-K factor it implies is bytes / (B*H*W*N*4).
    DIFFPURE_BATCH_INVARIANT=1 (rounds 1-5): a function of the layer shape only - what keeps results identical for ANY sharding of a
    batch - and only the <= 64-pixel levels are split.  Default (round 6): few-tile / long-K launches are also split per (layer shape,
    batch bucket): never below the shape rule, never more than 8 parts, >= 36 k-tiles per part, none once the launch has 128 tiles of
    128 x 256 - so the reference's per-GPU batch of 4 fills the chip, and a fixed per-GPU batch (weak scaling) always takes one rule."""
    import ctypes as C
    from diffpure_amd import _lib
    lib = _lib.load()
    factor = lambda b, h, w, ks, c, n: lib.dp_conv2d_nhwc_h2_workspace(b, h, w, ks, c, n) // (b * h * w * n * 4)
    old = C.c_int(0)
    assert lib.dp_get_tuning(b"DIFFPURE_BATCH_INVARIANT", C.addressof(old)) == 0
    try:
        assert lib.dp_set_tuning(b"DIFFPURE_BATCH_INVARIANT", 1) == 0
        for (h, w, ks, c, n) in [(4, 4, 3, 256, 256), (8, 8, 3, 512, 256), (8, 8, 3, 1024, 1024), (2, 2, 3, 128, 128), (8, 8, 1, 1024, 768)]:
            factors = {factor(b, h, w, ks, c, n) for b in (1, 2, 7, 64, 256, 1000)}
            assert len(factors) == 1 and factors.pop() in (2, 4), (h, w, ks, c, n)
            assert lib.dp_conv2d_nhwc_h2_splits_by_shape(h, w, ks, c, n) == 1
        for (h, w, ks, c, n) in [(16, 16, 3, 256, 256), (32, 32, 3, 128, 128), (256, 256, 3, 256, 256), (8, 8, 1, 256, 768), (9, 9, 3, 512, 256)]:
            assert all(lib.dp_conv2d_nhwc_h2_workspace(b, h, w, ks, c, n) == 0 for b in (1, 16, 256)), (h, w, ks, c, n)
            assert lib.dp_conv2d_nhwc_h2_splits_by_shape(h, w, ks, c, n) == 0
        assert lib.dp_set_tuning(b"DIFFPURE_BATCH_INVARIANT", 0) == 0
        # the guided UNet's middle levels at the reference's per-GPU batch of 4 / at 8 / at the benchmark's 64
        for (h, c, n) in [(32, 512, 512), (32, 1024, 512), (16, 1024, 1024), (16, 2048, 1024), (8, 1024, 1024)]:
            f4, f8, f64 = (factor(b, h, h, 3, c, n) for b in (4, 8, 64))
            nt = 9 * c // 32
            assert f4 in (2, 4, 8) and f4 >= f8 >= max(f64, 1), (h, c, n, f4, f8, f64)
            assert nt // f4 >= 36 and (4 * h * h // 128) * (n // 256) * f4 <= 384, (h, c, n, f4)
            assert f64 == (2 if h == 8 else 0), (h, c, n, f64)                     # B = 64: only the shape rule's 8 x 8 level
        for b in (1, 4, 64):        # layers that fill the chip, 1x1 layers and shapes the 128 x 256 tiles do not take: never split
            assert factor(b, 64, 64, 3, 512, 512) == 0 or b == 1
            assert lib.dp_conv2d_nhwc_h2_workspace(b, 32, 32, 1, 1024, 512) == 0 and lib.dp_conv2d_nhwc_h2_workspace(b, 32, 32, 3, 512, 128) == 0
        # NCSN++ at the benchmark batches: exactly the shape rule (the CIFAR numbers of rounds 2-5 are unchanged by the new rule)
        for b in (128, 256):
            assert factor(b, 8, 8, 3, 256, 256) == 2 and factor(b, 4, 4, 3, 256, 256) == 4 and factor(b, 4, 4, 3, 512, 256) == 4
            assert lib.dp_conv2d_nhwc_h2_workspace(b, 16, 16, 3, 256, 256) == 0 and lib.dp_conv2d_nhwc_h2_workspace(b, 32, 32, 3, 128, 128) == 0
    finally:
        lib.dp_set_tuning(b"DIFFPURE_BATCH_INVARIANT", old.value)
