"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol
include/diffpool_hip.h declares, and the ctypes binding agrees with the header on every signature.
No compute call is made."""
import ctypes as C
import os
import re

import pytest

from graph_pooling_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diffpool_hip.h")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \{.*?\} \w+;", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.strip().startswith("#"))
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(dp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        out[name] = (ret, arglist)
    return out


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    fns = _header_functions()
    assert len(fns) >= 30
    for name in fns:
        assert hasattr(lib, name), f"{name} declared in diffpool_hip.h but not exported by the .so"
    assert set(fns) == set(_lib.EXPORTED_SYMBOLS), set(fns) ^ set(_lib.EXPORTED_SYMBOLS)


def _kind(ctype_decl: str) -> str:
    d = ctype_decl.replace("const ", "").strip()
    if "*" in d:
        return "ptr"
    base = d.split()[0] if " " in d else d
    first = " ".join(d.split()[:-1]) if len(d.split()) > 1 else d
    for k in ("size_t", "long long", "long", "int", "float"):
        if first == k:
            return k
    return base


def test_ctypes_signatures_match_header(lib):
    fns = _header_functions()
    cmap = {C.c_int: "int", C.c_long: "long", C.c_float: "float", C.c_size_t: "size_t", C.c_void_p: "ptr",
            C.c_char_p: "ptr"}
    for name, (ret, args) in fns.items():
        res, argtypes = _lib._PROTOS[name]
        assert len(args) == len(argtypes), f"{name}: header has {len(args)} args, binding {len(argtypes)}"
        for i, (decl, ct) in enumerate(zip(args, argtypes)):
            kind = _kind(decl)
            got = "ptr" if (ct not in cmap) else cmap[ct]
            assert kind == got, f"{name} arg {i} ({decl!r}): header {kind}, binding {got}"


def test_version_and_error_string(lib):
    assert lib.dp_version() == 100
    assert isinstance(lib.dp_last_error_string(), bytes)
    assert lib.dp_sizeof_encoder_cfg() == C.sizeof(_lib.EncoderCfg)


def test_device_error_word_decode(lib):
    """diffpool_hip.h "Device-side failures": the text for every mask of the per-device error word, and a quiet read on
    a host without a GPU (no word can be set up: reads as 0, never crashes)."""
    assert lib.dp_device_error(0) == 0 and lib.dp_device_error(1) == 0
    none, bar = lib.dp_device_error_describe(0), lib.dp_device_error_describe(_lib.DEVERR_BARRIER)
    nonf, both = lib.dp_device_error_describe(_lib.DEVERR_NONFINITE_GRAD), lib.dp_device_error_describe(3)
    assert none == b"no device error"
    assert b"grid barrier gave up" in bar and b"DP_NO_LEVEL_FUSION" in bar and b"non-finite" not in bar
    assert b"non-finite gradient norm" in nonf and b"skipped the update" in nonf and b"barrier" not in nonf
    assert b"grid barrier gave up" in both and b"non-finite" in both
    assert b"unknown" in lib.dp_device_error_describe(64)


def test_argument_errors_are_reported_before_any_launch(lib):
    # NULL pointers / bad dims must come back as DP_ERR_INVALID_ARG with a message (no GPU needed)
    rc = lib.dp_bgemm_f32(None, None, None, None, 1, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 1.0, 0.0, 0, None)
    assert rc == -1 and b"NULL" in lib.dp_last_error_string()
    rc = lib.dp_masked_max_fwd(1, 4, None, 1, 4, 1, 0, 3, 4, None)
    assert rc == -1 and b"B=0" in lib.dp_last_error_string()
    cfg = _lib.EncoderCfg()
    assert lib.dp_encoder_save_bytes(C.byref(cfg)) == 0          # invalid cfg -> 0 + message
    assert b"must be positive" in lib.dp_last_error_string()


def test_workspace_queries_run_without_gpu(lib):
    assert lib.dp_gcn_layer_workspace_bytes(4, 16, 3, 8) > 0
    # the link loss is tile-fused: its workspace is per-tile loss partials and [splits,B,N,K] gradient partials,
    # never a [B,N,N] temporary
    assert 0 < lib.dp_linkpred_workspace_bytes(20, 500, 50) < 20 * 500 * 500 * 4 // 2
    assert lib.dp_linkpred_workspace_bytes(256, 1024, 256) < 1 << 20          # big batches need no split
    assert lib.dp_bn_node_workspace_bytes(4, 16, 8) > 0


def test_product_has_no_cpu_fallback():
    """The modules must refuse CPU tensors instead of silently computing something else."""
    import torch
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    m = SoftPoolingGcnEncoder(16, 3, 8, 8, 2, 3, 8, assign_ratio=0.25, linkpred=False)
    x = torch.zeros(2, 16, 3)
    adj = torch.zeros(2, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        m(x, adj, [3, 4], assign_x=x)


def test_nothing_in_the_product_imports_the_oracle():
    pkg = os.path.join(ROOT, "graph_pooling_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                for line in open(os.path.join(dirpath, f)).read().splitlines():
                    assert not re.match(r"\s*(from|import)\s+\.*oracle", line), (f, line)
                    assert "diffpool_oracle" not in line, (f, line)
