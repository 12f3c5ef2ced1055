"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/ar_mi355x.h
declares, and the ctypes binding agrees with the header.  No kernel is launched here."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "ar_mi355x.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(ar_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        if args == ["void"]:
            args = []
        out[m.group(2)] = (m.group(1), args)
    return out


@pytest.fixture(scope="module")
def built_lib():
    from auto_round_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for name in ("ar_qdq_int_fwd", "ar_qdq_int_bwd", "ar_qdq_int_bwd_sgd", "ar_sign_sgd", "ar_mse_loss_fwd_bwd",
                 "ar_gather_rows", "ar_pack_int", "ar_qdq_fp4_fwd", "ar_qdq_fp4_bwd_sgd", "ar_pack_fp4",
                 "ar_group_minmax", "ar_group_absmax", "ar_best_loss_update", "ar_fp4_act_bwd", "ar_pack_awq", "ar_search_fp4_scale", "ar_outlier_mse_loss_fwd_bwd"):
        assert name in fns


def test_library_exports_every_declared_symbol(built_lib):
    for name in header_functions():
        assert hasattr(built_lib, name), f"{name} declared in the header but not exported"


def test_binding_matches_header(built_lib):
    from auto_round_amd import _lib

    fns = header_functions()
    assert set(fns) == set(_lib.SIGNATURES), set(fns) ^ set(_lib.SIGNATURES)
    import ctypes

    def ctype_of(arg):
        arg = arg.strip()
        if "*" in arg or arg.startswith("ar_stream_t"):
            return ctypes.c_void_p
        if arg.startswith("int64_t"):
            return ctypes.c_int64
        if arg.startswith("int32_t"):
            return ctypes.c_int32
        if arg.startswith("float"):
            return ctypes.c_float
        if arg.startswith("int"):
            return ctypes.c_int
        raise AssertionError(arg)

    for name, (ret, args) in fns.items():
        want = [ctype_of(a) for a in args]
        got = _lib.SIGNATURES[name][1]
        assert len(want) == len(got), f"{name}: header has {len(want)} args, binding {len(got)}"
        for i, (w, g) in enumerate(zip(want, got)):
            assert w is g, f"{name} arg {i} ({args[i]}): header {w} vs binding {g}"


def test_abi_version_and_error_strings(built_lib):
    from auto_round_amd import _lib

    assert built_lib.ar_abi_version() == _lib.ABI_VERSION
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'ar_mi355x.h')).read()
    assert int(re.search(r'#define AR_ABI_VERSION (\d+)', hdr).group(1)) == _lib.ABI_VERSION      # header, library and binding agree
    assert built_lib.ar_error_string(0) == b"ok"
    assert b"not supported" in built_lib.ar_error_string(-1)
    assert built_lib.ar_mse_workspace_bytes() > 0


def test_no_cpu_fallback():
    """Ops must refuse CPU tensors loudly instead of computing something somewhere else."""
    import torch

    from auto_round_amd import _lib, ops

    w = torch.zeros(128, dtype=torch.bfloat16)
    with pytest.raises(_lib.Mi355xLibraryError):
        ops.group_minmax(w, 128)


def test_product_never_imports_oracle():
    """tier brief (3): nothing under auto_round_amd/ may import or call oracle/."""
    pkg = os.path.join(REPO, "auto_round_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libar_oracle" not in txt, f


def test_every_entry_point_cites_its_reference_interface_and_is_documented():
    """Each declaration in the header is preceded by a comment citing the reference file it replaces (or says it is binding
    hygiene), and INTEGRATION.md's table names every exported symbol."""
    src = open(HEADER).read()
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    for name in header_functions():
        assert name in doc, f"{name} missing from INTEGRATION.md"
        decl = re.search(r"\b(?:int64_t|int|const char\*)\s+" + name + r"\s*\(", src)
        head = src[:decl.start()]
        last_comment = head[head.rfind("/*"):]
        assert re.search(r"\.py|hygiene|version|return code|scratch|workspace", last_comment), f"{name}: no reference citation in its header comment"
