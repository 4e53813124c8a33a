"""The C-ABI library loads on a CPU-only machine and exports every symbol include/gnnb_hip.h
declares; the product path refuses to run without a GPU (no fallback).  No compute calls here."""
import re
import subprocess
from pathlib import Path

import pytest
import torch

from gnnbuilder_amd import runtime

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    if not runtime.LIB_PATH.exists():
        runtime.build_library()  # hipcc cross-compiles gfx950 without a GPU
    return runtime.LIB_PATH


def header_functions():
    text = (ROOT / "include" / "gnnb_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnnb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_runtime_binds():
    assert sorted(runtime.EXPORTED_SYMBOLS) == header_functions()


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib_path)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(gnnb_[a-z0-9_]+)", out))
    missing = [f for f in header_functions() if f not in exported]
    assert not missing, f"libgnnb_hip.so lacks {missing}"


def test_library_has_gfx950_code_object(lib_path):
    blob = lib_path.read_bytes()
    assert b"gfx950" in blob and b"k_aggregate" in blob and b"k_linear" in blob


def test_library_loads_and_reports_version(lib_path):
    lib = runtime.load_library(require_gpu=False)
    assert lib.gnnb_version() == 104
    for sym in runtime.EXPORTED_SYMBOLS:
        assert hasattr(lib, sym)


def test_struct_layout_matches_header():
    # 19 int32/float fields (version 103: + math) + pools[3] = 21 * 4 bytes, in the header's order
    import ctypes
    import re
    assert ctypes.sizeof(runtime.ModelDesc) == 21 * 4
    hdr = (ROOT / "include" / "gnnb_hip.h").read_text()
    body = hdr[hdr.index("typedef struct gnnb_model_desc {"):hdr.index("} gnnb_model_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(?:int32_t|float)\s+(\w+)(?:\[\d+\])?;", body)
    assert names == [f[0] for f in runtime.ModelDesc._fields_]
    assert ctypes.sizeof(runtime.GemmSeg) == 24


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine WITHOUT a GPU")
def test_product_path_fails_loudly_without_gpu(lib_path):
    with pytest.raises(runtime.GnnbUnavailable):
        runtime.load_library(require_gpu=True)
    from helpers import make_model
    with pytest.raises(runtime.GnnbUnavailable):
        runtime.CompiledModel.from_model(make_model("gcn", hidden=16), 4, 64, 128)


def test_invalid_description_is_rejected_on_host(lib_path):
    import ctypes as C
    lib = runtime.load_library(require_gpu=False)
    from helpers import make_model
    d = runtime.make_desc(make_model("gcn", hidden=16).spec())
    assert lib.gnnb_model_num_params(C.byref(d)) == 2 * 2 + 2 * 3
    d.conv_type = 9
    assert lib.gnnb_model_num_params(C.byref(d)) < 0
    assert b"conv_type" in lib.gnnb_last_error()
