"""The C-ABI shared library loads (no GPU needed) and exports exactly what include/dwamd.h declares; the ctypes
binding covers every declared entry point; the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    hdr = open(os.path.join(ROOT, "include", "dwamd.h")).read()
    return sorted(set(re.findall(r"\bint\s+(dw_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from distil_whisper_amd import build
    lib = build.build()
    h = ctypes.CDLL(lib)
    names = declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(h, n), n
    h.dw_version.restype = ctypes.c_int
    assert h.dw_version() >= 100


def test_binding_covers_header_and_struct_layout():
    from distil_whisper_amd import ops_hip
    assert set(declared()) == set(ops_hip.EXPORTED_SYMBOLS)
    # DwGemm: 7 pointers, 6 int64, 13 int32 (padded to 8), 1 int64; decode fusions: 4 pointers, 2 int64, 5 int32 + float;
    # z_is_gelu_grad int32 (+4 padding); colsum_out pointer
    assert ctypes.sizeof(ops_hip.DwGemm) == 7 * 8 + 6 * 8 + 14 * 4 + 8 + 4 * 8 + 2 * 8 + 6 * 4 + 8 + 8


def test_invalid_arguments_are_rejected_without_touching_the_gpu():
    from distil_whisper_amd import ops_hip
    lib = ops_hip.load_library()
    g = ops_hip.DwGemm()
    assert lib.dw_gemm_bf16(ctypes.byref(g), None) == -1          # null operands
    assert lib.dw_layernorm_fwd(None, 0, None, None, None, None, None, 4, 128, 1e-5, None) == -1
    assert lib.dw_logmel(None, 1, 480000, None, 80, None, None, None, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    from distil_whisper_amd.ops_hip import HipOps
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipOps("cuda:0")


def test_committed_measurements_are_keyed_by_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py quotes profiles/pmc_traffic.json (roofline.traffic) only when the file was measured with the kernel sources
    this tree holds (build.kernels_sha16 over csrc/ + the C header); the hash follows every byte of those files."""
    import argparse
    import importlib
    import json
    from distil_whisper_amd import build
    sha = build.kernels_sha16()
    assert len(sha) == 16 and sha == build.kernels_sha16()
    bench = importlib.import_module("bench")
    args = argparse.Namespace(model="large-v3", mode="full", batch=32)
    path = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles", "pmc_traffic.json")
    with open(path) as f:
        committed = json.load(f)
    assert "kernels_sha16" in committed and "gemm_t256_NN" in committed["classes"]
    if committed["kernels_sha16"] == sha:
        assert bench.pmc_traffic("gemm_t256_NN", args) == committed["classes"]["gemm_t256_NN"]["traffic_bytes_per_launch"]
    monkeypatch.setattr(bench, "_kernels_sha16", lambda: "0" * 16)          # any other kernel sources: not quoted
    assert bench.pmc_traffic("gemm_t256_NN", args) is None
    assert bench.pmc_traffic("gemm_t256_NN", argparse.Namespace(model="small.en", mode="full", batch=32)) is None
