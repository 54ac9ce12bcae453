"""Evidence hygiene: every kernel of this library that matters in a committed rocprofv3 summary has a class in
tools/pmc_traffic.py (round 4 added gemm_wp16_kernel without a pattern: a third of the quoted GEMM class and every
weight-gradient GEMM silently lost their counter bytes), and bench.py refuses a traffic figure whose launch count does
not match the step it instruments."""
import glob
import importlib.util
import json
import os
import re
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rows(md):
    for line in open(md):
        m = re.match(r"\| `(.+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            yield m.group(1), int(m.group(2)), float(m.group(7))


def test_every_hot_kernel_symbol_has_a_traffic_class():
    pt = _load(os.path.join(ROOT, "tools", "pmc_traffic.py"), "pmc_traffic_tool")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.md")))
    assert files, "no committed rocprofv3 summaries"
    foreign = ("at::native", "__amd_rocclr", "rccl", "nccl")
    for md in files[-2:]:                       # the two most recent rounds (older rounds had other kernels)
        for name, calls, pct in _rows(md):
            if pct < 1.0 or any(f in name for f in foreign):
                continue
            assert pt.classify(name) is not None, f"{os.path.basename(md)}: {name} ({pct} % of the step) has no class in tools/pmc_traffic.py"


def test_main_loop_kernels_map_to_the_keys_ops_hip_reports():
    pt = _load(os.path.join(ROOT, "tools", "pmc_traffic.py"), "pmc_traffic_tool")
    for ta in (False, True):
        for tb in (False, True):
            key = f"gemm_t256_{'T' if ta else 'N'}{'T' if tb else 'N'}"
            a, b = str(ta).lower(), str(tb).lower()
            assert pt.classify(f"void gemm_wp16_kernel<{a}, {b}, 256, 0, 4, 0>(GemmP)") == key
            assert pt.classify(f"void gemm_wp_kernel<{a}, {b}, 2, 4, true, 0, 320, 2>(GemmP)") == key
    # the small-M rule's kernels (dw_debug_set key 25) report under the keys ops_hip.py gives launches below two rounds of 256-row tiles
    for tb in (False, True):
        key = f"gemm_t128_N{'T' if tb else 'N'}"
        b = str(tb).lower()
        assert pt.classify(f"void gemm_wp_kernel<false, {b}, 2, 4, true, 0, 128, 3>(GemmP)") == key
        assert pt.classify(f"void gemm_wp16_kernel<false, {b}, 256, 0, 4, 1>(GemmP)") == key


def test_bench_refuses_traffic_with_a_wrong_launch_count(tmp_path, monkeypatch):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    sha = bench._kernels_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "pmc_traffic.json").write_text(json.dumps({"kernels_sha16": sha, "steps_profiled": 2, "classes": {
        "gemm_t256_NN": {"traffic_bytes_per_launch": 1.0e9, "launches": 400, "launches_per_step": 200.0}}}))
    monkeypatch.setattr(bench.os.path, "abspath", lambda p: str(tmp_path / "bench.py") if p == bench.__file__ else os.path.realpath(p))
    args = types.SimpleNamespace(model="large-v3", mode="full", batch=32)
    assert bench.pmc_traffic("gemm_t256_NN", args, 200) == 1.0e9
    assert bench.pmc_traffic("gemm_t256_NN", args, 298) is None        # round 4's situation: 98 launches uncounted
    assert bench.pmc_traffic("gemm_t256_TT", args, 145) is None        # class absent
