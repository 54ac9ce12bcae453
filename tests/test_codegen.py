"""Generated-code guards (cross-compiled for gfx950, no GPU): the two patterns that cost the round-4 kernels 1-3 % of the step
each until they were read in the ISA -- a pointer that may be LDS or global (generic: every access a flat load behind
vmcnt(0) + lgkmcnt(0)) and scratch reloads inside an MFMA loop (a VMEM load whose vmcnt wait also drains the operand loads
and the stores in flight).  tools/isa_lint.py prints the full census."""
import importlib.util
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _asm(lint, tu):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([HIPCC] + lint.FLAGS + [os.path.join(lint.CSRC, tu + ".hip"), "-o", out], capture_output=True, text=True, cwd=lint.CSRC)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(out).read()


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs the ROCm compiler")
@pytest.mark.parametrize("tu,loops_at_least", [("attention", 3), ("gemm_wp8_nn", 1)])
def test_hot_loops_have_no_flat_accesses_and_no_scratch_traffic(tu, loops_at_least):
    lint = _lint()
    asm = _asm(lint, tu)
    seen = 0
    for name, body in lint.kernels(asm):
        if name.endswith("Li12EEv5AttnP"):       # the 12-wave backward variants (dw_debug_set key 17, off: measured neutral / slower)
            continue
        _, whole = lint.census(body)
        assert whole["flat"] == 0, (name, "flat memory instructions: a pointer that may be LDS or global")
        for a, b, seg in lint.loops(body):
            n, c = lint.census(seg)
            if c["mfma"] > 64 or n > 700:      # (outer regions that enclose a whole tile: prologue / epilogue code, not a hot loop)
                continue
            seen += 1
            assert c["scratch"] == 0, (name, a, b, "scratch traffic inside an MFMA loop")
            assert c["flat"] == 0, (name, a, b)
    assert seen >= loops_at_least, seen
