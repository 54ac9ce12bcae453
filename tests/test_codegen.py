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


_ASM_CACHE = {}


def _asm(lint, tu):
    if tu not in _ASM_CACHE:
        _ASM_CACHE[tu] = _compile(lint, tu)
    return _ASM_CACHE[tu]


def _compile(lint, tu):
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


# Whole-kernel scratch budgets of the kernels the default dispatch launches (round-4 review: the 320-row kernels sat on a register
# cliff -- 100 B of scratch, 163 scratch instructions, one reload behind a vmcnt(0) per row group of the fp32-residual walk; a
# compiler bump could cost 8 % of the step unnoticed).  Since round 5 the epilogue re-derives its lane addresses per tile
# (gemm_common.h DW_EPI_LAUNDER) and the budgets are: nothing for the 256-row kernels and the attention kernels, 16 B / 6
# instructions (the run-time-flavour walk's end and the kernel exit) for the 320-row kernels.
SCRATCH_BUDGET = [
    ("attention", r"attn_(fwd_kernel<(true|false), 4, 0>|bwd_dkv_kernel<true, false, 2, 4>)", 0, 0),
    # (three waves per SIMD, 168 registers: values parked between kernel entry, the two copies of the tile loop -- one per mask
    # mode since round 5 -- and the output stores; NEVER inside a loop: test_hot_loops_have_no_flat_accesses_and_no_scratch_traffic)
    ("attention", r"attn_bwd_dq_kernel<(true|false), (true|false), 4>", 16, 6),
    ("attention", r"attn_bwd_dkv_kernel<false, false, 3, 4>", 40, 24),
    ("gemm_wp8_nn", r"gemm_wp_kernel<false, false, 2, 4, true, 0, 256, 2>", 0, 0),
    ("gemm_wp8_nt", r"gemm_wp_kernel<false, true, 2, 4, true, 0, 256, 2>", 0, 0),
    ("gemm_wp8_m320", r"gemm_wp_kernel<false, (true|false), 2, 4, true, 0, 320, 2>", 12, 4),
    # (round 6: the 128 x 256 tile in a three-stage ring for outputs below two rounds of 256-row tiles; 164 registers)
    ("gemm_wp8_m128", r"gemm_wp_kernel<false, (true|false), 2, 4, true, 0, 128, 3>", 0, 0),
    # (one lane index parked at the top of the epilogue, reloaded in the bf16-output walk only -- the weight-gradient launches store fp32)
    ("gemm_wp16_tt", r"gemm_wp16_kernel<true, true, 256, 0, 4, 0>", 8, 2),
    ("gemm_wp16_nn", r"gemm_wp16_kernel<false, false, 256, 0, 4, 0>", 0, 0),
    ("gemm_wp16_small", r"gemm_wp16_kernel<false, (true|false), 256, 0, 4, 1>", 8, 2),   # (the k-major-B form parks one lane index, like the weight-gradient kernel)
]


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs the ROCm compiler")
@pytest.mark.parametrize("tu,pattern,max_bytes,max_ops", SCRATCH_BUDGET)
def test_default_kernels_stay_within_their_scratch_budget(tu, pattern, max_bytes, max_ops):
    import re
    lint = _lint()
    asm = _asm(lint, tu)
    names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in lint.kernels(asm)), capture_output=True, text=True).stdout.split("\n")
    hit = 0
    for (name, body), pretty in zip(lint.kernels(asm), names):
        if not re.search(pattern, pretty):
            continue
        hit += 1
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", asm[asm.index(".amdhsa_kernel " + name):])
        scratch_bytes = int(m.group(1))
        ops = sum(1 for l in body if l.strip().startswith("scratch_"))
        assert scratch_bytes <= max_bytes and ops <= max_ops, (pretty, scratch_bytes, ops)
    assert hit >= 1, f"no kernel of {tu} matches {pattern}"

@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs the ROCm compiler")
def test_attention_tile_loops_wait_for_vmem_only_at_their_top():
    """Round 5 finding: the compiler's counted waits for ordinary loads issued in front of a tile loop (the stationary operand's
    fragments) stayed inside the loop -- `s_waitcnt vmcnt(3) .. vmcnt(0)` in front of the first MFMAs -- where they also waited for
    the tile prefetch the inline-assembly DMA had just issued.  With the loop-top wait as an instruction the compiler sees
    (common.h wait_vm0_seen) the only VMEM wait of a forward / dK-dV tile loop is that one; a dQ loop has exactly one as well."""
    import re
    lint = _lint()
    asm = _asm(lint, "attention")
    names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in lint.kernels(asm)), capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for (name, body), pretty in zip(lint.kernels(asm), names):
        if not re.search(r"attn_(fwd_kernel<(true|false), 4, 0>|bwd_dkv_kernel<false, false, 3, 4>|bwd_dq_kernel<false, (true|false), 4>)", pretty):
            continue
        for a, b, seg in lint.loops(body):
            n, c = lint.census(seg)
            if c["mfma"] == 0 or c["mfma"] > 64:
                continue
            waits = [l.strip() for l in seg if re.search(r"s_waitcnt.*vmcnt\(", l)]
            seen += 1
            assert len(waits) == 1 and "vmcnt(0)" in waits[0], (pretty, a, b, waits)
    assert seen >= 5, seen
