"""Same-process A/B of the attention kernels between the current build and distil_whisper_amd/libdwamd_base.so (another
commit's build, tools/build_base_lib.sh): rounds interleaved, encoder / decoder-self / cross shapes of the bench step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd import ops_hip
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
_new = ops.lib
_bp = os.path.join(os.path.dirname(ops_hip.LIB_PATH), "libdwamd_base.so")
_base = ops_hip.load_library(_bp) if os.path.exists(_bp) else _new
# tag -> (library, {dw_debug_set key: value}): 16 / 17 = waves per workgroup of the forward / backward kernels, 18 = 1: plain
# workgroup order instead of the XCD-aware one
libs = {"base": (_base, {16: 4, 17: 4, 18: 0}), "new": (_new, {16: 4, 17: 4, 18: 0})}
if os.environ.get("DW_ATTN_ALL_LIBS"):   # every libdwamd_base[N].so next to the library (compiler-flag variants, tools/build_variant_lib.sh); "base" = the current build
    libs = {"base": (_new, {16: 4, 17: 4, 18: 0})}
    for j in ("", "2", "3", "4", "5"):
        pj = _bp.replace("_base.so", f"_base{j}.so")
        if os.path.exists(pj):
            libs[f"lib{j or 1}"] = (ops_hip.load_library(pj), {16: 4, 17: 4, 18: 0})
if os.environ.get("DW_ATTN_STAGE"):      # dw_debug_set key 3 (bit 0: dq, bit 1: dkv 32-bit staging flag, bit 2: dkv at three waves per SIMD)
    libs = {f"stage {v}": (_new, {16: 4, 17: 4, 18: 0, 3: v}) for v in (5, 1, 7, 3)}
    libs["base"] = libs.pop("stage 5")
if os.environ.get("DW_ATTN_VARIANTS"):
    libs.update({"new plain order": (_new, {16: 4, 17: 4, 18: 1}), "new 8 / 12 waves": (_new, {16: 8, 17: 12, 18: 0})})
D, H = 1280, 20
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, B, Lq, Lk, causal in (("encoder self", 32, 1500, 1500, False), ("decoder self", 32, 448, 448, True),
                                ("cross", 32, 448, 1500, False)):
    q = torch.randn(B * Lq, D, device="cuda").bfloat16()
    kv = torch.randn(B * Lk, 2 * D, device="cuda").bfloat16()
    k, v = kv[:, :D], kv[:, D:]
    do = torch.randn(B * Lq, D, device="cuda").bfloat16()
    res, outs = {}, {}
    for rnd in range(3):
        for tag, (lib, keys) in libs.items():
            ops.lib = lib
            for kk, vv in keys.items():
                assert lib.dw_debug_set(kk, vv) == 0
            o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
            g = ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, causal, 0.125)
            outs[tag] = (o, lse) + tuple(g)
            res.setdefault(tag + " fwd", []).append(timed(lambda: ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)))
            res.setdefault(tag + " bwd", []).append(timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, causal, 0.125)))
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    for kk, t in res.items():
        m = sorted(t)[len(t) // 2]
        f = fl * (2.5 if "bwd" in kk else 1.0)
        print(f"{name:13s} {kk:20s} us: " + " ".join(f"{x:.0f}" for x in t) + f"   median {f / m / 1e6:.0f} TFLOP/s")
    for tag in libs:
        if tag == "base": continue
        d = [(a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-30) for a, b in zip(outs[tag], outs["base"])]
        print(f"{name:13s} max |{tag} - base| / max|base| for o, lse, dq, dk, dv: " + " ".join(f"{x:.2e}" for x in d))
