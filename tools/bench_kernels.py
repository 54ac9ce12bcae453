"""Per-kernel timing on the MI355X (HIP events around repeated launches) -> prints TFLOP/s or GB/s per kernel.
Used during bring-up and tuning; bench.py is the contract benchmark."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ops = HipOps("cuda:0")
    if os.environ.get("DW_PERSIST"):
        ops.lib.dw_debug_set(2, int(os.environ["DW_PERSIST"]))
    res = []
    B = int(os.environ.get("DW_B", "32"))

    def rnd(shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(shape, device="cuda") * scale).to(dtype)

    # ---- GEMM shapes of distil-large-v3 (D=1280) at B=32 ----
    shapes = [] if os.environ.get("DW_QUICK") else [
        ("enc qkv fwd NT", B * 1500, 3840, 1280, False, False),
        ("enc out fwd NT", B * 1500, 1280, 1280, False, False),
        ("enc fc1 fwd NT", B * 1500, 5120, 1280, False, False),
        ("enc fc2 fwd NT", B * 1500, 1280, 5120, False, False),
        ("enc fc1 dX NN", B * 1500, 1280, 5120, False, True),
        ("enc fc2 dX NN", B * 1500, 5120, 1280, False, True),
        ("enc fc1 dW TN", 5120, 1280, B * 1500, True, True),
        ("enc fc2 dW TN", 1280, 5120, B * 1500, True, True),
        ("dec qkv fwd NT", B * 447 + (-B * 447) % 64, 3840, 1280, False, False),
        ("lm head fwd NT", B * 447, 51866, 1280, False, False),
        ("lm head dX NN", B * 447, 1280, 51968, False, True),
        ("lm head dW TN", 51866 + 6, 1280, B * 447 + (-B * 447) % 64, True, True),
        ("conv2 fwd NT", B * 1500, 1280, 3840, False, False),
    ]
    for name, M, N, K, ta, tb in shapes:
        a = rnd((K, M) if ta else (M, K))
        b = rnd((K, N) if tb else (N, K), 0.05)
        for tile in (128, 256):
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            try:
                t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, tile=tile))
                tf = 2.0 * M * N * K / t / 1e12
            except Exception as ex:  # noqa: BLE001
                t, tf = float("nan"), float("nan")
                print("ERR", name, ex)
            res.append({"kernel": f"gemm {name} t{tile}", "M": M, "N": N, "K": K, "ms": t * 1e3, "tflops": tf})
            print(res[-1], flush=True)
        del a, b, out

    # ---- epilogue variants at the encoder shapes (what the step actually runs) ----
    M = B * 1500
    x = rnd((M, 1280)); w1 = rnd((5120, 1280), 0.05); b1 = rnd((5120,), 0.1, torch.float32)
    a_out = torch.empty(M, 5120, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(x, w1, bias=b1, act=1, want_z=True, out=a_out))
    res.append({"kernel": "gemm fc1 bias+gelu+z", "ms": t * 1e3, "tflops": 2.0 * M * 5120 * 1280 / t / 1e12}); print(res[-1], flush=True)
    w2 = rnd((1280, 5120), 0.05); b2 = rnd((1280,), 0.1, torch.float32); resid = rnd((M, 1280), 1.0, torch.float32)
    t = timeit(lambda: ops.gemm(a_out, w2, bias=b2, residual=resid, out_dtype=torch.float32))
    res.append({"kernel": "gemm fc2 bias+residual f32", "ms": t * 1e3, "tflops": 2.0 * M * 5120 * 1280 / t / 1e12}); print(res[-1], flush=True)
    wq = rnd((3840, 1280), 0.05); bq = rnd((3840,), 0.1, torch.float32)
    t = timeit(lambda: ops.gemm(x, wq, bias=bq))
    res.append({"kernel": "gemm qkv bias", "ms": t * 1e3, "tflops": 2.0 * M * 3840 * 1280 / t / 1e12}); print(res[-1], flush=True)
    dy = rnd((M, 1280)); zz = rnd((M, 5120))
    t = timeit(lambda: ops.gemm(dy, w2, trans_b=True, zgrad=zz))
    res.append({"kernel": "gemm fc2 dX *gelu'", "ms": t * 1e3, "tflops": 2.0 * M * 5120 * 1280 / t / 1e12}); print(res[-1], flush=True)
    gw = torch.zeros(5120, 1280, device="cuda")
    t = timeit(lambda: ops.gemm(zz, x, trans_a=True, trans_b=True, out=gw, atomic_acc=True))
    res.append({"kernel": "gemm fc1 dW auto", "ms": t * 1e3, "tflops": 2.0 * M * 5120 * 1280 / t / 1e12}); print(res[-1], flush=True)
    for sk in (2, 3, 5):
        for tile in (128, 256):
            t = timeit(lambda: ops.gemm(zz, x, trans_a=True, trans_b=True, out=gw, atomic_acc=True, tile=tile, split_k=sk))
            res.append({"kernel": f"gemm fc1 dW atomic t{tile} sk{sk}", "ms": t * 1e3, "tflops": 2.0 * M * 5120 * 1280 / t / 1e12}); print(res[-1], flush=True)
    gw = torch.zeros(1280, 1280, device="cuda")
    t = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, out=gw, atomic_acc=True))
    res.append({"kernel": "gemm out dW auto", "ms": t * 1e3, "tflops": 2.0 * M * 1280 * 1280 / t / 1e12}); print(res[-1], flush=True)
    for sk in (5, 10, 20):
        for tile in (128, 256):
            t = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, out=gw, atomic_acc=True, tile=tile, split_k=sk))
            res.append({"kernel": f"gemm out dW atomic t{tile} sk{sk}", "ms": t * 1e3, "tflops": 2.0 * M * 1280 * 1280 / t / 1e12}); print(res[-1], flush=True)
    del x, w1, a_out, w2, resid, wq, dy, zz, gw
    if os.environ.get("DW_QUICK"):
        return

    # ---- attention ----
    H = 20
    for name, Lq, Lk, causal in (("enc self", 1500, 1500, False), ("dec self", 447, 447, True),
                                 ("cross", 447, 1500, False)):
        q, k, v = rnd((B * Lq, H * 64)), rnd((B * Lk, H * 64)), rnd((B * Lk, H * 64))
        t = timeit(lambda: ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125))
        fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
        res.append({"kernel": f"attn fwd {name}", "ms": t * 1e3, "tflops": fl / t / 1e12})
        print(res[-1], flush=True)
        o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
        do = rnd((B * Lq, H * 64))
        t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, causal, 0.125))
        res.append({"kernel": f"attn bwd {name}", "ms": t * 1e3, "tflops": 2.5 * fl / t / 1e12})
        print(res[-1], flush=True)

    # ---- HBM-bound kernels ----
    rows, D = B * 1500, 1280
    x = rnd((rows, D), 1.0, torch.float32)
    g, bt = rnd((D,), 1.0, torch.float32), rnd((D,), 1.0, torch.float32)
    t = timeit(lambda: ops.layernorm_fwd(x, g, bt))
    res.append({"kernel": "layernorm fwd f32", "ms": t * 1e3, "gbps": rows * D * 6 / t / 1e9})
    print(res[-1], flush=True)
    y, mu, rs = ops.layernorm_fwd(x, g, bt)
    dres = torch.zeros_like(x)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    t = timeit(lambda: ops.layernorm_bwd(y, x, mu, rs, g, dres, dg, db))
    res.append({"kernel": "layernorm bwd f32", "ms": t * 1e3, "gbps": rows * D * 14 / t / 1e9})
    print(res[-1], flush=True)
    del x, y, dres

    rowsL, V, ld = B * 447, 51866, 51968
    s, tl = rnd((rowsL, ld)), rnd((rowsL, ld))
    labels = torch.randint(0, V, (rowsL,), device="cuda")
    t = timeit(lambda: ops.distill_loss(s, tl, labels, V, 2.0, 0.8, 1.0, 1.0, True), iters=5)
    res.append({"kernel": "distill loss+grad", "ms": t * 1e3, "gbps": rowsL * V * 2 * 3 / t / 1e9})
    print(res[-1], flush=True)
    del s, tl

    audio = torch.randn(B, 480000, device="cuda") * 0.1
    from transformers.audio_utils import mel_filter_bank
    filt = torch.tensor(mel_filter_bank(201, 128, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney"),
                        dtype=torch.float32).cuda().contiguous()
    t = timeit(lambda: ops.logmel(audio, filt), iters=5)
    res.append({"kernel": "logmel 128", "ms": t * 1e3, "gbps": B * (480000 * 4 + 128 * 3000 * 4) / t / 1e9,
                "audio_s_per_s": B * 30 / t})
    print(res[-1], flush=True)

    n = 200_000_000
    p, gr, m, v = (torch.zeros(n, device="cuda") for _ in range(4))
    sh = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    ss = torch.ones(1, device="cuda")
    t = timeit(lambda: ops.adamw(p, gr, m, v, sh, ss, 1.0, 1.0, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1), iters=5)
    res.append({"kernel": "adamw", "ms": t * 1e3, "gbps": n * 30 / t / 1e9})
    print(res[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
