"""Does the row stride of the operands matter at K = 5120 (fc2 forward: 10 240-byte rows)?  The same GEMM with operand row strides
padded by 64 / 128 / 192 elements, row-major (NN) and k-major B (NT)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 48000
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, N, K in (("fc2 fwd  N=1280 K=5120", 1280, 5120), ("out fwd  N=1280 K=1280", 1280, 1280)):
    for pa, pb in ((0, 0), (64, 0), (64, 64)):
        A = torch.randn(M, K + pa, device="cuda").bfloat16(); a = A[:, :K]
        Bm = (torch.randn(N, K + pb, device="cuda") * 0.03).bfloat16(); b = Bm[:, :K]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ts = sorted(timed(lambda: ops.gemm(a, b, out=out)) for _ in range(3))
        print(f"{name}  pad A {pa:3d} B {pb:3d}: {ts[1]:7.1f} us  {2.0 * M * N * K / ts[1] / 1e6:6.0f} TFLOP/s", flush=True)
print("--- dX (k-major B) and dW (both k-major), operands padded by P elements per row")
for name, Mm, N, K, ta, tb in (("dX fc1  N=1280 K=5120", M, 1280, 5120, False, True), ("dX fc2  N=5120 K=1280", M, 5120, 1280, False, True),
                               ("dX qkv  N=1280 K=3840", M, 1280, 3840, False, True), ("dW fc1  5120x1280 K=48000", 5120, 1280, M, True, True),
                               ("dW fc2  1280x5120 K=48000", 1280, 5120, M, True, True), ("dW qkv  3840x1280 K=48000", 3840, 1280, M, True, True)):
    for P in (0, 64, 128):
        a_shape = (K, Mm) if ta else (Mm, K)
        b_shape = (K, N) if tb else (N, K)
        A = torch.randn(a_shape[0], a_shape[1] + P, device="cuda").bfloat16(); a = A[:, :a_shape[1]]
        Bm = (torch.randn(b_shape[0], b_shape[1] + P, device="cuda") * 0.03).bfloat16(); b = Bm[:, :b_shape[1]]
        if ta:
            out = torch.zeros(Mm, N, device="cuda", dtype=torch.float32)
            fn = lambda: ops.gemm(a, b, trans_a=True, trans_b=True, out=out, atomic_acc=True)
        else:
            O = torch.empty(Mm, N + P, device="cuda", dtype=torch.bfloat16); out = O[:, :N]
            fn = lambda: ops.gemm(a, b, trans_b=tb, out=out)
        ts = sorted(timed(fn) for _ in range(3))
        print(f"{name}  pad {P:3d}: {ts[1]:7.1f} us  {2.0 * Mm * N * K / ts[1] / 1e6:6.0f} TFLOP/s", flush=True)
print("--- attention, encoder shape, q / k / v as column blocks of one [B*L, 3*D + P] buffer, o / dO / dq.. in [B*L, D + P]")
B, H, L, D = 32, 20, 1500, 1280
for P in (0, 64, 128):
    QKV = torch.randn(B * L, 3 * D + P, device="cuda").bfloat16()
    q, k, v = QKV[:, :D], QKV[:, D:2 * D], QKV[:, 2 * D:3 * D]
    o, lse = ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
    DO = torch.randn(B * L, D + P, device="cuda").bfloat16(); do = DO[:, :D]
    tf = sorted(timed(lambda: ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)) for _ in range(3))[1]
    tb_ = sorted(timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, False, 0.125)) for _ in range(3))[1]
    fl = 4.0 * B * H * L * L * 64
    print(f"attention pad {P:3d}: fwd {tf:7.1f} us {fl / tf / 1e6:5.0f} TFLOP/s   bwd {tb_:7.1f} us {2.5 * fl / tb_ / 1e6:5.0f} TFLOP/s", flush=True)
