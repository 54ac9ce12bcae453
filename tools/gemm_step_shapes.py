"""The step's dominant GEMM launches (distil-large-v3, B = 32: M = 48 000 encoder rows) through the DEFAULT dispatch, a few
launches each with the activation row pitches of the engine: target of the `rocprofv3 --pmc` passes behind
profiles/r6_gemm_pmc_table.md (tools/gemm_pmc.py turns the results into the table; SHAPES env selects a subset)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, D, F, PAD = 48000, 1280, 5120, 64
dev = "cuda"


def act(rows, cols, dtype=torch.bfloat16, pad=PAD):
    return torch.randn(rows, cols + pad, device=dev).to(dtype)[:, :cols]


w = {"qkv": (torch.randn(3 * D, D, device=dev) * 0.02).bfloat16(), "o": (torch.randn(D, D, device=dev) * 0.02).bfloat16(),
     "fc1": (torch.randn(F, D, device=dev) * 0.02).bfloat16(), "fc2": (torch.randn(D, F, device=dev) * 0.02).bfloat16()}
b = {n: torch.randn(n, device=dev) * 0.02 for n in (D, 3 * D, F)}
h, a, x32 = act(M, D), act(M, F), act(M, D, torch.float32, 32)
dy, dz, dqkv = act(M, D), act(M, F), act(M, 3 * D)
zg = act(M, F, torch.float16)
gw = torch.zeros(F, D, device=dev)
shapes = {
    "fwd qkv  NN n3840 k1280 bias": lambda: ops.gemm(h, w["qkv"], bias=b[3 * D]),
    "fwd fc1  NN n5120 k1280 bias gelu + stored gelu'": lambda: ops.gemm(h, w["fc1"], bias=b[F], act=1, want_z="grad", z_row_pad=PAD),
    "fwd fc2  NN n1280 k5120 bias + fp32 residual": lambda: ops.gemm(a, w["fc2"], bias=b[D], residual=x32, round_res=True, out_dtype=torch.float32, out_row_pad=32),
    "fwd out  NN n1280 k1280 bias + fp32 residual": lambda: ops.gemm(h, w["o"], bias=b[D], residual=x32, round_res=True, out_dtype=torch.float32, out_row_pad=32),
    "dX  fc2  NT n5120 k1280 x gelu'": lambda: ops.gemm(dy, w["fc2"], trans_b=True, zgrad=zg),
    "dX  fc1  NT n1280 k5120": lambda: ops.gemm(dz, w["fc1"], trans_b=True, out_row_pad=PAD),
    "dX  qkv  NT n1280 k3840": lambda: ops.gemm(dqkv, w["qkv"], trans_b=True, out_row_pad=PAD),
    "dW  fc1  TT m5120 n1280 k48000 (split-K slabs)": lambda: ops.gemm(dz, h, trans_a=True, trans_b=True, out_dtype=torch.float32, out=gw, atomic_acc=True),
}
sel = os.environ.get("SHAPES")
for name, fn in shapes.items():
    if sel and sel not in name:
        continue
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print("ran", name, flush=True)
