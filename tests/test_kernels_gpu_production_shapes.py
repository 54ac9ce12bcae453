"""The hot kernels at the BENCHMARK's shapes (distil-large-v3 / large-v3, per-GPU batch 32: 48 000 encoder rows, 14 304
decoder rows, d_model 1280, FFN 5120, 20 heads, 1500 x 1500 attention, vocabulary 51 866) against the torch restatement of
oracle/ref_ops.py on the same device tensors -- every GEMM flavour the step launches with the kernel the dispatcher picks
for it (320-row / 256-row tiles, flavoured epilogue walks, split-K slabs), attention forward and backward for the three
shape classes at a batch the fp32 restatement holds, LayerNorm forward / backward, the fused loss over the full
vocabulary.  tests/test_kernels_gpu.py covers the same kernels on small ragged shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
M_ENC, M_DEC, D, F, H, V = 48000, 14304, 1280, 5120, 20, 51866


@pytest.fixture(scope="module")
def ops():
    from distil_whisper_amd.ops_hip import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_ops import RefOps
    return RefOps("cuda:0")


def rnd(shape, scale=1.0, dtype=torch.bfloat16, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda", dtype=torch.float32) * scale).to(dtype)


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _case(ops, ref, name, M, N, K, tb, kw, tol=6e-3):
    a = rnd((M, K), 1.0, seed=1)
    b = rnd((K, N) if tb else (N, K), 0.03, seed=2)
    got = ops.gemm(a, b, trans_b=tb, **kw)
    want = ref.gemm(a, b, trans_b=tb, **{k: (v.clone() if k == "colsum" else v) for k, v in kw.items()})
    if isinstance(got, tuple):
        assert relerr(got[1], want[1]) < 2e-3, (name, "second output", relerr(got[1], want[1]))
        got, want = got[0], want[0]
    e = relerr(got, want)
    assert e < tol, (name, e)
    del a, b, got, want
    torch.cuda.empty_cache()


def test_forward_and_dgrad_gemms_of_the_step(ops, ref):
    bias_d, bias_f, bias_3d = rnd((D,), 0.1, torch.float32, 3), rnd((F,), 0.1, torch.float32, 4), rnd((3 * D,), 0.1, torch.float32, 5)
    r32, r16 = rnd((M_ENC, D), 1.0, torch.float32, 6), rnd((M_ENC, D), 1.0, torch.bfloat16, 7)
    zg = rnd((M_ENC, F), 0.5, torch.float16, 8)
    cs = torch.zeros(F, device="cuda")
    for name, M, N, K, tb, kw in (
            ("qkv", M_ENC, 3 * D, D, False, dict(bias=bias_3d)),
            ("out-proj student", M_ENC, D, D, False, dict(bias=bias_d, residual=r32, out_dtype=torch.float32)),
            ("out-proj teacher", M_ENC, D, D, False, dict(bias=bias_d, residual=r16)),
            ("fc1 student", M_ENC, F, D, False, dict(bias=bias_f, act=1, want_z="grad")),
            ("fc1 teacher", M_ENC, F, D, False, dict(bias=bias_f, act=1)),
            ("fc2 student", M_ENC, D, F, False, dict(bias=bias_d, residual=r32, out_dtype=torch.float32)),
            ("fc2 teacher", M_ENC, D, F, False, dict(bias=bias_d, residual=r16)),
            ("dX fc2 + fc1.bias sums", M_ENC, F, D, True, dict(zgrad=zg, colsum=cs)),
            ("dX fc1", M_ENC, D, F, True, {}), ("dX qkv", M_ENC, D, 3 * D, True, {}), ("dX out", M_ENC, D, D, True, {}),
            ("decoder qkv", M_DEC, 3 * D, D, False, dict(bias=bias_3d)),
            ("decoder out-proj", M_DEC, D, D, False, dict(bias=bias_d, residual=r32[:M_DEC], out_dtype=torch.float32)),
            ("teacher decoder fc1, padded rows", 14400, F, D, False, dict(bias=bias_f, act=1))):
        _case(ops, ref, name, M, N, K, tb, kw)
    # the column sums the dX-fc2 GEMM added = column sums of its own output
    a, b = rnd((M_ENC, D), 1.0, seed=1), rnd((D, F), 0.03, seed=2)
    cs2 = torch.zeros(F, device="cuda")
    out = ops.gemm(a, b, trans_b=True, zgrad=zg, colsum=cs2)
    assert relerr(cs2, out.float().sum(0)) < 2e-4 and relerr(cs, cs2) < 1e-5


def test_lm_head_and_weight_gradient_gemms(ops, ref):
    ldv = (V + 63) // 64 * 64
    hf = rnd((M_DEC, D), 1.0, seed=11)
    e_pad = rnd((ldv, D), 0.03, seed=12)
    logits = ops.gemm(hf, e_pad)
    assert relerr(logits, ref.gemm(hf, e_pad)) < 6e-3
    dl = rnd((M_DEC + 32, ldv), 0.01, seed=13)                       # rows padded to a multiple of 64 (K of the dW GEMM)
    dl[M_DEC:].zero_()
    x = rnd((M_DEC + 32, D), 1.0, seed=14)
    x[M_DEC:].zero_()
    g1, g2 = torch.zeros(V, D, device="cuda"), torch.zeros(V, D, device="cuda")
    ops.gemm(dl[:, :V], x, trans_a=True, trans_b=True, out_dtype=torch.float32, out=g1, atomic_acc=True)
    ref.gemm(dl[:, :V], x, trans_a=True, trans_b=True, out_dtype=torch.float32, out=g2, atomic_acc=True)
    assert relerr(g1, g2) < 2e-3
    del logits, dl, g1, g2
    # encoder dW: K = 48 000 tokens, cut into slabs
    dy, xe = rnd((M_ENC, F), 0.05, seed=15), rnd((M_ENC, D), 1.0, seed=16)
    w1, w2 = torch.zeros(F, D, device="cuda"), torch.zeros(F, D, device="cuda")
    ops.gemm(dy, xe, trans_a=True, trans_b=True, out_dtype=torch.float32, out=w1, atomic_acc=True)
    ref.gemm(dy, xe, trans_a=True, trans_b=True, out_dtype=torch.float32, out=w2, atomic_acc=True)
    assert relerr(w1, w2) < 2e-3


def test_row_tail_of_the_wide_256_row_launches_is_bit_identical(ops):
    """dw_debug_set key 22 (default on): QKV / teacher-fc1 launches at M = 48 000 hand their 128-row tail to the 128-tile kernel
    (2 805 tiles = 10.96 rounds + 30 small tiles instead of 11.02 rounds -> 12).  Same fp32 chain per output element: the
    results must equal the single launch bit for bit, also for a ragged tail and with the GELU epilogue."""
    try:
        for M, N, kw in ((M_ENC, 3 * D, dict(bias=rnd((3 * D,), 1.0, torch.float32, seed=3))),
                         (M_ENC, F, dict(bias=rnd((F,), 1.0, torch.float32, seed=4), act=1)),
                         (M_ENC - 37, 3 * D, dict(bias=rnd((3 * D,), 1.0, torch.float32, seed=5)))):
            a = rnd((M, D), 1.0, seed=6)
            b = rnd((N, D), 0.03, seed=7)
            assert ops.lib.dw_debug_set(22, 0) == 0
            one = ops.gemm(a, b, **kw).clone()
            assert ops.lib.dw_debug_set(22, 1) == 0
            two = ops.gemm(a, b, **kw)
            assert torch.equal(one, two), (M, N)
    finally:
        ops.lib.dw_debug_set(22, 1)


@pytest.mark.parametrize("Lq,Lk,causal", [(1500, 1500, False), (447, 447, True), (447, 1500, False)])
def test_attention_at_model_shapes(ops, ref, Lq, Lk, causal):
    B = 4
    qkv = rnd((B * max(Lq, Lk), 3 * D), 1.0, seed=21)
    q, k, v = qkv[: B * Lq, :D], qkv[: B * Lk, D:2 * D], qkv[: B * Lk, 2 * D:]
    o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
    ro, rlse = ref.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
    assert (lse - rlse).abs().max().item() < 2e-3 and relerr(o, ro) < 1e-2
    do = rnd((B * Lq, D), 1.0, seed=22)
    dq, dk, dv = ops.attn_bwd(q, k, v, ro, do, rlse, B, H, Lq, Lk, causal, 0.125)
    rdq, rdk, rdv = ref.attn_bwd(q, k, v, ro, do, rlse, B, H, Lq, Lk, causal, 0.125)
    for a, b, n in ((dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")):
        assert relerr(a, b) < 1.5e-2, (n, relerr(a, b))


def test_layernorm_and_loss_at_model_shapes(ops, ref):
    x = rnd((M_ENC, D), 2.0, torch.float32, 31)
    gamma, beta = 1.0 + rnd((D,), 0.1, torch.float32, 32), rnd((D,), 0.1, torch.float32, 33)
    y, mu, rs = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    ry, rmu, rrs = ref.layernorm_fwd(x, gamma, beta, 1e-5)
    assert relerr(y, ry) < 3e-3 and relerr(mu, rmu) < 1e-5 and relerr(rs, rrs) < 1e-5
    dy = rnd((M_ENC, D), 1.0, seed=34)
    dres, rdres = rnd((M_ENC, D), 1.0, torch.float32, 35), None
    rdres = dres.clone()
    dg, db, rdg, rdb = (torch.zeros(D, device="cuda") for _ in range(4))
    lo, rlo = torch.empty(M_ENC, D, device="cuda", dtype=torch.bfloat16), torch.empty(M_ENC, D, device="cuda", dtype=torch.bfloat16)
    cs, rcs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy, x, mu, rs, gamma, dres, dg, db, out_lowp=lo, colsum=cs)
    ref.layernorm_bwd(dy, x, rmu, rrs, gamma, rdres, rdg, rdb, out_lowp=rlo, colsum=rcs)
    assert relerr(dres, rdres) < 1e-5 and relerr(lo, rlo) < 3e-3
    assert relerr(dg, rdg) < 1e-3 and relerr(db, rdb) < 1e-3 and relerr(cs, rcs) < 1e-3
    del x, y, ry, dy, dres, rdres, lo, rlo
    torch.cuda.empty_cache()
    ldv = (V + 63) // 64 * 64
    rows = 2048                                   # (the fp32 restatement keeps several [rows, V] temporaries)
    s, t = rnd((rows, ldv), 2.0, seed=41), rnd((rows, ldv), 2.0, seed=42)
    labels = torch.randint(0, 50257, (rows,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(43))
    labels[::7] = -100
    s2 = s.clone()
    got = ops.distill_loss(s, t, labels, V, 2.0, 0.8, 1.0, 1.0, True)
    want = ref.distill_loss(s2, t, labels, V, 2.0, 0.8, 1.0, 1.0, True)
    assert relerr(got[:3], want[:3]) < 1e-5 and got[3].item() == want[3].item()
    assert relerr(s[:, :V], s2[:, :V]) < 1e-2 and float(s[:, V:].float().abs().max()) == 0.0
