"""Parity of every HIP kernel (through the C ABI, via HipOps) against the torch restatement in oracle/ref_ops.py,
on the same seeded device tensors.  bf16 kernels: the restatement rounds at the same points, so tolerances are a few
bf16 ulps of the output scale; fp32 kernels (log-mel, loss, AdamW, LayerNorm stats) get tight tolerances."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


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


def maxerr(a, b):
    return (a.float() - b.float()).abs().max().item()


def test_tr16_semantics(ops):
    """ds_read_b64_tr_b16: output lane i (of a 16-lane group), element j = element (i&3) of the 8 bytes supplied by
    lane 4*j + (i>>2) of the same group.  Every transposed-operand fragment in gemm/attention relies on this."""
    out = ops.selftest_tr16().cpu()
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            src_lane = g * 16 + 4 * j + (i >> 2)
            assert out[lane, j].item() == src_lane * 4 + (i & 3), (lane, j, out[lane].tolist())


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(ops, ref, ta, tb, tile):
    M, N, K = 520, 392, 320  # ragged M/N edges (multiples of 8), K multiple of 64
    a = rnd((K, M) if ta else (M, K), seed=1)
    b = rnd((K, N) if tb else (N, K), seed=2)
    c = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=tile)
    r = ref.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
    assert relerr(c, r) < 1e-5, relerr(c, r)


def test_gemm_odd_m_rows(ops, ref):
    a = rnd((447 * 3, 384), seed=3)          # M = 1341: not a multiple of anything
    b = rnd((51866 // 16, 384), seed=4)      # N = 3241 (odd)
    c = ops.gemm(a, b, out_dtype=torch.float32)
    assert relerr(c, ref.gemm(a, b, out_dtype=torch.float32)) < 1e-5


def test_gemm_strided_views(ops, ref):
    """q/k/v are column slices of the fused QKV buffer: lda != K."""
    buf = rnd((300, 3 * 128), seed=5)
    w = rnd((256, 128), seed=6)
    a = buf[:, 128:256]
    c = ops.gemm(a, w, out_dtype=torch.float32)
    assert relerr(c, ref.gemm(a, w, out_dtype=torch.float32)) < 1e-5


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_epilogues(ops, ref, tile):
    M, N, K = 300, 512, 256
    a, b = rnd((M, K), 0.5, seed=7), rnd((N, K), 0.1, seed=8)
    bias = rnd((N,), 0.5, torch.float32, seed=9)
    # bias + gelu with pre-activation kept
    c, z = ops.gemm(a, b, bias=bias, act=1, want_z=True, tile=tile)
    rc, rz = ref.gemm(a, b, bias=bias, act=1, want_z=True)
    assert maxerr(z, rz) <= 0.04 and relerr(z, rz) < 3e-3
    assert relerr(c, rc) < 6e-3
    # fp32 residual stream, rounded projection output (student)
    res = rnd((M, N), 1.0, torch.float32, seed=10)
    c = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32, tile=tile)
    rc = ref.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    assert relerr(c, rc) < 2e-3
    # in-place accumulation (R aliases C), no rounding
    acc = res.clone()
    ops.gemm(a, b, residual=acc, round_res=False, out_dtype=torch.float32, out=acc, tile=tile)
    rc = ref.gemm(a, b, residual=res, round_res=False, out_dtype=torch.float32)
    assert relerr(acc, rc) < 1e-5
    # bf16 residual stream (teacher) + positional broadcast (row modulo)
    pos = rnd((100, N), 1.0, torch.float32, seed=11)
    c = ops.gemm(a, b, bias=bias, act=1, residual=pos, r_row_mod=100, out_dtype=torch.float32, tile=tile)
    rc = ref.gemm(a, b, bias=bias, act=1, residual=pos, r_row_mod=100, out_dtype=torch.float32)
    assert relerr(c, rc) < 3e-3
    # fused GELU backward
    zg = rnd((M, N), 1.0, seed=12)
    c = ops.gemm(a, b, zgrad=zg, tile=tile)
    rc = ref.gemm(a, b, zgrad=zg)
    assert relerr(c, rc) < 6e-3
    # the student's FFN keeps gelu'(z) in fp16 instead of z (DwGemm.z_is_gelu_grad): same C, derivative within fp16
    # rounding of the fp32 derivative, and the backward epilogue that multiplies by it equals the one that evaluates it
    # up to that rounding; ragged N exercises the scalar edge path of both
    for n in (N, N - 3):
        c2, g = ops.gemm(a, b[:n], bias=bias[:n], act=1, want_z="grad", tile=tile)
        c1, z1 = ops.gemm(a, b[:n], bias=bias[:n], act=1, want_z=True, tile=tile)
        rc, rg = ref.gemm(a, b[:n], bias=bias[:n], act=1, want_z="grad")
        assert g.dtype == torch.float16 and torch.equal(c1, c2)
        want = ref.gemm(a, b[:n], zgrad=z1)                   # derivative evaluated from the kernel's own z
        dg = ref.gemm(torch.zeros_like(a), b[:n], bias=torch.ones_like(bias[:n]), zgrad=z1, out_dtype=torch.float32)
        assert (g.float() - dg).abs().max().item() <= 1.5e-3   # fp16 ulp at 1 is 9.8e-4, + bf16-z tie flips
        got = ops.gemm(a, b[:n], zgrad=g, tile=tile)
        assert relerr(got, want) < 6e-3 and relerr(got, ops.gemm(a, b[:n], zgrad=z1, tile=tile)) < 3e-3
    with pytest.raises(RuntimeError):                        # gelu'(z) is a by-product of the GELU epilogue only
        g2 = torch.empty((M, N), dtype=torch.float16, device=a.device)
        import ctypes
        from distil_whisper_amd import ops_hip
        d = ops_hip.DwGemm()
        ctypes.memset(ctypes.byref(d), 0, ctypes.sizeof(d))
        cc = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
        d.a, d.b, d.c, d.z_out = a.data_ptr(), b.data_ptr(), cc.data_ptr(), g2.data_ptr()
        d.m, d.n, d.k, d.lda, d.ldb, d.ldc, d.ldz = M, N, K, K, K, N, N
        d.c_dtype, d.z_is_gelu_grad = ops_hip.DW_BF16, 1
        ops._chk(ops.lib.dw_gemm_bf16(ctypes.byref(d), ops._stream()), "z_is_gelu_grad without act")


@pytest.mark.parametrize("variant,ta,tb", [(131, False, True), (19, False, False), (35, False, True), (67, True, True)])
def test_gemm_pipelined_variants_are_bit_identical(ops, ref, variant, ta, tb):
    """gemm_phased.hip (variant bit 7: counted-vmcnt, slot-staggered main loop, region-major LDS image) and the 8-wave
    software-pipelined kernels of gemm_wp.h (bits 4-6: register double-buffered fragments, pinned MFMA / LDS / DMA
    interleave, buffer-addressed operand DMA) against the plain 16-wave kernel: same k order per accumulator, so every
    output bit must agree -- each operand layout the kernels are built for, ragged M/N edges, 1..5 K tiles (prologue /
    tail wait counts), persistent job walk (> 256 tiles), fused epilogue, split-K slices, repeated launches (race
    screen)."""
    try:
        for M, N, K in ((520, 392, 64), (520, 392, 128), (304, 512, 192), (776, 1024, 320), (4096, 4608, 256),
                        (8 * 1500, 1280, 1280)):
            a = rnd((K, M) if ta else (M, K), 0.5, seed=41)
            b = rnd((K, N) if tb else (N, K), 0.1, seed=42)
            bias = rnd((N,), 0.5, torch.float32, seed=43)
            ops.lib.dw_debug_set(0, 3)
            want = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, act=1, tile=256).clone()
            want32 = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=256).clone()
            ops.lib.dw_debug_set(0, variant)
            for rep in range(3):
                got = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, act=1, tile=256)
                assert torch.equal(got, want), (M, N, K, rep, (got.float() - want.float()).abs().max().item())
                got32 = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=256)
                assert torch.equal(got32, want32), (M, N, K, rep)
            assert relerr(want32, ref.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)) < 1e-5
        if ta and tb:       # weight-gradient form: K = tokens, split into slices, fp32 partials + deterministic reduce
            a, b = rnd((6400, 1280), 0.5, seed=44), rnd((6400, 384), 0.5, seed=45)
            ops.lib.dw_debug_set(0, 3)
            want = torch.zeros(1280, 384, device="cuda")
            ops.gemm(a, b, trans_a=True, trans_b=True, out_dtype=torch.float32, out=want, atomic_acc=True, split_k=5)
            ops.lib.dw_debug_set(0, variant)
            got = torch.zeros(1280, 384, device="cuda")
            ops.gemm(a, b, trans_a=True, trans_b=True, out_dtype=torch.float32, out=got, atomic_acc=True, split_k=5)
            assert torch.equal(got, want)
    finally:
        ops.lib.dw_debug_set(0, 2163)


@pytest.mark.parametrize("tb", [False, True])
def test_gemm_320_row_tiles_are_bit_identical(ops, ref, tb):
    """gemm_wp8_m320.hip (variant bit 2048): 320 x 256 block tiles, 160 x 64 per wave, one address register per operand
    with the piece stride in the scalar offset.  Same k order per accumulator as the 16-wave 256-row kernel, so every
    output bit must agree: both B layouts, 1..5 and 20 K tiles, one and several tile rounds (persistent walk), every
    fused epilogue flavour (bias + GELU, stored gelu', GELU' input, bf16 / fp32 residual, fp32 output), repeated launches.
    Shapes the 320-row tile is not built for (M % 320, N % 256) must fall back to the 256-row kernels unchanged."""
    try:
        for M, N, K in ((640, 512, 64), (960, 256, 128), (1280, 768, 192), (3200, 1280, 320), (320 * 150, 1280, 1280),
                        (320 * 30, 3840, 256)):
            a = rnd((M, K), 0.5, seed=51)
            b = rnd((K, N) if tb else (N, K), 0.1, seed=52)
            bias = rnd((N,), 0.5, torch.float32, seed=53)
            r16 = rnd((M, N), 1.0, seed=54)
            r32 = rnd((M, N), 1.0, torch.float32, seed=55)
            zin = rnd((M, N), 1.0, seed=56)
            flavours = (dict(bias=bias, act=1), dict(out_dtype=torch.float32), dict(bias=bias, residual=r16),
                        dict(bias=bias, residual=r32, out_dtype=torch.float32), dict(zgrad=zin), dict(bias=bias, act=1, want_z=True),
                        dict(bias=bias, act=1, want_z="grad"), dict(zgrad=zin.to(torch.float16)))
            want = []
            ops.lib.dw_debug_set(0, 3)
            for f in flavours:
                o = ops.gemm(a, b, trans_b=tb, tile=256, **f)
                want.append([t.clone() for t in o] if isinstance(o, tuple) else [o.clone()])
            ops.lib.dw_debug_set(0, 115 | 2048 | 4096)
            for rep in range(2):
                for f, w in zip(flavours, want):
                    o = ops.gemm(a, b, trans_b=tb, tile=256, **f)
                    o = list(o) if isinstance(o, tuple) else [o]
                    for g_, w_ in zip(o, w):
                        assert torch.equal(g_, w_), (M, N, K, tb, sorted(f), rep, (g_.float() - w_.float()).abs().max().item())
            assert relerr(want[1][0], ref.gemm(a, b, trans_b=tb, out_dtype=torch.float32)) < 1e-5
        # automatic tile choice: a grid too small for two rounds of 256-tiles that is ONE round of 320-row tiles (the
        # teacher decoder's padded M = 14400, N = 1280: 225 workgroups) takes that round instead of 128-tiles
        if not tb:
            a, b = rnd((14400, 1280), 0.5, seed=59), rnd((1280, 1280), 0.1, seed=60)
            bias, r16 = rnd((1280,), 0.5, torch.float32, seed=61), rnd((14400, 1280), 1.0, seed=62)
            ops.lib.dw_debug_set(0, 3)
            w = ops.gemm(a, b, bias=bias, residual=r16).clone()              # 128-tiles
            ops.lib.dw_debug_set(0, 2163)
            assert torch.equal(ops.gemm(a, b, bias=bias, residual=r16), w)   # one round of 320-row tiles
        # not eligible: ragged M / N -> the 256-row kernels, same bits
        a, b = rnd((1000, 128), 0.5, seed=57), rnd((128, 304) if tb else (304, 128), 0.1, seed=58)
        ops.lib.dw_debug_set(0, 3)
        w = ops.gemm(a, b, trans_b=tb, tile=256).clone()
        ops.lib.dw_debug_set(0, 115 | 2048 | 4096)
        assert torch.equal(ops.gemm(a, b, trans_b=tb, tile=256), w)
    finally:
        ops.lib.dw_debug_set(0, 2163)


@pytest.mark.gpu
@pytest.mark.parametrize("tb", [False, True])
def test_gemm_small_m_rule_is_bit_identical(ops, ref, tb):
    """Outputs with fewer than two rounds of 256-row tiles (dw_debug_set key 25, gemm.hip): the 128 x 256 tile in a THREE-stage
    operand ring (gemm_wp8_m128.hip: counted vmcnt waits, buffers rotating over t mod 3), the 256-row tile on 16x16x32 or the
    320-row tile, chosen by the rounds of the CUs each needs.  Same fp32 chain over k per element as the lock-step 128 x 128
    kernel the rule replaces: every bit agrees -- ragged M and N, 1..5 / 20 / 80 K tiles (every prologue / tail path of the
    ring), one and several rounds of tiles, the step's epilogue flavours, repeated launches."""
    try:
        shapes = ((200, 392, 64), (520, 392, 128), (776, 1024, 192), (1100, 520, 256), (1300, 768, 320), (4258, 1280, 1280),
                  (4480, 3840, 1280), (2240, 1280, 5120), (7136, 1280, 1280), (4352, 5120, 1280))
        for M, N, K in shapes:
            a = rnd((M, K), 0.5, seed=81)
            b = rnd((K, N) if tb else (N, K), 0.1, seed=82)
            bias = rnd((N,), 0.5, torch.float32, seed=83)
            r16 = rnd((M, N), 1.0, seed=84)
            r32 = rnd((M, N), 1.0, torch.float32, seed=85)
            zin = rnd((M, N), 1.0, seed=86).to(torch.float16)
            flavours = (dict(), dict(bias=bias), dict(bias=bias, act=1), dict(bias=bias, residual=r16),
                        dict(bias=bias, residual=r32, out_dtype=torch.float32), dict(zgrad=zin), dict(bias=bias, act=1, want_z="grad"))
            if K >= 5120: flavours = flavours[:2] + flavours[3:4]
            ops.lib.dw_debug_set(25, 0)
            want = []
            for f in flavours:
                o = ops.gemm(a, b, trans_b=tb, tile=128, **f)
                want.append([t.clone() for t in o] if isinstance(o, tuple) else [o.clone()])
            ops.lib.dw_debug_set(25, 1)
            for tile in (0, 129):
                for rep in range(2):
                    for f, w in zip(flavours, want):
                        o = ops.gemm(a, b, trans_b=tb, tile=tile, **f)
                        o = list(o) if isinstance(o, tuple) else [o]
                        for g_, w_ in zip(o, w):
                            assert torch.equal(g_, w_), (M, N, K, tb, tile, sorted(f), rep, (g_.float() - w_.float()).abs().max().item())
            assert relerr(want[0][0].float(), ref.gemm(a, b, trans_b=tb, out_dtype=torch.float32)) < 1e-2
    finally:
        ops.lib.dw_debug_set(25, 1)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_16x16x32_main_loop_is_bit_identical(ops, ref, ta, tb):
    """gemm_wp16.h: the software-pipelined main loop on v_mfma_f32_16x16x32_bf16 (dw_debug_set key 20; the default for the
    weight-gradient layout and for wide row-major outputs) -- 16 x 32 fragments under their own LDS swizzles, the k-major image through ds_read_b64_tr_b16 in
    the 16 x 16 x 32 operand order, accumulators as 16 x 16 blocks through the LAY = 16 epilogue.  The hardware accumulates a
    32-deep instruction in the order two 16-deep ones do, so every output bit agrees with the 32x32x16 kernels: each
    layout, 256-row and 320-row tiles, ragged edges, 1..5 and 20 K tiles, persistent walk, every epilogue flavour of the
    step (both epilogue walks), split-K slices, repeated launches."""
    try:
        shapes = ((520, 392, 64), (776, 1024, 320), (640, 512, 128), (3200, 1280, 192), (320 * 30, 3840, 256), (8 * 1500, 1280, 1280))
        for M, N, K in shapes:
            a = rnd((K, M) if ta else (M, K), 0.5, seed=71)
            b = rnd((K, N) if tb else (N, K), 0.1, seed=72)
            bias = rnd((N,), 0.5, torch.float32, seed=73)
            r16 = rnd((M, N), 1.0, seed=74)
            r32 = rnd((M, N), 1.0, torch.float32, seed=75)
            zin = rnd((M, N), 1.0, seed=76).to(torch.float16)
            flavours = (dict(), dict(bias=bias), dict(bias=bias, act=1), dict(out_dtype=torch.float32), dict(bias=bias, residual=r16),
                        dict(bias=bias, residual=r32, out_dtype=torch.float32), dict(zgrad=zin), dict(bias=bias, act=1, want_z="grad"))
            want = []
            ops.lib.dw_debug_set(20, 0)
            for f in flavours:
                o = ops.gemm(a, b, trans_a=ta, trans_b=tb, tile=256, **f)
                want.append([t.clone() for t in o] if isinstance(o, tuple) else [o.clone()])
            # 7: every layout on 16x16x32; 7 | 32: + the 256-row tile where the 320-row one is the rule's choice; 8 | 16: the four-wave
            # layout (one wave per SIMD, 128 x 128 per wave, accumulators pinned to AGPRs) for the row-major-A layouts
            for mask in ((7, 7 | 32) if ta else (7, 7 | 32, 8 | 16)):
                ops.lib.dw_debug_set(20, mask)
                if mask & 8: ops.lib.dw_debug_set(0, 115)          # (the four-wave kernels take the 256-row tile's place)
                for rep in range(2):
                    for f, w in zip(flavours, want):
                        o = ops.gemm(a, b, trans_a=ta, trans_b=tb, tile=256, **f)
                        o = list(o) if isinstance(o, tuple) else [o]
                        for g_, w_ in zip(o, w):
                            assert torch.equal(g_, w_), (M, N, K, ta, tb, mask, sorted(f), rep, (g_.float() - w_.float()).abs().max().item())
                ops.lib.dw_debug_set(0, 2163)
            assert relerr(want[3][0], ref.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)) < 1e-5
        if ta and tb:       # weight-gradient form with K slices
            a, b = rnd((6400, 1280), 0.5, seed=77), rnd((6400, 384), 0.5, seed=78)
            outs = []
            for v in (0, 7):
                ops.lib.dw_debug_set(20, v)
                o = torch.zeros(1280, 384, device="cuda")
                ops.gemm(a, b, trans_a=True, trans_b=True, out_dtype=torch.float32, out=o, atomic_acc=True, split_k=5)
                outs.append(o)
            assert torch.equal(outs[0], outs[1])
    finally:
        ops.lib.dw_debug_set(20, 36)
        ops.lib.dw_debug_set(0, 2163)


def test_gemm_dynamic_job_handout_is_invisible(ops, ref):
    """Persistent 256-tile kernels draw their tiles from per-XCD device counters (dw_debug_set key 10; the last
    workgroup resets them): the tile -> workgroup assignment changes, the results must not -- three launches in a row
    (the counters must come back to zero), every operand layout, a grid smaller than the CU count (key 9), and a side
    stream with its own counters."""
    M, N, K = 9000, 2048, 128      # 288 tiles on 256 workgroups
    try:
        for ta, tb in ((False, False), (False, True), (True, True)):
            a = rnd((K, M) if ta else (M, K), 0.5, seed=81)
            b = rnd((K, N) if tb else (N, K), 0.1, seed=82)
            ops.lib.dw_debug_set(10, 0)
            want = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=256).clone()
            ops.lib.dw_debug_set(10, 1)
            for cus in (256, 64, 256):
                ops.lib.dw_debug_set(9, cus)
                for rep in range(3):
                    assert torch.equal(ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=256), want)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, tile=256)
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(got, want)
    finally:
        ops.lib.dw_debug_set(9, 256); ops.lib.dw_debug_set(10, 1)


def test_gemm_next_tile_staging_under_the_epilogue_is_invisible(ops, ref):
    """The software-pipelined kernels request the NEXT tile's first operand block into LDS buffer 0 while the epilogue
    of the current tile runs out of buffer 1 (dw_debug_set key 11; even K-tile counts only).  Same bits with it on and
    off: every operand layout, 2 / 4 / 20 K tiles (and an odd count, which must not stage), more tiles than
    workgroups with ragged edges, static and dynamic hand-out, epilogues with and without side inputs, repeated
    launches (race screen)."""
    try:
        for ta, tb in ((False, False), (False, True), (True, True)):
            for M, N, K in ((9000, 2048, 128), (9000, 2000, 256), (4104, 5120, 1280), (9000, 2048, 192)):
                a = rnd((K, M) if ta else (M, K), 0.5, seed=91)
                b = rnd((K, N) if tb else (N, K), 0.1, seed=92)
                bias = rnd((N,), 0.5, torch.float32, seed=93)
                res = rnd((M, N), 1.0, torch.float32, seed=94)
                zg = rnd((M, N), 1.0, seed=95)
                runs = (lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, act=1, tile=256),
                        lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, residual=res,
                                         out_dtype=torch.float32, tile=256),
                        lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, zgrad=zg, tile=256))
                ops.lib.dw_debug_set(11, 0)
                want = [f().clone() for f in runs]
                ops.lib.dw_debug_set(11, 1)
                for dyn in (1, 0):
                    ops.lib.dw_debug_set(10, dyn)
                    for rep in range(2):
                        for f, w in zip(runs, want):
                            assert torch.equal(f(), w), (ta, tb, M, N, K, dyn)
    finally:
        ops.lib.dw_debug_set(10, 1); ops.lib.dw_debug_set(11, 1)


@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,n_new", [(1, 1), (16, 1), (12, 3), (32, 2)])
def test_gemm_skinny_layernorm_on_load_and_kv_append(ops, ref, xdtype, M, n_new):
    """Decode-step fusions of the weight-streaming kernel: A = bf16(LayerNorm(x)) built on load (x f32 or bf16) and the
    K/V columns of the fused QKV projection stored straight into the cache rows of their positions -- against the
    unfused launches (LayerNorm kernel + GEMM + strided copy) and the torch restatement."""
    D, ML, t0 = 1280, 24, 5
    B = M // n_new
    x = rnd((M, D), 2.0, xdtype, seed=70) + 0.5
    gamma, beta = rnd((D,), 0.5, torch.float32, seed=71) + 1.0, rnd((D,), 0.5, torch.float32, seed=72)
    w, bias = rnd((3 * D, D), 0.03, seed=73), rnd((3 * D,), 0.5, torch.float32, seed=74)
    cache = rnd((B * ML, 2 * D), 1.0, seed=75)
    want_cache, ref_cache = cache.clone(), cache.clone()
    h = ops.layernorm_fwd(x, gamma, beta, 1e-5, save_stats=False)[0]
    want = ops.gemm(h, w, bias=bias)
    want_cache.view(B, ML, 2 * D)[:, t0:t0 + n_new].copy_(want[:, D:].view(B, n_new, 2 * D))
    got = ops.gemm(x, w, bias=bias, ln=(gamma, beta, 1e-5), kv_append=(cache, D, n_new, ML, t0))
    r = ref.gemm(x, w, bias=bias, ln=(gamma, beta, 1e-5), kv_append=(ref_cache, D, n_new, ML, t0))
    # the fused statistics sum in a different order than csrc/norm.hip: bf16 operands may differ in the last bit
    assert relerr(got[:, :D], want[:, :D]) < 4e-3 and relerr(got[:, :D], r[:, :D]) < 4e-3
    assert relerr(cache, want_cache) < 4e-3 and relerr(cache, ref_cache) < 4e-3
    untouched = torch.ones(B, ML, dtype=torch.bool); untouched[:, t0:t0 + n_new] = False
    assert torch.equal(cache.view(B, ML, 2 * D)[untouched.cuda()], want_cache.view(B, ML, 2 * D)[untouched.cuda()])
    # GELU epilogue + LayerNorm on load (the fc1 launch of the token step)
    w1, b1 = rnd((5120, D), 0.03, seed=76), rnd((5120,), 0.5, torch.float32, seed=77)
    assert relerr(ops.gemm(x, w1, bias=b1, act=1, ln=(gamma, beta)), ops.gemm(h, w1, bias=b1, act=1)) < 6e-3
    with pytest.raises(RuntimeError):        # the fusions exist in the skinny-M kernel only
        ops.gemm(rnd((128, D), 1.0, xdtype, seed=78), w, bias=bias, ln=(gamma, beta))


@pytest.mark.parametrize("M", [1, 16, 17, 40, 64])
@pytest.mark.parametrize("N,K", [(1280, 1280), (5120, 1280), (1280, 5120), (48, 64), (51904, 1280)])
def test_gemm_skinny_decode_shapes(ops, ref, M, N, K):
    """The weight-streaming kernel of the decode step (M = batch rows <= 64) against the restatement and against the
    tile kernel, with every epilogue the decode step uses."""
    a, w = rnd((M, K), 1.0, seed=30), rnd((N, K), 0.03, seed=31)
    bias = rnd((N,), 1.0, torch.float32, seed=32)
    c = ops.gemm(a, w, out_dtype=torch.float32, tile=16)
    assert relerr(c, ref.gemm(a, w, out_dtype=torch.float32)) < 1e-5
    assert relerr(c, ops.gemm(a, w, out_dtype=torch.float32, tile=128)) < 1e-5
    c = ops.gemm(a, w, bias=bias, act=1, tile=16)
    assert relerr(c, ref.gemm(a, w, bias=bias, act=1)) < 4e-3
    for rdt in (torch.float32, torch.bfloat16):
        res = rnd((M, N), 1.0, rdt, seed=33)
        c = ops.gemm(a, w, bias=bias, residual=res, round_res=True, out_dtype=rdt)      # auto dispatch (M <= 64)
        assert relerr(c, ref.gemm(a, w, bias=bias, residual=res, round_res=True, out_dtype=rdt)) < 4e-3
    # strided activations / weights (views into larger buffers, as the fused QKV weight and the KV cache rows are)
    big_a, big_w = rnd((M, K + 64), 1.0, seed=34), rnd((N + 16, K), 0.03, seed=35)
    c = ops.gemm(big_a[:, 64:], big_w[16:], out_dtype=torch.float32, tile=16)
    assert relerr(c, ref.gemm(big_a[:, 64:], big_w[16:], out_dtype=torch.float32)) < 1e-5


@pytest.mark.parametrize("tile,split_k", [(128, 1), (128, 5), (256, 3), (0, 0)])
def test_gemm_atomic_split_k(ops, ref, tile, split_k):
    """Weight-gradient form: out (fp32) += A^T . B over a long token dimension, K range split across workgroups."""
    K, M, N = 64 * 37, 328, 520
    a, b = rnd((K, M), 1.0, seed=15), rnd((K, N), 1.0, seed=16)
    base = rnd((M, N), 1.0, torch.float32, seed=17)
    out = base.clone()
    ops.gemm(a, b, trans_a=True, trans_b=True, out=out, atomic_acc=True, tile=tile, split_k=split_k)
    r = base + ref.gemm(a, b, trans_a=True, trans_b=True, out_dtype=torch.float32)
    assert relerr(out, r) < 1e-5


def test_gemm_full_size_linearity(ops):
    """BASELINE-size property check (no oracle needed): (A1+A2).W == A1.W + A2.W up to fp32 accumulation, and the
    256-tile and 128-tile kernels agree, at the distil-large-v3 encoder FFN shape."""
    M, N, K = 32 * 1500, 5120, 1280
    a = rnd((M, K), 1.0, seed=13)
    w = rnd((N, K), 0.03, seed=14)
    c256 = ops.gemm(a, w, out_dtype=torch.float32, tile=256)
    c128 = ops.gemm(a, w, out_dtype=torch.float32, tile=128)
    assert relerr(c256, c128) < 1e-6
    rows = torch.randint(0, M, (64,), device="cuda")
    r = a[rows].float() @ w.float().t()
    assert relerr(c256[rows], r) < 1e-5


@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [384, 768, 1280])
def test_layernorm(ops, ref, D, x_dtype):
    rows = 517
    x = rnd((rows, D), 2.0, x_dtype, seed=20) + 0.5
    gamma = rnd((D,), 1.0, torch.float32, seed=21)
    beta = rnd((D,), 1.0, torch.float32, seed=22)
    y, mu, rs = ops.layernorm_fwd(x, gamma, beta)
    ry, rmu, rrs = ref.layernorm_fwd(x, gamma, beta)
    assert maxerr(mu, rmu) < 1e-5 and relerr(rs, rrs) < 1e-5
    assert relerr(y, ry) < 3e-3
    dy = rnd((rows, D), 1.0, seed=23)
    dres = rnd((rows, D), 1.0, torch.float32, seed=24)
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    rdg, rdb = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    out = ops.layernorm_bwd(dy, x, mu, rs, gamma, dres.clone(), dg, db)
    rout = ref.layernorm_bwd(dy, x, rmu, rrs, gamma, dres.clone(), rdg, rdb)
    assert relerr(out, rout) < 1e-5
    assert relerr(dg, rdg) < 1e-4 and relerr(db, rdb) < 1e-4
    out2 = ops.layernorm_bwd(dy, x, mu, rs, gamma, None, dg, db)
    assert relerr(out2, rout - dres) < 1e-4
    # fused low-precision copy + column sums (bias gradient of the next branch)
    lo, rlo = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16), torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    cs, rcs = torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
    o3 = ops.layernorm_bwd(dy, x, mu, rs, gamma, dres.clone(), dg, db, out_lowp=lo, colsum=cs)
    r3 = ref.layernorm_bwd(dy, x, rmu, rrs, gamma, dres.clone(), rdg, rdb, out_lowp=rlo, colsum=rcs)
    assert relerr(o3, r3) < 1e-5 and relerr(lo, rlo) < 3e-3 and relerr(cs, rcs) < 1e-3


@pytest.mark.parametrize("D,x_dtype,rows", [(1280, torch.float32, 9000), (1280, torch.bfloat16, 517), (384, torch.float32, 517)])
def test_layernorm_with_padded_row_pitches_is_bit_identical(ops, D, x_dtype, rows):
    """dw_layernorm_fwd_ld / _bwd_ld: every 2-D argument as a [rows, D] view of a buffer whose rows are D + pad elements apart
    (the engine's activation buffers, padded by 128 bytes per row) -- same values as with dense rows, bit for bit, and the pad
    columns untouched.  (rows = 9000 takes the persistent forward / prefetching backward kernels of the encoder shape.)"""
    def padded(t, pad, fill):
        buf = torch.full((t.shape[0], t.shape[1] + pad), fill, device="cuda", dtype=t.dtype)
        buf[:, :t.shape[1]] = t
        return buf, buf[:, :t.shape[1]]
    x = rnd((rows, D), 2.0, x_dtype, seed=30) + 0.5
    gamma, beta = rnd((D,), 1.0, torch.float32, seed=31), rnd((D,), 1.0, torch.float32, seed=32)
    y, mu, rs = ops.layernorm_fwd(x, gamma, beta)
    xb, xp = padded(x, 32 if x_dtype == torch.float32 else 64, 7.0)
    yb, yp = padded(torch.zeros_like(y), 64, 3.0)
    y2, mu2, rs2 = ops.layernorm_fwd(xp, gamma, beta, out=yp)
    assert torch.equal(y2, y) and torch.equal(mu2, mu) and torch.equal(rs2, rs)
    assert bool((yb[:, D:] == 3.0).all())
    dy = rnd((rows, D), 1.0, seed=33)
    dres = rnd((rows, D), 1.0, torch.float32, seed=34)
    lo = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    dg, db, cs = (torch.zeros(D, device="cuda") for _ in range(3))
    out = ops.layernorm_bwd(dy, x, mu, rs, gamma, dres.clone(), dg, db, out_lowp=lo, colsum=cs)
    dyb, dyp = padded(dy, 64, 5.0)
    drb, drp = padded(dres, 32, 9.0)
    lob, lop = padded(torch.zeros_like(lo), 64, 2.0)
    dg2, db2, cs2 = (torch.zeros(D, device="cuda") for _ in range(3))
    out2 = ops.layernorm_bwd(dyp, xp, mu, rs, gamma, drp, dg2, db2, out_lowp=lop, colsum=cs2)
    assert out2.data_ptr() == drp.data_ptr()
    assert torch.equal(out2, out) and torch.equal(lop, lo)
    assert bool((drb[:, D:] == 9.0).all()) and bool((lob[:, D:] == 2.0).all())
    # (dgamma / dbeta / column sums are atomic sums over the same per-workgroup partials: equal up to the order of the atomics)
    assert relerr(dg2, dg) < 1e-5 and relerr(db2, db) < 1e-5 and relerr(cs2, cs) < 1e-5


ATTN_CASES = [
    (2, 3, 1500, 1500, False),   # encoder self
    (2, 3, 447, 447, True),      # decoder self (causal)
    (2, 3, 447, 1500, False),    # cross
    (1, 2, 64, 64, True),
    (1, 1, 5, 70, False),        # tiny ragged
]


@pytest.mark.parametrize("B,H,Lq,Lk,causal", ATTN_CASES)
def test_attention_fwd_bwd(ops, ref, B, H, Lq, Lk, causal):
    D = H * 64
    # q,k,v as column slices of fused projections, like the engine uses them
    qkv = rnd((B * max(Lq, Lk), 3 * D), 1.0, seed=30)
    q = qkv[: B * Lq, :D]
    k = qkv[: B * Lk, D:2 * D]
    v = qkv[: B * Lk, 2 * D:]
    o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
    ro, rlse = ref.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
    assert maxerr(lse, rlse) < 2e-3, maxerr(lse, rlse)
    assert relerr(o, ro) < 1e-2, relerr(o, ro)
    do = rnd((B * Lq, D), 1.0, seed=31)
    dq, dk, dv = ops.attn_bwd(q, k, v, ro, do, rlse, B, H, Lq, Lk, causal, 0.125)
    rdq, rdk, rdv = ref.attn_bwd(q, k, v, ro, do, rlse, B, H, Lq, Lk, causal, 0.125)
    assert relerr(dv, rdv) < 1.5e-2, relerr(dv, rdv)
    assert relerr(dq, rdq) < 1.5e-2, relerr(dq, rdq)
    assert relerr(dk, rdk) < 1.5e-2, relerr(dk, rdk)
    # dw_attn_bwd_ex: the q / v bias gradients (column sums of the stored dq / dv) come out of the same kernels, ADDED to
    # their buffers; the matrices themselves are unchanged by it
    bq, bv = torch.full((D,), 0.5, device="cuda"), torch.full((D,), -0.25, device="cuda")
    dq2, dk2, dv2 = ops.attn_bwd(q, k, v, ro, do, rlse, B, H, Lq, Lk, causal, 0.125, dq_colsum=bq, dv_colsum=bv)
    assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv)
    assert relerr(bq - 0.5, dq.float().sum(0)) < 1e-4 and relerr(bv + 0.25, dv.float().sum(0)) < 1e-4


@pytest.mark.parametrize("B,H,Lq,Lk,pitch", [(2, 2, 6, 37, 48), (3, 1, 1, 9, 16), (1, 2, 70, 200, 200), (2, 1, 33, 33, 40),
                                              (1, 1, 130, 447, 448)])
def test_attention_bottom_right_causal_against_padded_kv_cache(ops, ref, B, H, Lq, Lk, pitch):
    """causal = 2: query i sees keys <= i + Lk - Lq, K/V read in place from a cache with `pitch` rows per batch --
    the multi-token verify / prefill step of cached decoding (engine.decode_multi)."""
    D = H * 64
    q = rnd((B * Lq, D), 1.0, seed=35)
    cache = rnd((B * pitch, 2 * D), 1.0, seed=36)
    o, lse = ops.attn_fwd(q, cache[:, :D], cache[:, D:], B, H, Lq, Lk, 2, 0.125, kv_batch_rows=pitch)
    ro, rlse = ref.attn_fwd(q, cache[:, :D], cache[:, D:], B, H, Lq, Lk, 2, 0.125, kv_batch_rows=pitch)
    assert maxerr(lse, rlse) < 2e-3 and relerr(o, ro) < 1e-2
    # the last query row sees every key; the first one exactly Lk - Lq + 1 of them
    o1, _ = ops.attn_fwd(q.view(B, Lq, D)[:, -1].contiguous(), cache[:, :D], cache[:, D:], B, H, 1, Lk, False, 0.125,
                         kv_batch_rows=pitch)
    assert relerr(o.view(B, Lq, D)[:, -1], o1) < 5e-3      # (a couple of bf16 ulps: tile kernel vs streaming kernel)


@pytest.mark.parametrize("H,T,lens,Lk", [(2, 40, [7, 40, 1, 33], 70), (1, 224, [130, 224, 64, 65, 1], 1500), (3, 96, [96, 5], 200)])
def test_attention_varlen_equals_rectangular_rows(ops, ref, H, T, lens, Lk):
    """dw_attn_fwd_varlen over packed rows (the frozen teacher's decoder over the live positions of a batch): every live row is
    BIT-IDENTICAL to the row dw_attn_fwd computes over the (batch, position) rectangle -- causal self-attention whose keys /
    values are rows of the same packed buffers, and cross-attention against rectangular K / V; filler pseudo-sequences and
    zero-length table entries (a captured plan's fixed table) leave finite rows and touch nothing else."""
    from distil_whisper_amd.engine import LiveRows
    B, D = len(lens), H * 64
    rect = rnd((B * T, 3 * D), 1.0, seed=70)
    fill_to = sum(lens) + 9 if sum(lens) + 9 <= B * T else None
    idx = LiveRows.host_index(lens, T, fill_to=fill_to).cuda()
    st, ln = LiveRows.seq_table(lens, T, fill_to=fill_to, entries=B + 3 if fill_to else None)
    st, ln = st.cuda(), ln.cuda()
    R = idx.numel()
    packed = rect.index_select(0, idx.long()).contiguous()
    # causal self-attention
    want, _ = ops.attn_fwd(rect[:, :D], rect[:, D:2 * D], rect[:, 2 * D:], B, H, T, T, True, 0.125)
    got = torch.full((R + 4, D), float("nan"), device="cuda").bfloat16()
    ops.attn_fwd_varlen(packed[:, :D], packed[:, D:2 * D], packed[:, 2 * D:], H, T, st, ln, True, 0.125, got[:R], self_attention=True)
    n_live = sum(lens)
    assert torch.equal(got[:n_live], want.index_select(0, idx[:n_live].long()))
    assert torch.isfinite(got[:R].float()).all() and torch.isnan(got[R:].float()).all()     # filler rows finite, nothing behind written
    r = ref.attn_fwd_varlen(packed[:, :D], packed[:, D:2 * D], packed[:, 2 * D:], H, T, st, ln, True, 0.125,
                            torch.zeros_like(got[:R]), self_attention=True)
    assert relerr(got[:n_live], r[:n_live]) < 1e-2
    # cross-attention against rectangular keys / values
    kv = rnd((B * Lk, 2 * D), 1.0, seed=71)
    want, _ = ops.attn_fwd(rect[:, :D], kv[:, :D], kv[:, D:], B, H, T, Lk, False, 0.125)
    got = torch.full((R + 4, D), float("nan"), device="cuda").bfloat16()
    ops.attn_fwd_varlen(packed[:, :D], kv[:, :D], kv[:, D:], H, T, st, ln, False, 0.125, got[:R], Lk=Lk, kv_batches=B, self_attention=False)
    assert torch.equal(got[:n_live], want.index_select(0, idx[:n_live].long()))
    assert torch.isfinite(got[:R].float()).all() and torch.isnan(got[R:].float()).all()


def test_attention_spiked_row(ops, ref):
    """Force the online-softmax rescale: one key per tile dominates a query's row."""
    B, H, L = 1, 1, 320
    q, k, v = rnd((L, 64), 1.0, seed=32), rnd((L, 64), 1.0, seed=33), rnd((L, 64), 1.0, seed=34)
    for t in range(5):
        k[t * 64 + 7] = q[3] * (2.0 + t)
    o, lse = ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
    ro, rlse = ref.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
    assert maxerr(lse, rlse) < 2e-2 and relerr(o, ro) < 1e-2


@pytest.mark.parametrize("V,ld", [(51866, 51968), (1000, 1000), (51864, 51864)])
def test_distill_loss(ops, ref, V, ld):
    rows = 97
    s = rnd((rows, ld), 2.0, seed=40)
    t = rnd((rows, ld), 2.0, seed=41)
    labels = torch.randint(0, V, (rows,), device="cuda")
    labels[::5] = -100
    s_ref, s_hip = s.clone(), s.clone()
    rl = ref.distill_loss(s_ref, t, labels, V, 2.0, 0.8, 1.0, 1.0, True)
    hl = ops.distill_loss(s_hip, t, labels, V, 2.0, 0.8, 1.0, 1.0, True)
    assert relerr(hl[:3], rl[:3]) < 2e-5, (hl.tolist(), rl.tolist())
    assert hl[3].item() == rl[3].item()
    assert relerr(s_hip[:, :V], s_ref[:, :V]) < 1e-2
    assert s_hip[:, V:].abs().max().item() == 0 if ld > V else True
    hl2 = ops.distill_loss(s.clone(), t, labels, V, 2.0, 0.8, 1.0, 1.0, False)
    assert torch.equal(hl2, hl)


@pytest.mark.parametrize("mfma", [1, 0])
def test_logmel(ops, ref, mfma):
    """Both builds of the front end -- the folded DFT on the fp32 matrix pipe (default) and the direct DFT on the VALU
    -- against the float64 restatement: noise, a zero-padded clip (1e-10 clamp, max-8 clip), a ramped clip and a tonal
    clip with a 60 dB dynamic range inside every frame (weak bins next to strong ones need full fp32 accuracy)."""
    from transformers.audio_utils import mel_filter_bank
    g = torch.Generator().manual_seed(0)
    audio = (0.1 * torch.randn(5, 480000, generator=g))
    audio[2, 200000:] = 0.0  # a padded clip: exercises the 1e-10 clamp and the max-8 clip
    audio[3] *= torch.linspace(0.0, 1.0, 480000) ** 2
    t = torch.arange(480000, dtype=torch.float64) / 16000.0
    audio[4] = (0.3 * torch.sin(2 * torch.pi * 440.0 * t) + 0.05 * torch.sin(2 * torch.pi * 3100.0 * t + 1.0) +
                3e-4 * torch.sin(2 * torch.pi * 6050.0 * t)).float()
    audio = audio.cuda()
    try:
        ops.lib.dw_debug_set(5, mfma)
        for M in (80, 128):
            filt = torch.tensor(mel_filter_bank(201, M, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney"),
                                dtype=torch.float32).cuda().contiguous()
            out = ops.logmel(audio, filt)
            r = ref.logmel(audio, filt)
            assert out.shape == (5, M, 3000)
            assert maxerr(out, r) < 1e-4, (M, maxerr(out, r))
    finally:
        ops.lib.dw_debug_set(5, 1)


def test_embedding(ops, ref):
    B, T, D, V = 3, 447, 384, 1000
    ids = torch.randint(0, V, (B, T), device="cuda")
    tok, pos = rnd((V, D), 1.0, torch.float32, seed=50), rnd((448, D), 1.0, torch.float32, seed=51)
    assert torch.equal(ops.embed_fwd(ids, tok, pos, torch.float32), ref.embed_fwd(ids, tok, pos, torch.float32))
    tb, pb = tok.bfloat16(), pos.bfloat16()
    assert maxerr(ops.embed_fwd(ids, tb, pb, torch.bfloat16), ref.embed_fwd(ids, tb, pb, torch.bfloat16)) == 0
    dx = rnd((B * T, D), 1.0, torch.float32, seed=52)
    dt, dp = torch.zeros(V, D, device="cuda"), torch.zeros(448, D, device="cuda")
    rdt, rdp = torch.zeros(V, D, device="cuda"), torch.zeros(448, D, device="cuda")
    ops.embed_bwd(dx, ids, dt, dp)
    ref.embed_bwd(dx, ids, rdt, rdp)
    assert relerr(dt, rdt) < 1e-5 and relerr(dp, rdp) < 1e-5


def test_conv_helpers(ops, ref):
    B, Cm, T, D = 2, 80, 3000, 384
    mel = rnd((B, Cm, T), 1.0, torch.float32, seed=60)
    assert torch.equal(ops.im2col_mel(mel, 256), ref.im2col_mel(mel, 256))
    a = rnd((B * T, D), 1.0, seed=61)
    assert torch.equal(ops.im2col_s2(a, B, T), ref.im2col_s2(a, B, T))
    dxcol = rnd((B * T // 2, 3 * D), 1.0, seed=62)
    z = rnd((B * T, D), 1.0, seed=63)
    assert relerr(ops.col2im_s2_gelu_bwd(dxcol, z, B, T), ref.col2im_s2_gelu_bwd(dxcol, z, B, T)) < 3e-3
    w = rnd((D, Cm, 3), 1.0, torch.float32, seed=64)
    assert torch.equal(ops.pack_conv_weight(w, 256), ref.pack_conv_weight(w, 256))
    gwp = rnd((D, 256), 1.0, torch.float32, seed=65)
    g1, g2 = torch.ones(D, Cm, 3, device="cuda"), torch.ones(D, Cm, 3, device="cuda")
    ops.unpack_conv_grad(gwp, g1, True)
    ref.unpack_conv_grad(gwp, g2, True)
    assert torch.equal(g1, g2)


def test_conv_as_gemm_matches_conv1d(ops):
    """im2col + GEMM reproduces F.conv1d(k=3,p=1) + GELU of the reference front end (stride 1 and 2)."""
    import torch.nn.functional as F
    B, Cm, T, D = 2, 80, 3000, 384
    mel = rnd((B, Cm, T), 1.0, torch.float32, seed=66)
    w1, b1 = rnd((D, Cm, 3), 0.05, torch.float32, seed=67), rnd((D,), 0.1, torch.float32, seed=68)
    w2, b2 = rnd((D, D, 3), 0.03, torch.float32, seed=69), rnd((D,), 0.1, torch.float32, seed=70)
    x1 = ops.im2col_mel(mel, 256)
    a1 = ops.gemm(x1, ops.pack_conv_weight(w1, 256), bias=b1, act=1)
    r1 = F.gelu(F.conv1d(mel.bfloat16().float(), w1.bfloat16().float(), b1, padding=1).bfloat16().float())
    assert relerr(a1.reshape(B, T, D).transpose(1, 2), r1) < 6e-3
    x2 = ops.im2col_s2(a1, B, T)
    a2 = ops.gemm(x2, ops.pack_conv_weight(w2, 3 * D), bias=b2, act=1)
    r2 = F.gelu(F.conv1d(a1.reshape(B, T, D).transpose(1, 2).float(), w2.bfloat16().float(), b2, stride=2,
                         padding=1).bfloat16().float())
    assert relerr(a2.reshape(B, T // 2, D).transpose(1, 2), r2) < 6e-3


def test_small_streaming(ops, ref):
    x = rnd((1000, 384), 1.0, torch.float32, seed=80)
    assert torch.equal(ops.cast_bf16(x), x.bfloat16())
    assert torch.equal(ops.cast_f32(x.bfloat16()), x.bfloat16().float())
    xb = rnd((1003, 1280), 1.0, seed=81)
    out = torch.ones(1280, device="cuda")
    ops.colsum(xb, out, True)
    assert relerr(out, 1.0 + xb.float().sum(0)) < 1e-5
    ops.colsum(xb[:, 256:512], out[:256], False)
    assert relerr(out[:256], xb[:, 256:512].float().sum(0)) < 1e-5
    y = ops.add(x, x.bfloat16(), torch.bfloat16)
    assert torch.equal(y, (x + x.bfloat16().float()).bfloat16())


@pytest.mark.parametrize("M", [1024, 520, 2000])
def test_gemm_column_sums_of_the_output_in_the_epilogue(ops, ref, M):
    """DwGemm.colsum_out: the dX GEMM of fc2 (k-major B, x gelu'(z) epilogue) also ADDS the column sums of its bf16
    output to the fc1.bias gradient -- interior tiles through the flavoured walk, ragged last row tile through the
    general one; also with the plain bf16 and the fp32-output epilogues and on the 128-tile kernel."""
    N, K = 768, 256
    dy = rnd((M, K), 1.0, torch.bfloat16, seed=61)
    w = rnd((K, N), 0.05, torch.bfloat16, seed=62)
    g = rnd((M, N), 0.5, torch.float16, seed=63)
    for kw in (dict(zgrad=g), dict(), dict(out_dtype=torch.float32)):
        for tile in (256, 128):
            acc = torch.full((N,), 0.25, device="cuda")
            out = ops.gemm(dy, w, trans_b=True, colsum=acc, tile=tile, **kw)
            want = 0.25 + out.float().sum(0)
            assert relerr(acc, want) < 2e-5, (kw.keys(), tile)
            r_acc = torch.zeros(N, device="cuda")
            r_out = ref.gemm(dy, w, trans_b=True, colsum=r_acc, **kw)
            assert relerr(out, r_out) < 1e-2 and relerr(acc - 0.25, r_acc) < 2e-2


def test_adamw_and_clip(ops, ref):
    n = 1_000_003
    p = rnd((n,), 1.0, torch.float32, seed=90)
    g = rnd((n,), 0.01, torch.float32, seed=91)
    st = {}
    for name, o in (("hip", ops), ("ref", ref)):
        pp, m, v = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
        sh = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        for step in (1, 2, 3):
            ss = torch.zeros(1, device="cuda")
            o.sumsq(g, ss)
            o.adamw(pp, g, m, v, sh, ss, 1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
        st[name] = (pp, m, v, sh, ss)
    assert relerr(st["hip"][4], st["ref"][4]) < 1e-5
    for i in range(3):
        assert relerr(st["hip"][i], st["ref"][i]) < 1e-5, i
    assert relerr(st["hip"][3], st["ref"][3]) < 1e-3


def test_adamw_device_state_equals_host_scalars(ops, ref):
    """dw_adam_tick + dw_adamw_dev (optimizer scalars resident on the device: lr, step count, betas -- what lets a HIP
    graph replay the update) against dw_adamw with the host's scalars: same parameters, moments, shadow after three
    steps with an LR change in between; a closed gate (n_valid = 0) leaves everything, the step count included, as is."""
    n = 300_007
    p = rnd((n,), 1.0, torch.float32, seed=92)
    g = rnd((n,), 0.01, torch.float32, seed=93)
    ss = torch.zeros(1, device="cuda")
    ops.sumsq(g, ss)
    lrs = (1e-3, 1e-3, 5e-4)
    pa, ma, va = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    sa = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for step, lr in zip((1, 2, 3), lrs):
        ops.adamw(pa, g, ma, va, sa, ss, 1.0, 0.5, lr, 0.9, 0.999, 1e-8, 0.01, step)
    for o in (ops, ref):
        pb, mb, vb = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
        sb = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        state = o.adam_state(lrs[0], 0.9, 0.999, 0)
        open_gate, closed = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
        for i, lr in enumerate(lrs):
            state[0:1].fill_(lr)
            o.adam_tick(state, open_gate if i else None)
            o.adamw_dev(pb, g, mb, vb, sb, ss, 1.0, 0.5, state, 1e-8, 0.01)
            if i == 1:
                before = pb.clone()
                o.adam_tick(state, closed)
                o.adamw_dev(pb, g, mb, vb, sb, ss, 1.0, 0.5, state, 1e-8, 0.01)
                assert torch.equal(pb, before) and float(state[1]) == 2.0 and float(state[6]) == 0.0
        assert float(state[1]) == 3.0
        tol = 0.0 if o is ops else 1e-5
        for x, y in ((pa, pb), (ma, mb), (va, vb)):
            assert relerr(y, x) <= tol
        assert relerr(sb, sa) <= (0.0 if o is ops else 1e-3)


def test_greedy_select_matches_restatement(ops, ref):
    """csrc/decode.hip (logits processors + argmax + EOS bookkeeping in one launch) against the torch restatement whose
    timestamp rules are pinned against transformers' WhisperTimeStampLogitsProcessor (tests/test_longform.py): random
    logits and histories, every rule combination, exact token equality (ties are broken towards the smaller id)."""
    V, ld, eos, no_ts = 1000, 1024, 900, 911
    tb = no_ts + 1
    g = torch.Generator().manual_seed(0)
    sup = torch.zeros(V, dtype=torch.uint8)
    sup[torch.randint(0, V, (150,), generator=g)] = 1
    bsup = torch.zeros(V, dtype=torch.uint8)
    bsup[[220, eos]] = 1
    for trial in range(120):
        B, begin = 6, 3
        n = begin + int(torch.randint(0, 9, (1,), generator=g))
        toks = torch.randint(0, eos, (B, 24), generator=g)
        for b in range(B):                       # sprinkle non-decreasing timestamps into the generated part
            curts = tb
            for j in range(begin, n):
                if float(torch.rand(1, generator=g)) < 0.45:
                    curts = min(V - 1, curts + int(torch.randint(0, 4, (1,), generator=g)))
                    toks[b, j] = curts
        logits = (torch.randn(B, ld, generator=g) * (3.0 if trial % 2 else 0.3))
        if trial % 3 == 0:
            logits[:, tb:] += 2.0                # the "timestamps together beat the best text token" branch
        logits = logits.to(torch.bfloat16)
        done0 = torch.rand(B, generator=g) < 0.2
        kw = dict(suppress=sup if trial % 4 else None, begin_suppress=bsup, first=(n == begin), no_eos=bool(trial % 5 == 0),
                  forced=bool(trial % 17 == 16), ts_begin=tb if trial % 2 == 0 else -1, max_initial=7 if trial % 4 < 2 else -1,
                  begin_index=begin, eos=eos, fill=eos if trial % 7 else 0)
        out = {}
        for name, o in (("hip", ops), ("ref", ref)):
            t = toks.clone().cuda()
            cur = torch.zeros(B, 1, dtype=torch.long, device="cuda")
            done = done0.clone().cuda()
            k2 = dict(kw)
            for m in ("suppress", "begin_suppress"):
                k2[m] = None if k2[m] is None else k2[m].cuda()
            o.greedy_select(logits.cuda(), V, t, n, cur, done=done, **k2)
            out[name] = (t.cpu(), cur.cpu(), done.cpu())
        for a, b in zip(out["hip"], out["ref"]):
            assert torch.equal(a, b), (trial, kw, out["hip"][1].view(-1).tolist(), out["ref"][1].view(-1).tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,cols", [(torch.bfloat16, 1280), (torch.float32, 1280), (torch.bfloat16, 3840), (torch.int32, 4)])
def test_move_rows_gather_scatter(dtype, cols):
    """dw_move_rows: rows through an index list, both directions, strided source / destination (bit-exact)."""
    from distil_whisper_amd.ops_hip import HipOps
    ops = HipOps("cuda:0")
    g = torch.Generator().manual_seed(7)
    n_src, n = 7136, 4103
    src = (torch.randn(n_src, cols, generator=g) * 100).to(dtype).cuda()
    idx = torch.randperm(n_src, generator=g)[:n].sort().values.to(torch.int32).cuda()
    out = torch.zeros(n + 5, cols, dtype=dtype, device="cuda")
    ops.gather_rows(src, idx, out)
    assert torch.equal(out[:n], src[idx.long()]) and float(out[n:].float().abs().max()) == 0.0
    wide = (torch.randn(n_src, 3 * cols, generator=g) * 100).to(dtype).cuda()       # a column block of a wider matrix
    ops.gather_rows(wide[:, cols:2 * cols], idx, out)
    assert torch.equal(out[:n], wide[idx.long(), cols:2 * cols])
    back = torch.full((n_src, cols), 7, dtype=dtype, device="cuda")
    ops.scatter_rows(out, idx, back)
    ref = torch.full((n_src, cols), 7, dtype=dtype, device="cuda")
    ref[idx.long()] = out[:n]
    assert torch.equal(back, ref)
