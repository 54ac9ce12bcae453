"""In-process A/B of GEMM tuning knobs on the full distillation step (interleaved rounds)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
filt = torch.tensor(si.mel_filter_bank(128), dtype=torch.float32, device=dev).contiguous()
tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, mel_filters=filt)
del t_sd, s_sd
B, T = 32, 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = 50258
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone()
# DW_LENS=1: the bench's label lengths (U{32..224}) and the step over the live decoder rows only (bench.py's default)
if os.environ.get("DW_LENS"):
    lens = torch.randint(32, 225, (B,), generator=torch.Generator().manual_seed(1234)).tolist()
    labels[torch.arange(T, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]] = -100
else:
    lens = None
    labels[:, 200:] = -100
def step():
    return tr.train_step(tr.features(audio), dec_in, labels, valid_len=lens)
# (GEMM variant, strip[, attention-backward mode]); variant 3 = default, 4 = phase-pipelined kernel; strip 0 = auto rule;
# attention mode = dw_debug_set key 3 (default 1)
configs = eval(os.environ.get("DW_AB", "[(115,0,5),(2163,0,5)]"))
# optional 13th field: 1 = run the config on distil_whisper_amd/libdwamd_base.so (a build of another commit, copied there by
# hand) instead of libdwamd.so -- the only way to compare two BUILDS on one box (boxes of the pool differ by up to 10 %)
import ctypes
from distil_whisper_amd import ops_hip as _oh
_libs = {0: ops.lib}
_base = os.path.join(os.path.dirname(_oh.LIB_PATH), "libdwamd_base.so")
if os.path.exists(_base):
    _libs[1] = _oh.load_library(_base)
for _i in (2, 3, 4, 5):
    if os.path.exists(_base.replace("_base.so", f"_base{_i}.so")):
        _libs[_i] = _oh.load_library(_base.replace("_base.so", f"_base{_i}.so"))
step(); torch.cuda.synchronize()
res = {c: [] for c in configs}
for r in range(4):
    for c in configs:
        ops.lib = _libs[c[12] if len(c) > 12 else 0]
        ops.lib.dw_debug_set(0, c[0]); ops.lib.dw_debug_set(1, c[1]); ops.lib.dw_debug_set(3, c[2] if len(c) > 2 else 5); ops.lib.dw_debug_set(6, c[3] if len(c) > 3 else 8); ops.lib.dw_debug_set(10, c[4] if len(c) > 4 else 1)
        ops.lib.dw_debug_set(11, c[6] if len(c) > 6 else 1)
        ops.lib.dw_debug_set(9, c[10] if len(c) > 10 else 256)             # CUs the persistent GEMM grids occupy
        ops.lib.dw_debug_set(12, c[11] if len(c) > 11 else 0)              # start offsets of the persistent workgroups
        type(tr.student).ffn_keeps_gelu_grad = bool(c[5]) if len(c) > 5 else True
        type(tr.student).fuse_fc1_bias_grad = bool(c[13]) if len(c) > 13 else True   # fc1.bias gradient in the dX-fc2 GEMM epilogue
        type(tr.student).fuse_attn_bias_grad = bool(c[14]) if len(c) > 14 else False  # q / v bias gradients in the attention backward
        ops.lib.dw_debug_set(20, c[15] if len(c) > 15 else 36)             # v_mfma_f32_16x16x32_bf16 main loops (bit mask NN / NT / TT)
        type(tr.student).ffn_row_pad = type(tr.teacher).ffn_row_pad = c[16] if len(c) > 16 else 64   # row pitch pad of the FFN-wide buffers
        type(tr.student).row_pad = type(tr.teacher).row_pad = c[17] if len(c) > 17 else 64
        type(tr.student).stream_row_pad = type(tr.teacher).stream_row_pad = c[18] if len(c) > 18 else 128
        tr.set_overlap_wgrad(bool(c[7]) if len(c) > 7 else False)          # weight-gradient GEMMs on a second stream
        tr.overlap_teacher = bool(c[8]) if len(c) > 8 else False           # teacher forward on a second stream
        tr.teacher.pad_gemm_rows = bool(c[9]) if len(c) > 9 else False     # teacher decoder GEMMs over M padded to 320 rows
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): step()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / 2 * 1e3)
for c in configs:
    print(c, "ms/step:", " ".join(f"{x:.1f}" for x in res[c]), "median", sorted(res[c])[len(res[c]) // 2])
