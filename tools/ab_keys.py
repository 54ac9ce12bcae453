"""In-process A/B of dw_debug_set keys on the full distillation step (bench.py's batch: label lengths U{32..224}, packed live
rows), interleaved rounds.  DW_AB = python list of configs, each a dict {key: value, ..., "lib": index}; keys not named keep
the library default.  `lib` > 0 runs the config on distil_whisper_amd/libdwamd_base[N].so (another build on the same box).
DW_STREAMS=1: teacher / weight-gradient side streams (bench.py's eager_side_streams).  String keys are Python-level switches:
"varlen" (WhisperEngine.varlen_attention), "pack" (WhisperEngine.pack_train_layers), "overwrite" (DistillationTrainer.overwrite_wgrad); default 1."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd import ops_hip as _oh
from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
MODEL = os.environ.get("MODEL", "large-v3")            # large-v3 | small.en | tiny.en
tdims = si.PRESETS[MODEL]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, *si.STUDENT_LAYERS[MODEL])
filt = torch.tensor(si.mel_filter_bank(tdims.n_mels), dtype=torch.float32, device=dev).contiguous()
tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, mel_filters=filt)
del t_sd, s_sd
B, T = 32, 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = tdims.decoder_start_token_id
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone()
lens = torch.randint(32, 225, (B,), generator=torch.Generator().manual_seed(1234)).tolist()
labels[torch.arange(T, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]] = -100
if int(os.environ.get("DW_STREAMS", "0")):
    tr.set_overlap_wgrad(True); tr.overlap_teacher = True
def step():
    return tr.train_step(tr.features(audio), dec_in, labels, valid_len=lens)
configs = eval(os.environ.get("DW_AB", "[{}, {25: 0}]"))
defaults = {}
libs = {0: ops.lib}
base = os.path.join(os.path.dirname(_oh.LIB_PATH), "libdwamd_base.so")
for i, path in [(1, base)] + [(j, base.replace("_base.so", f"_base{j}.so")) for j in (2, 3, 4, 5)]:
    if os.path.exists(path): libs[i] = _oh.load_library(path)
allkeys = sorted({k for c in configs for k in c if isinstance(k, int)})
from distil_whisper_amd.engine import WhisperEngine
DEF = {0: 2163, 1: 0, 3: 5, 6: 4, 9: 256, 10: 1, 11: 1, 12: 0, 20: 36, 22: 1, 23: 8, 24: 0, 25: 1, 26: 0, 27: 0, 28: 0}
step(); torch.cuda.synchronize()
res = [[] for _ in configs]
NS = int(os.environ.get("DW_NS", "3"))
for r in range(int(os.environ.get("DW_ROUNDS", "4"))):
    for ci, c in enumerate(configs):
        ops.lib = libs[c.get("lib", 0)]
        for k in allkeys: ops.lib.dw_debug_set(k, c.get(k, DEF.get(k, 0)))
        WhisperEngine.varlen_attention = bool(c.get("varlen", 1))
        WhisperEngine.pack_train_layers = bool(c.get("pack", 1))
        WhisperEngine.fuse_attn_bias_grad = bool(c.get("attn_bias", 0))     # q / v bias gradients from the attention-backward kernels
        tr.overwrite_wgrad = bool(c.get("overwrite", 1))
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NS): step()
        torch.cuda.synchronize()
        res[ci].append((time.perf_counter() - t0) / NS * 1e3)
for c, r in zip(configs, res):
    print(c, "ms/step:", " ".join(f"{x:.1f}" for x in r), "median", f"{sorted(r)[len(r) // 2]:.2f}", flush=True)
