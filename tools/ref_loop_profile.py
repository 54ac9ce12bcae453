"""The reference's loop body over the drop-in modules (bench.py's via_reference_loop legs), a few steps: target of
rocprofv3 --kernel-trace --stats, to see which launches the module surface adds to the native trainer's step.
LEG: verbatim | skip | full (valid_len + fused KD loss + FusedAdamW)."""
import functools, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd import modeling as M
from distil_whisper_amd import student_init as si
from distil_whisper_amd.optim import FusedAdamW
from oracle.reference_loop import ReferenceLoop
dev = "cuda:0"
ops = HipOps(dev)
LEG = os.environ.get("LEG", "full")
FusedAdamW.merge_segments = os.environ.get("MERGE", "1") != "0"
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, seed=0, device=dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
student = M.WhisperForConditionalGeneration(sdims, ops=ops, state_dict=s_sd)
teacher = M.WhisperForConditionalGeneration(tdims, ops=ops, state_dict=t_sd, dtype=torch.bfloat16)
del t_sd, s_sd
filt = torch.tensor(si.mel_filter_bank(128), dtype=torch.float32, device=dev).contiguous()
B, T = 32, 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = 50258
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone()
lens = torch.randint(32, 225, (B,), generator=torch.Generator().manual_seed(1234)).tolist()
labels[torch.arange(T, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]] = -100
student.skip_dead_positions = teacher.skip_dead_positions = LEG == "skip"
loop = ReferenceLoop(student, teacher, M.BaseModelOutput, share_hidden_states=False, teacher_dtype=torch.bfloat16,
                     fused_loss=M.fused_distillation_loss if LEG == "full" else None,
                     optimizer_cls=functools.partial(FusedAdamW, model=student) if LEG == "full" else None)
def one():
    feats = ops.logmel(audio, filt)
    batch = {"input_features": feats, "decoder_input_ids": dec_in, "labels": labels}
    if LEG == "full": batch["valid_len"] = lens
    return loop.training_iteration(batch, temperature=2.0)
for _ in range(2): one()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = int(os.environ.get("N", 4))
s.record()
for _ in range(N): one()
e.record(); torch.cuda.synchronize()
print(f"LEG={LEG}: {s.elapsed_time(e) / N:.1f} ms per step", flush=True)
