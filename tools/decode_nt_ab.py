"""Token step of the distil-large-v3 student (batch 16, graphs) and of the 32-layer teacher with the streamed-once loads carrying
the non-temporal hint (-DDW_DECODE_NT=1 GEMV weights, 2 attention K / V, 3 both; variant libraries built by
tools/build_variant_lib.sh) against the default build, same process, interleaved: ms per decode step."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd import ops_hip
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.modeling import WhisperForConditionalGeneration
from distil_whisper_amd.decoding import GreedyDecoder
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
here = os.path.dirname(ops_hip.LIB_PATH)
libs = {"default": ops.lib}
for m in (1, 2, 3):
    p = os.path.join(here, f"libdwamd_nt{m}.so")
    if os.path.exists(p):
        libs[f"nt{m}"] = ops_hip.load_library(p)
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
B, NEW = 16, 128
res = {}
for tag, dims, sd, kw in (("student_2_layers", sdims, s_sd, {}), ("teacher_32_layers", tdims, t_sd, {"dtype": torch.bfloat16})):
    model = WhisperForConditionalGeneration(dims, ops=ops, state_dict=sd, **kw)
    enc, _ = model.engine.encode(torch.randn(B, 128, 3000, device=dev) * 0.5, save=False)
    prompt = torch.full((B, 1), dims.decoder_start_token_id, dtype=torch.long, device=dev)
    t = {k: [] for k in libs}
    toks = {}
    for rnd in range(4):
        for name, lib in libs.items():
            ops.lib = lib
            dec = GreedyDecoder(model.engine, B, 1 + NEW, use_graphs=False)
            dec.run(enc, prompt, NEW); torch.cuda.synchronize()
            t0 = time.perf_counter(); out = dec.run(enc, prompt, NEW); torch.cuda.synchronize()
            t[name].append((time.perf_counter() - t0) / NEW * 1e3)
            toks[name] = out
    ops.lib = libs["default"]
    res[tag] = {k: {"ms_per_decode_step": sorted(v)[len(v) // 2], "same_tokens": bool(torch.equal(toks[k], toks["default"]))} for k, v in t.items()}
    del model
print(json.dumps(res, indent=1))
