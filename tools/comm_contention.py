"""What a communication kernel resident on a few CUs does to the persistent GEMM kernels (emulation on one GPU).
A hog kernel (tools/cu_hog.hip: nwg workgroups that each take a CU exclusively and spin) runs on a side stream for the
whole step; the step is timed with the persistent 256-tile kernels sized for all 256 CUs or for fewer (dw_debug_set 9)."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
import subprocess
_here = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(_here, "libcuhog.so")):       # hipcc cross-compiles; build it before going to the GPU box
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(_here, "cu_hog.hip"), "-o", os.path.join(_here, "libcuhog.so")])
hog = ctypes.CDLL(os.path.join(_here, "libcuhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
filt = torch.tensor(si.mel_filter_bank(128), dtype=torch.float32, device=dev).contiguous()
tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, mel_filters=filt)
del t_sd, s_sd
B, T = 32, 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = 50258
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone(); labels[:, 200:] = -100
sink = torch.zeros(4, device=dev)
side = torch.cuda.Stream()
def step(): return tr.train_step(tr.features(audio), dec_in, labels)
for _ in range(2): step()
torch.cuda.synchronize()
for nwg in (0, 8, 16, 32):
    for cus, dyn in ((256, 0), (256, 1), (248, 1)):
        ops.lib.dw_debug_set(9, cus); ops.lib.dw_debug_set(10, dyn)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            if nwg:
                for _ in range(120):       # ~600 ms of occupancy, queued ahead on the side stream
                    hog.hog_launch(nwg, 5000, sink.data_ptr(), ctypes.c_void_p(side.cuda_stream))
            time.sleep(0.01)
            t0 = time.perf_counter(); step(); torch.cuda.current_stream().synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        print(f"hog {nwg:2d} CUs, GEMM grid {cus}, dynamic hand-out {dyn}: step {sorted(ts)[1]:.1f} ms", flush=True)
ops.lib.dw_debug_set(9, 256); ops.lib.dw_debug_set(10, 1)
