"""Build libdwamd.so (the C-ABI HIP library, see include/dwamd.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdwamd.so")
SOURCES = ["gemm.hip", "gemm_tile256.hip", "gemm_tile128.hip", "gemm_tile128w4.hip", "gemm_phased.hip", "gemm_wp8_nn.hip", "gemm_wp8_nn_ref.hip", "gemm_wp8_dbg.hip", "gemm_wp8_m320.hip", "gemm_wp8_m128.hip", "gemm_wp8_nt.hip", "gemm_wp8_tt.hip", "gemm_wp16_nn.hip", "gemm_wp16_nt.hip", "gemm_wp16_small.hip", "gemm_wp16_tt.hip", "gemm_wp16_w4.hip", "gemm_skinny.hip", "attention.hip", "norm.hip", "loss.hip", "logmel.hip", "elementwise.hip", "optim.hip", "decode.hip"]
# -munsafe-fp-atomics: float atomicAdd becomes the hardware global_atomic_add_f32 instead of a CAS loop
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics"]


def kernels_sha16() -> str:
    """Identity of the kernel sources (every file under csrc/ + the C header): measurements that depend on the kernels
    (profiles/pmc_traffic.json, this library's side of the vendor calibration) carry it, and bench.py only quotes a
    committed measurement whose hash is the current one."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(HERE, "..", "include", "dwamd.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_kernel.h"), os.path.join(CSRC, "gemm_wp.h"), os.path.join(CSRC, "gemm_wp16.h"),
               os.path.join(HERE, "..", "include", "dwamd.h")]
    objs, procs = [], []
    flags = FLAGS + (["-DDW_ABLATE"] if os.environ.get("DW_ABLATE") else [])   # timing-experiment kernels (tools/attn_ablate.py)
    flags += os.environ.get("DW_EXTRA_FLAGS", "").split()                      # -D switches of A/B builds (tools/build_variant_lib.sh)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


# NOTE for callers that dlopen the result: import torch BEFORE loading libdwamd.so.  torch bundles its own copy of
# libamdhip64; loading this library first would pull a second HIP runtime into the process and every launch would
# fail with hipErrorNoDevice (100).  ops_hip.load_library() does this in the right order.

if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
