"""Reads the generated code of the hot loops -- the check that found every gain of round 4's second half.

    python tools/isa_lint.py [translation units, default: the GEMM main loops, attention, norm]

Per kernel: registers / scratch, flat memory instructions (a pointer that may be LDS or global is a GENERIC pointer: every
access is a flat_load waited for with vmcnt(0) AND lgkmcnt(0)), generic -> LDS pointer conversions (src_shared_base), and per
innermost loop that holds MFMAs: instruction census, scratch traffic (a scratch reload is a VMEM load: its vmcnt wait also
waits for every global store / operand load in flight), vmcnt(0) waits, 64-bit lane arithmetic.  Cross-compiles for gfx950;
needs no GPU."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distil_whisper_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
         "-I", os.path.join(ROOT, "include")]
DEFAULT = ["gemm_wp8_m320", "gemm_wp8_nn", "gemm_wp8_nt", "gemm_wp16_nn", "gemm_wp16_tt", "attention", "norm"]


def kernels(asm):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for s in starts:
        e = next((i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end")), None)
        if e is not None:
            yield lines[s].split(":")[0], lines[s:e]


def census(seg):
    c = collections.Counter(x.strip().split(" ")[0] for x in seg if x.strip() and not x.strip().startswith((";", ".")))
    n = sum(c.values())
    return n, {"mfma": sum(v for k, v in c.items() if "mfma" in k), "valu": sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k),
               "salu": sum(v for k, v in c.items() if k.startswith("s_")), "ds": sum(v for k, v in c.items() if k.startswith("ds_")),
               "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))), "scratch": sum(v for k, v in c.items() if k.startswith("scratch_")),
               "flat": sum(v for k, v in c.items() if k.startswith("flat_")), "u64": c["v_mad_u64_u32"] + c["v_lshl_add_u64"],
               "vmcnt0": sum(1 for x in seg if re.search(r"s_waitcnt.*vmcnt\(0\)", x))}


def loops(body):
    """Innermost loops that hold MFMAs.  A loop may be rotated (its latch block sits in front of the header label): the loop runs
    from the closest back-edge target at or before an `Inner Loop Header` label to the last branch to that target."""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    def is_header(i):        # (the remark is on the label's line, or on one of the comment lines right under it when the loop is nested)
        if not re.match(r"^\.LBB\d+_\d+:", body[i]):
            return False
        j = i
        while True:
            if "Inner Loop Header" in body[j]:
                return True
            j += 1
            if j >= len(body) or j > i + 6 or not body[j].lstrip().startswith(";"):
                return False
    headers = [i for i in range(len(body)) if is_header(i)]
    back = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            back.append((i, labels[m.group(1)]))
    seen = set()
    for h in headers:
        nxt = min([x for x in headers if x > h], default=len(body))
        cand = [(i, t) for i, t in back if h < i and t <= h and i < nxt + 4000]
        if not cand:
            continue
        t0 = max(t for _, t in cand)
        end = max(i for i, t in cand if t == t0)
        if (t0, end) in seen:
            continue
        seen.add((t0, end))
        seg = body[t0:end]
        if any("v_mfma" in x for x in seg):
            yield t0, end, seg


def main():
    tus = sys.argv[1:] or DEFAULT
    for tu in tus:
        src = os.path.join(CSRC, tu + ".hip")
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", src, "-o", out], capture_output=True, text=True, cwd=CSRC)
            if r.returncode:
                print(tu, "FAILED", r.stderr[-400:]); continue
            res = {}
            name = None
            for l in r.stderr.split("\n"):
                m = re.search(r"Function Name: (\S+)", l)
                if m: name = m.group(1); res[name] = {}
                for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]"):
                    m = re.search(key + r": (\d+)", l)
                    if m and name: res[name][key.split(" ")[0].replace("\\", "")] = int(m.group(1))
            asm = open(out).read()
        print(f"== {tu}")
        for kname, body in kernels(asm):
            n, c = census(body)
            demangled = subprocess.run(["c++filt", kname], capture_output=True, text=True).stdout.strip()[:110]
            r_ = res.get(kname, {})
            flag = " <-- FLAT" if c["flat"] else ""
            print(f"  {demangled}: {n} instr, VGPR {r_.get('VGPRs')} AGPR {r_.get('AGPRs')} scratch {r_.get('ScratchSize')} B, occupancy {r_.get('Occupancy')}; "
                  f"flat {c['flat']}, src_shared_base {sum('src_shared_base' in x for x in body)}, scratch ops {c['scratch']}{flag}")
            for a, b, seg in loops(body):
                n, c = census(seg)
                warn = " <-- scratch in the loop" if c["scratch"] else ""
                print(f"      loop @{a}-{b}: {n} instr: mfma {c['mfma']} valu {c['valu']} salu {c['salu']} ds {c['ds']} vmem {c['vmem']} "
                      f"scratch {c['scratch']} vmcnt(0) {c['vmcnt0']} 64-bit lane ops {c['u64']}{warn}")


if __name__ == "__main__":
    main()
