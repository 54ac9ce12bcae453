"""HBM-side traffic per launch of the bench's kernels from two rocprofv3 --pmc passes of the SAME bench command.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/pmc_traffic.py <fetch results.db> <write results.db> profiles/pmc_traffic.json

FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots).  Units and corrections follow
/opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
wide (16 B/lane) streaming reads, so it is doubled; WRITE_SIZE was calibrated in round 1 against a GEMM whose output
bytes are known exactly (profiles/r1_gemm_pmc.md: 480 000 KiB = M*N*2) and is taken as is.  The counters sit on the
fabric side of L2, so Infinity-Cache (MALL) hits are included: this is traffic leaving the XCD L2s, an upper bound of
DRAM traffic.  bench.py reads the JSON this writes and fills roofline.traffic for the kernel class it reports.
"""
import json
import os
import re
import sqlite3
import sys

CLASS_OF = [  # kernel symbol -> the per-class key bench.py / ops_hip.py use
    (r"gemm_kernel<256, 256, \d+, \d+, false, false", "gemm_t256_NN"), (r"gemm_kernel<256, 256, \d+, \d+, false, true", "gemm_t256_NT"),
    (r"gemm_kernel<256, 256, \d+, \d+, true, false", "gemm_t256_TN"), (r"gemm_kernel<256, 256, \d+, \d+, true, true", "gemm_t256_TT"),
    (r"gemm_kernel<128, 128, \d+, \d+, false, false", "gemm_t128_NN"), (r"gemm_kernel<128, 128, \d+, \d+, false, true", "gemm_t128_NT"),
    # round 6: the small-M rule's kernels (outputs below two rounds of 256-row tiles; ops_hip.py keys those launches gemm_t128_*):
    # the 128 x 256 tile in a three-stage ring and the 256-row 16x16x32 kernels under their second symbol (TAG = 1)
    (r"gemm_wp_kernel<false, false, 2, 4, true, 0, 128, 3>", "gemm_t128_NN"), (r"gemm_wp_kernel<false, true, 2, 4, true, 0, 128, 3>", "gemm_t128_NT"),
    (r"gemm_wp16_kernel<false, false, 256, 0, 4, 1>", "gemm_t128_NN"), (r"gemm_wp16_kernel<false, true, 256, 0, 4, 1>", "gemm_t128_NT"),
    (r"gemm_wp_kernel<false, false", "gemm_t256_NN"), (r"gemm_wp_kernel<false, true", "gemm_t256_NT"),
    (r"gemm_wp_kernel<true, true", "gemm_t256_TT"), (r"gemm_wp_kernel<true, false", "gemm_t256_TN"),
    # the 16x16x32 main loops (gemm_wp16.h): same class keys as the 32x32x16 kernels they replace per shape -- round 4 added the
    # kernels and not these patterns, so a third of the NN class and the whole weight-gradient class had no counter bytes
    (r"gemm_wp16_kernel<false, false", "gemm_t256_NN"), (r"gemm_wp16_kernel<false, true", "gemm_t256_NT"),
    (r"gemm_wp16_kernel<true, true", "gemm_t256_TT"), (r"gemm_wp16_kernel<true, false", "gemm_t256_TN"),
    (r"gemm_phased_kernel", "gemm_t256_NT"), (r"gemm_skinny", "gemm_skinny"), (r"attn_fwd_kernel", "attn_fwd"), (r"attn_bwd_dkv", "attn_bwd_dkv"),
    (r"attn_bwd_dq", "attn_bwd_dq"), (r"attn_decode", "attn_decode"),
    (r"ln_fwd", "ln_fwd"), (r"ln_bwd", "ln_bwd"), (r"adamw", "adamw"), (r"loss_row", "loss"),
    (r"colsum", "colsum"), (r"logmel", "logmel"), (r"sumsq", "sumsq"), (r"reduce_slices", "reduce_slices"),
    (r"move_rows", "move_rows"), (r"im2col|col2im", "conv_im2col"), (r"cast_f32_bf16|cast_bf16_f32", "cast"), (r"gelu_bwd", "gelu_bwd"),
    (r"embed_fwd|embed_bwd", "embed"),
]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, avg(value), sum(value), count(*) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def classify(name):
    for pat, key in CLASS_OF:
        if re.search(pat, name.replace(",", ", ").replace("  ", " ")):
            return key
    return None


def main(fetch_db, write_db, out_path, steps_profiled=2):
    """steps_profiled: training steps each PMC pass ran (tools/profile_round.sh: --steps 1 --warmup 1 = 2); bench.py compares
    launches / steps_profiled of the class it quotes with the launches its own instrumented step counted."""
    steps_profiled = int(steps_profiled)
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    classes, unclassified = {}, []
    for name in sorted(set(f) | set(w)):
        key = classify(name)
        if key is None:
            # kept visible: a kernel of this library that falls through the table is lost coverage, not noise
            unclassified.append({"symbol": name[:120], "fetch_kib_sum": f.get(name, (0, 0, 0))[1], "launches": f.get(name, (0, 0, 0))[2]})
            continue
        c = classes.setdefault(key, {"fetch_kib_sum": 0.0, "write_kib_sum": 0.0, "launches_f": 0, "launches_w": 0, "symbols": []})
        c["symbols"].append(name[:120])
        if name in f:
            c["fetch_kib_sum"] += f[name][1]; c["launches_f"] += f[name][2]
        if name in w:
            c["write_kib_sum"] += w[name][1]; c["launches_w"] += w[name][2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from distil_whisper_amd.build import kernels_sha16
    out = {"kernels_sha16": kernels_sha16(), "steps_profiled": steps_profiled, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-roofline; FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes",
           "classes": {}}
    for key, c in classes.items():
        if not c["launches_f"] or not c["launches_w"]:
            continue
        fetch = 2.0 * 1024.0 * c["fetch_kib_sum"] / c["launches_f"]
        write = 1024.0 * c["write_kib_sum"] / c["launches_w"]
        out["classes"][key] = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                               "traffic_bytes_per_launch": fetch + write, "launches": c["launches_f"],
                               "launches_per_step": c["launches_f"] / steps_profiled, "symbols": c["symbols"]}
    out["unclassified"] = sorted(unclassified, key=lambda u: -u["fetch_kib_sum"])[:40]
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    for key, c in sorted(out["classes"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{key:16s} launches {c['launches']:5d}  fetch {c['fetch_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"write {c['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main(*sys.argv[1:5])
