"""Where a decode token step spends its wall time: kernel durations against the gaps between dependent kernels.
Input: the rocpd sqlite database of `rocprofv3 --kernel-trace -- python tools/decode_profile.py`.  For the kernels of the
token steps (everything between two consecutive greedy_select launches, steady state) it prints, per step: number of
launches, the sum of kernel durations, the sum of the gaps (start of a kernel minus end of its predecessor), and a
histogram of the gaps; then the average duration per kernel symbol inside a step.  Output is markdown for profiles/."""
import sqlite3
import sys
from collections import defaultdict


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    s_col = "start" if "start" in cols else next(c for c in cols if "start" in c)
    e_col = "end" if "end" in cols else next(c for c in cols if "end" in c)
    rows = db.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    sel = [i for i, r in enumerate(rows) if "greedy_select" in r[0]]
    steps = []
    for a, b in zip(sel[:-1], sel[1:]):
        ks = rows[a + 1:b + 1]                      # one token step: the kernels up to and including its selection
        if 10 <= len(ks) <= 40:
            steps.append(ks)
    steps = steps[len(steps) // 4:]                 # steady state: drop the first quarter (warm-up, graph capture)
    gaps, per_step = [], []
    by_name = defaultdict(list)
    for ks in steps:
        dur = sum(e - s for _, s, e in ks) / 1e3
        g = [(ks[i + 1][1] - ks[i][2]) / 1e3 for i in range(len(ks) - 1)]
        gaps += g
        per_step.append((len(ks), dur, sum(g), (ks[-1][2] - ks[0][1]) / 1e3))
        for n, s, e in ks:
            by_name[n.split("(")[0][:70]].append((e - s) / 1e3)
    n = len(per_step)
    with open(out_path, "w") as f:
        f.write("# Decode token step: kernel time against launch boundaries (rocprofv3 --kernel-trace of tools/decode_profile.py)\n\n")
        f.write(f"{n} steady-state token steps.  Per step (averages): launches {sum(p[0] for p in per_step) / n:.1f}, "
                f"sum of kernel durations {sum(p[1] for p in per_step) / n:.1f} us, sum of gaps between consecutive kernels "
                f"{sum(p[2] for p in per_step) / n:.1f} us, first start to last end {sum(p[3] for p in per_step) / n:.1f} us.\n\n")
        f.write("Gap between the end of a kernel and the start of the next (us):\n\n| gap | count | share |\n|---|---|---|\n")
        edges = [0.5, 1, 1.5, 2, 3, 4, 6, 8, 12, 1e9]
        lo = -1e9
        for hi in edges:
            c = sum(1 for g in gaps if lo <= g < hi)
            f.write(f"| {'< ' + str(edges[0]) if lo < 0 else (str(lo) + ' - ' + (str(hi) if hi < 1e8 else 'inf'))} | {c} | {100.0 * c / max(1, len(gaps)):.1f} % |\n")
            lo = hi
        gs = sorted(gaps)
        f.write(f"\nmedian {gs[len(gs) // 2]:.2f} us, mean {sum(gs) / len(gs):.2f} us, p90 {gs[int(0.9 * len(gs))]:.2f} us\n\n")
        f.write("Kernels inside a step:\n\n| kernel | launches per step | avg us | us per step |\n|---|---|---|---|\n")
        for name, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{name}` | {len(v) / n:.1f} | {sum(v) / len(v):.2f} | {sum(v) / n:.1f} |\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
