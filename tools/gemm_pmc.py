"""Per-launch PMC averages of the step's GEMM launches, ONE ROW PER SHAPE (tools/gemm_step_shapes.py with SHAPES=<name> under
`rocprofv3 --pmc ... --kernel-trace`, one pass per counter group and shape) as a markdown table:
    python tools/gemm_pmc.py "<shape label>=<results.db>[,<results.db>...]" ... > profiles/rN_gemm_pmc_table.md
Derived columns: matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); VALU : MFMA instruction
ratio; LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict cycles over cycles the LDS index unit is
busy); wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave-cycles spent waiting on any counter)."""
import sqlite3, sys
from collections import defaultdict
rows = []
for arg in sys.argv[1:]:
    label, paths = arg.split("=", 1)
    vals = defaultdict(dict)
    for path in paths.split(","):
        db = sqlite3.connect(path)
        for k, c, v, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                     "where kernel_name like '%gemm%' and kernel_name not like '%reduce%' group by kernel_name, counter_name"):
            vals[k.split("(")[0][:64]][c] = (v, n)
    # the shape's main kernel = the one with the most MFMA instructions (a row-tail launch rides along with some shapes)
    if not vals:
        continue
    k = max(vals, key=lambda kk: vals[kk].get("SQ_INSTS_MFMA", (0, 0))[0])
    rows.append((label, k, vals[k]))
names = sorted({c for _, _, d in rows for c in d})
print("| launch (M = 48 000) | kernel | " + " | ".join(names) + " | matrix pipe busy | VALU per MFMA | LDS bank-conflict share | wait share |")
print("|---|---|" + "---|" * (len(names) + 4))
for label, k, d in rows:
    get = lambda c: d.get(c, (0, 0))[0]
    g, busy, mf, va = get("GRBM_GUI_ACTIVE"), get("SQ_VALU_MFMA_BUSY_CYCLES"), get("SQ_INSTS_MFMA"), get("SQ_INSTS_VALU")
    bc, la, wa, wc = get("SQ_LDS_BANK_CONFLICT"), get("SQ_LDS_IDX_ACTIVE"), get("SQ_WAIT_ANY"), get("SQ_WAVE_CYCLES")
    print(f"| {label} | `{k}` | " + " | ".join(f"{d[c][0] / 1e6:.2f} M" if c in d else "-" for c in names) +
          f" | {(100.0 * busy / (1024 * g / 8) if g else 0):.1f} % | {(va / mf if mf else 0):.2f} | {(100.0 * bc / la if la else 0):.1f} % | "
          f"{(100.0 * wa / wc if wc else 0):.1f} % |")
