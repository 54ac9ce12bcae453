"""Per-launch PMC averages of the attention kernels (tools/attn_single.py under rocprofv3 --pmc ... --kernel-trace) as a
markdown table: python tools/attn_pmc.py <results.db> [<results.db> ...] > profiles/rN_attn_pmc.md"""
import sqlite3, sys
from collections import defaultdict
vals = defaultdict(dict)
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    for k, c, v, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                 "where kernel_name like '%attn%' group by kernel_name, counter_name"):
        vals[k.split("(")[0][:60]][c] = (v, n)
names = sorted({c for d in vals.values() for c in d})
print("| kernel | launches | " + " | ".join(names) + " | matrix pipe busy | VALU per MFMA |")
print("|---|---|" + "---|" * (len(names) + 2))
for k, d in vals.items():
    g = d.get("GRBM_GUI_ACTIVE", (0, 0))[0]; busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
    mf = d.get("SQ_INSTS_MFMA", (0, 0))[0]; va = d.get("SQ_INSTS_VALU", (0, 0))[0]
    n = max(x[1] for x in d.values())
    pipe = f"{100.0 * busy / (1024 * g / 8):.1f} %" if g else "-"       # busy cycles over 1024 SIMDs x (GUI_ACTIVE summed over 8 XCDs / 8)
    print(f"| `{k}` | {n} | " + " | ".join(f"{d[c][0] / 1e6:.2f} M" if c in d else "-" for c in names) +
          f" | {pipe} | {va / mf:.1f} |" if mf else " | - |")
