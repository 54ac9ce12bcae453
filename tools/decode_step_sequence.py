"""One steady-state decode token step, kernel by kernel in launch order (name, grid, workgroup, VGPRs, LDS, duration):
input the rocpd database of `rocprofv3 --kernel-trace -- python tools/decode_profile.py`."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
want = [c for c in ("name", "start", "end", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "arch_vgpr_count", "accum_vgpr_count", "lds_block_size", "lds_size", "group_segment_size", "scratch_size") if c in cols]
rows = db.execute(f"select {', '.join(want)} from kernels order by start").fetchall()
ni, si, ei = want.index("name"), want.index("start"), want.index("end")
sel = [i for i, r in enumerate(rows) if "greedy_select" in r[ni]]
steps = [(a, b) for a, b in zip(sel[:-1], sel[1:]) if 10 <= b - a <= 40]
a, b = steps[len(steps) * 3 // 4]
print("columns:", want)
# average over the last quarter of the steps, position by position
last = steps[len(steps) * 3 // 4:]
L = b - a
acc = [0.0] * L; cnt = 0
for (x, y) in last:
    if y - x != L: continue
    cnt += 1
    for k in range(L): acc[k] += (rows[x + 1 + k][ei] - rows[x + 1 + k][si]) / 1e3
for k in range(L):
    r = rows[a + 1 + k]
    extra = " ".join(f"{c}={r[want.index(c)]}" for c in want if c not in ("name", "start", "end"))
    print(f"{k:2d} {acc[k] / max(cnt, 1):7.2f} us  {r[ni][:60]:60s} {extra}")
print(f"sum {sum(acc) / max(cnt, 1):.1f} us over {cnt} steps")
