"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        f.write("| kernel | calls | total_us | avg_us | min_us | max_us | pct |\n|---|---|---|---|---|---|---|\n")
        for name, n, tot, avg, mn, mx in rows:
            f.write(f"| `{name[:110]}` | {n} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
                    f"{100.0 * tot / total:.2f} |\n")
    print(open(out_path).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
