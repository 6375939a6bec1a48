"""Summarise a rocprofv3 rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) into the same
per-kernel table `--stats` prints: calls, total / average / min / max duration.   python tools/rocpd_stats.py DB [> out.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       f"group by {name} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, c, s, a, mn, mx in rows:
        print(f"{n[:80]:80s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    print(f"{'TOTAL':80s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
