"""Summarise PMC counters from a rocprofv3 rocpd sqlite database per kernel: python tools/pmc_summary.py DB [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    rows = cur.execute("select * from pmc_events").fetchall()
    idx = {c: i for i, c in enumerate(cols)}
    namecol = [c for c in cols if c in ("name", "kernel_name")]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in rows:
        k = r[idx[namecol[0]]] if namecol else "?"
        c = r[idx["counter_name"]] if "counter_name" in idx else r[idx.get("pmc_name", 0)]
        v = r[idx["value"]] if "value" in idx else r[idx["counter_value"]]
        if filt in k:
            agg[k][c] += v
            cnt[k][c] += 1
    for k in agg:
        print(k[:110])
        for c in sorted(agg[k]):
            print(f"    {c:32s} per-dispatch {agg[k][c]/cnt[k][c]:16.1f}   dispatches {cnt[k][c]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
