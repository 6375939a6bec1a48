"""Summarise PMC counters from a rocprofv3 rocpd sqlite database per kernel: python tools/pmc_summary.py DB [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt="", last=0):
    """last > 0: only the last `last` launches (dispatch ids) of every kernel — bench.py's target rendering uses the same kernels in other modes"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    rows = cur.execute("select * from pmc_events").fetchall()
    idx = {c: i for i, c in enumerate(cols)}
    namecol = [c for c in cols if c in ("name", "kernel_name")]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    keep = None
    if int(last) > 0 and "dispatch_id" in idx and namecol:
        ids = defaultdict(set)
        for r in rows:
            ids[r[idx[namecol[0]]]].add(r[idx["dispatch_id"]])
        keep = {k: set(sorted(v)[-int(last):]) for k, v in ids.items()}
    for r in rows:
        if keep is not None and r[idx["dispatch_id"]] not in keep[r[idx[namecol[0]]]]:
            continue
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
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else 0)
