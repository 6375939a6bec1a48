"""Print the kernel timeline of one full step in a rocprofv3 rocpd database: start offset, duration, gap to the previous
kernel's end, kernel name.  A step runs from one marker kernel (default: the schedule_next kernel that opens every step; without one, the kernel
that follows an Adam launch) to the next.   python tools/rocpd_timeline.py DB [marker-substring] [step-index]     (step-index: 0-based from the first marker; default
the last full step — bench.py ends with keep_image steps, its timed loss-only steps are the indices warmup .. warmup+steps-1)"""
import sqlite3
import sys


def main(path, marker="schedule_next", step=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = ", stream_id" if "stream_id" in cols else (", queue_id" if "queue_id" in cols else "")
    rows = cur.execute(f"select {name}, start, end{extra} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3 and marker == "schedule_next":
        # folded steps (FitEngine.fold_step) have no schedule kernel: a step then runs from the kernel after one Adam launch (the last
        # kernel of every step) through the next Adam launch
        # (with `split_adam` the maps' own adam_dev launch sits inside the step: the step ends with adam_dev2 — or, appearance-only, with the LAST adam_dev)
        last = "adam_dev2" if any("adam_dev2" in r[0] for r in rows) else "adam_dev"
        marks = [i + 1 for i, r in enumerate(rows) if last in r[0] and i + 1 < len(rows)]
    if len(marks) < 3:
        print("not enough steps"); return
    lo, hi = (marks[-2], marks[-1]) if step is None else (marks[int(step)], marks[int(step) + 1])
    seg = rows[lo:hi]
    t0 = seg[0][1]
    prev_end = t0
    print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s} {'q':>4s}  kernel")
    for r in seg:
        q = r[3] if len(r) > 3 else 0
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} {(r[1]-prev_end)/1e3:8.1f} {str(q)[-4:]:>4s}  {r[0][:90]}")
        prev_end = max(prev_end, r[2])
    print(f"step span {(max(r[2] for r in seg)-t0)/1e3:.1f} us, kernels {len(seg)}, sum of durations {sum(r[2]-r[1] for r in seg)/1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
