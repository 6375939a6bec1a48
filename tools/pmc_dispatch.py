"""per-dispatch PMC sums for kernels matching a substring: python tools/pmc_dispatch.py DB substr [every_nth]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
idx = {c: i for i, c in enumerate(cols)}
per = defaultdict(lambda: defaultdict(float))
for r in cur.execute("select * from pmc_events"):
    name = r[idx['name']] if 'name' in idx else r[idx['kernel_name']]
    if sys.argv[2] not in name: continue
    per[r[idx['dispatch_id']]][r[idx['counter_name']]] += r[idx['value'] if 'value' in idx else idx['counter_value']]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for i, k in enumerate(sorted(per)):
    if i % nth == 0: print(k, {c: int(v) for c, v in sorted(per[k].items())})
