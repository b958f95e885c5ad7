"""Do kernels of different streams overlap in a rocprofv3 --kernel-trace of the 3-lane bench?  Prints the fraction of
the traced window in which 0, 1, 2, 3+ kernels are in flight and the kernels that most often run alone.
    python tools/trace_overlap.py gpurun_out/prof/NAME_results.db"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select d.start, d.end, d.stream_id, s.kernel_name from {kd} d join {sym} s on d.kernel_id=s.id "
                  f"order by d.start").fetchall()
rows = rows[len(rows) // 3:]          # skip warm-up / model load
events = []
for i, (a, b, st, name) in enumerate(rows):
    events.append((a, 1, i))
    events.append((b, -1, i))
events.sort()
level, last, hist = 0, events[0][0], Counter()
alone = Counter()
active = set()
for t, d, i in events:
    hist[min(level, 3)] += t - last
    if level == 1:
        alone[rows[next(iter(active))][3][:60]] += t - last
    last = t
    level += d
    (active.add if d > 0 else active.discard)(i)
total = sum(hist.values())
print("streams:", len({r[2] for r in rows}), " kernels:", len(rows), " window %.1f ms" % (total / 1e6))
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {100.0 * hist[k] / total:5.1f} %")
print("most time alone on the GPU:")
for name, ns in alone.most_common(6):
    print(f"  {100.0 * ns / total:5.1f} %  {name}")
