"""Per-lane view of one timed block in a rocprofv3 --kernel-trace of bench.py: for every stream, when each pass (marked by
its preprocess kernel) starts and when the lane's last kernel ends, relative to the block's first kernel.
   python tools/trace_lanes.py NAME_results.db [block index]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
lane_col = "stream_id" if "stream_id" in cols else "queue_id"
rows = db.execute(f"select d.start, d.end, d.{lane_col}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
blocks, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-50:]) > 0.3e6:
        blocks.append(cur)
        cur = []
    cur.append(r)
blocks.append(cur)
blocks = [b for b in blocks if len(b) > 200]        # a burst of 20 images is >= 5 passes of ~95 launches
if not blocks:
    print("no block of more than 200 kernels in this trace")
    raise SystemExit(0)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(blocks) // 2
blk = blocks[which]
t0, t1 = blk[0][0], max(r[1] for r in blk)
print(f"block {which} of {len(blocks)}: {len(blk)} kernels, {(t1 - t0) / 1e6:.3f} ms, lanes by {lane_col}")
lanes = {}
for a, b, lane, name in blk:
    lanes.setdefault(lane, []).append((a, b, name))
for lane, ks_ in sorted(lanes.items()):
    starts = [(a - t0) / 1e6 for a, b, n in ks_ if "preprocess" in n]
    first = (ks_[0][0] - t0) / 1e6
    end = (max(b for a, b, n in ks_) - t0) / 1e6
    busy = sum(b - a for a, b, n in ks_) / 1e6
    print(f"lane {lane}: {len(ks_)} kernels, first kernel {first:.2f}, passes start at " + " ".join(f"{s:.2f}" for s in starts) + f" | last kernel ends {end:.3f} ms | kernel time {busy:.2f} ms")
