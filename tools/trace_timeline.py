"""Rough CU-occupancy timeline of the last timed block in a rocprofv3 --kernel-trace of bench.py: per 0.5 ms bin the sum over
kernels of (time in the bin x min(workgroups, 256)) / (256 x bin) -- above 1 when kernels queue for CUs, well below 1 where
the chip idles (block edges).   python tools/trace_timeline.py NAME_results.db [bin_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bin_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.5e6
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = db.execute(f"select start, end, grid_size_x * grid_size_y / workgroup_size_x / workgroup_size_y from {kd} order by start").fetchall()
# blocks = runs of kernels separated by gaps > 0.3 ms; take the last long one
blocks, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-50:]) > 0.3e6:
        blocks.append(cur)
        cur = []
    cur.append(r)
blocks.append(cur)
blocks = [b for b in blocks if len(b) > 500]
for k, b in enumerate(blocks):
    print(f"block {k}: {len(b)} kernels, {(max(r[1] for r in b) - b[0][0]) / 1e6:.2f} ms")
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(blocks) // 2
blk = blocks[which]
t0, t1 = blk[0][0], max(r[1] for r in blk)
n = int((t1 - t0) / bin_ns) + 1
occ = [0.0] * n
for a, b, wgs in blk:
    w = min(float(wgs), 256.0)
    i = int((a - t0) / bin_ns)
    while a < b:
        e = min(b, t0 + (i + 1) * bin_ns)
        occ[i] += (e - a) * w
        a, i = e, i + 1
print(f"block of {len(blk)} kernels, {(t1 - t0) / 1e6:.2f} ms; occupancy per {bin_ns / 1e6:.2f} ms bin:")
print(" ".join(f"{o / (256.0 * bin_ns):.2f}" for o in occ))
