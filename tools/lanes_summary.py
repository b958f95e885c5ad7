"""Summary of an OVERLAPPED kernel trace of bench.py (all execution lanes running, the regime `value` is measured in), from a
rocprofv3 --kernel-trace database:  python tools/lanes_summary.py NAME_results.db [images per block] [flop per image]

For the timed blocks in the middle of the run (blocks = runs of kernels separated by idle gaps) it prints
  * per kernel group (kernel name + workgroup count): launches per block, average duration UNDER CONTENTION, CU.us per
    image = workgroup-slots x duration (slots = min(workgroups, 256 x workgroups-per-CU the kernel allows, taken as 1 for
    the kernels with > 64 KB of LDS and 2 otherwise)), share of the sum
  * the sum over groups per image against the 256 CUs x wall time per image: the chip's occupancy by this accounting
  * wall time per image and chip_frac = FLOP per image / wall time per image / 2.5 PFLOP/s -- what bench.py's
    roofline.chip_frac reports, recomputed from the trace alone
  * a CU-occupancy timeline of one block (0.25 ms bins)
It is the tracked counterpart of `roofline.under_lanes` in the bench line (profiles/r04_lanes_summary.txt)."""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
args = [x for x in sys.argv[2:] if not x.startswith('--')]
images_per_block = int(args[0]) if len(args) > 0 else 8
flop_per_image = float(args[1]) if len(args) > 1 else (941.7e9 + 3.62e9)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
gx, gy, wx, wy = (("grid_size_x", "grid_size_y", "workgroup_size_x", "workgroup_size_y") if "grid_size_x" in cols
                  else ("grid_x", "grid_y", "workgroup_x", "workgroup_y"))
lds_col = next((c for c in ("lds_block_size", "group_segment_size", "lds_size") if c in cols), None)
sel_lds = f", d.{lds_col}" if lds_col else ", 0"
lane_col = "stream_id" if "stream_id" in cols else "queue_id"
rows = db.execute(f"select d.start, d.end, (d.{gx} / d.{wx}) * (d.{gy} / d.{wy}), s.kernel_name{sel_lds}, d.{lane_col} from {kd} d "
                  f"join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
blocks, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur[-50:]) > 0.3e6:
        blocks.append(cur)
        cur = []
    cur.append(r)
blocks.append(cur)


def describe(b):
    # images of a block: the pre-processing kernel runs 512 workgroups per image, whatever the step queue made of the requests
    pre = sum(int(r[2]) // 512 for r in b if "preprocess" in r[3])
    lanes = len({r[5] for r in b})
    return len(b), (max(r[1] for r in b) - b[0][0]) / 1e3, pre, lanes


# the timed repeats: blocks that hold exactly `images_per_block` images (warm-up and the per-stage rates have other sizes)
cand = [b for b in blocks if describe(b)[2] == images_per_block]
if "--blocks" in sys.argv:
    for i, b in enumerate(blocks):
        n, w, pre, lanes = describe(b)
        print(f"block {i:3d}: {n:5d} kernels {w:9.1f} us, {pre} images, {lanes} lanes")
med = statistics.median([describe(b)[1] for b in cand]) if cand else 0.0
timed = [b for b in cand if describe(b)[1] <= 1.3 * med]
print(f"{len(blocks)} blocks of kernels in the trace; {len(cand)} hold {images_per_block} images (the timed repeats; lanes used: "
      f"{sorted({describe(b)[3] for b in cand})}), {len(timed)} of them within 1.3 x the median wall time are used")
if not timed:
    print(f"no block of this trace holds {images_per_block} images: the per-kernel table below covers ALL kernels of the trace, and the "
          "wall-time lines describe arbitrary blocks")
    timed = [b for b in blocks if describe(b)[0] > 0]
    if not timed:
        print("the trace holds no kernels at all")
        raise SystemExit(0)

def short(name):
    name = name.replace("_ZN5dlimg12_GLOBAL__N_1", "").replace(".kd", "")
    for a, b in (("EEEvNS_1k8GemmArgsEi", ""), ("EEEvNS_1k8GemmArgsE", ""), ("14gemm_pp_kernelILi", "gemm_pp<"), ("17gemm_pp128_kernelILi", "gemm_pp128<"),
                 ("ELi", ",")):
        name = name.replace(a, b)
    return name[:64]


wall = [(max(r[1] for r in b) - b[0][0]) / 1e3 for b in timed]          # us
groups = {}
for b in timed:
    for a, e, wgs, name, lds, _lane in b:
        per_cu = 1 if (lds and lds > 64 * 1024) else 2
        slots = min(float(wgs), 256.0 * per_cu) / per_cu
        g = groups.setdefault((short(name), int(wgs)), [0, 0.0, 0.0])
        g[0] += 1
        g[1] += (e - a) / 1e3
        g[2] += slots * (e - a) / 1e3
nb, imgs = len(timed), len(timed) * images_per_block
total_cu_us = sum(g[2] for g in groups.values()) / imgs
wall_per_image = statistics.mean(wall) / images_per_block
print(f"wall time per block {statistics.mean(wall):.1f} us (min {min(wall):.1f}, max {max(wall):.1f}) -> {wall_per_image:.1f} us per image "
      f"= {1e6 / wall_per_image:.0f} images/s under the profiler")
print(f"chip_frac recomputed from the trace: {flop_per_image / 1e12:.4f} TFLOP per image / {wall_per_image:.1f} us / 2500 TFLOP/s = "
      f"{flop_per_image / (wall_per_image * 1e-6) / 2.5e15:.3f}")
print(f"sum over kernels of CU-slots x duration: {total_cu_us / 1e3:.1f} k CU.us per image = {total_cu_us / 256:.1f} us on 256 CUs "
      f"-> occupancy {total_cu_us / 256 / wall_per_image:.2f} of the wall time")
print(f"\n{'launches/blk':>12} {'avg_us':>8} {'kCU.us/img':>10} {'share':>6}  kernel (workgroups)")
for (name, wgs), g in sorted(groups.items(), key=lambda kv: -kv[1][2])[:28]:
    print(f"{g[0] / nb:12.1f} {g[1] / g[0]:8.1f} {g[2] / imgs / 1e3:10.2f} {g[2] / imgs / total_cu_us:6.3f}  {name} ({wgs})")

blk = timed[len(timed) // 2]
t0, t1 = blk[0][0], max(r[1] for r in blk)
bin_ns = 0.25e6
occ = [0.0] * (int((t1 - t0) / bin_ns) + 1)
for a, e, wgs, name, lds, _lane in blk:
    per_cu = 1 if (lds and lds > 64 * 1024) else 2
    w = min(float(wgs), 256.0 * per_cu) / per_cu
    i = int((a - t0) / bin_ns)
    while a < e:
        nxt = min(e, t0 + (i + 1) * bin_ns)
        occ[i] += (nxt - a) * w
        a, i = nxt, i + 1
print(f"\nCU occupancy of one block per 0.25 ms (sum of CU-slots in use / 256; above 1 = kernels queue for CUs):")
print(" ".join(f"{o / (256.0 * bin_ns):.2f}" for o in occ))
