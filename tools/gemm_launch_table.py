"""GEMM launches of one profiled bench run grouped by kernel and grid size (= problem shape), from a rocprofv3
rocpd database: python tools/gemm_launch_table.py gpurun_out/prof/NAME_results.db [images]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
images = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, d.grid_size_x, d.workgroup_size_x, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6 "
     f"from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 6 desc")
print(f"{'calls/img':>9} {'avg_us':>8} {'ms/img':>8} {'blocks':>7}  kernel")
for name, grid, wg, n, avg, tot in db.execute(q).fetchall()[:40]:
    name = name.replace("_ZN5dlimg12_GLOBAL__N_1", "").replace("EEEvNS_1k8GemmArgsE.kd", "").replace(".kd", "")
    print(f"{n / images:9.1f} {avg:8.1f} {tot / images:8.3f} {grid // max(wg, 1):7d}  {name[:90]}")
