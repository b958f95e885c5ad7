"""Per-kernel launch count / average / total from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME).
usage: python tools/kernel_stats.py gpurun_out/prof/NAME_results.db [top] [--by-grid]
--by-grid splits a kernel's launches by workgroup count (tells qkv / fc1 / proj / fc2 launches of one GEMM kernel apart)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    top = int(args[0]) if args else 25
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    if by_grid:
        cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
        gx, wx = ("grid_size_x", "workgroup_size_x") if "grid_size_x" in cols else ("grid_x", "workgroup_x")
        q = (f"select s.kernel_name || ' wgs=' || cast(d.{gx} / d.{wx} as text), count(*), avg(d.end-d.start)/1000.0, "
             f"sum(d.end-d.start)/1e6 from {kd} d join {sym} s on d.kernel_id=s.id group by 1 order by 4 desc")
    else:
        q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6 from {kd} d "
             f"join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
    print(f"{'calls':>7} {'avg_us':>9} {'total_ms':>9}  kernel")
    for name, n, avg, tot in db.execute(q).fetchall()[:top]:
        name = name.replace("_ZN5dlimg12_GLOBAL__N_1", "").replace("EEEvNS_1k8GemmArgsE.kd", "")
        print(f"{n:7d} {avg:9.1f} {tot:9.2f}  {name[:110]}")


if __name__ == "__main__":
    main()
