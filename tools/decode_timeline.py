"""Timeline of one compute_mask call from a rocprofv3 --kernel-trace database of tools/decode_probe.py: every kernel
between two prompt_tokens launches with its start offset, duration and the idle gap in front of it.
usage: python tools/decode_timeline.py DB [call index from the end, default 3]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id=s.id "
                      f"order by d.start").fetchall()
    starts = [i for i, r in enumerate(rows) if "decoder_start" in r[0] or "prompt_tokens" in r[0]]
    a, b = starts[-back - 1], starts[-back]
    t0, prev_end, busy = rows[a][1], rows[a][1], 0.0
    for name, s, e in rows[a:b]:
        name = name.replace("_ZN5dlimg12_GLOBAL__N_1", "")[:60]
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}  {name}")
        busy += (e - s) / 1e3
        prev_end = e
    print(f"{b - a} launches, kernels busy {busy:.1f} us, first start to last end {(prev_end - t0) / 1e3:.1f} us, "
          f"next call starts at {(rows[b][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
