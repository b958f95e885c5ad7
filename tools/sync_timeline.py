"""One synchronous caller of slots 3 / 4 under rocprofv3 --kernel-trace: per call (from one preprocess_kernel to the next) the
GPU's busy time, the gaps between its kernels, and how long the GPU sits idle between calls (= host time: upload, wake-up,
copy-out, launch latency).
    rocprofv3 --kernel-trace -d /tmp/st -o st -- python3 tools/process_probe.py ; python3 tools/sync_timeline.py /tmp/st/.../st_results.db"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id=s.id order by d.start").fetchall()
starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0]]
calls = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    has_decode = any("decoder_start" in r[0] for r in seg)
    busy = sum(e - s for _, s, e in seg) / 1e3
    span = (seg[-1][2] - seg[0][1]) / 1e3
    gaps = sum(max(0, seg[i + 1][1] - seg[i][2]) for i in range(len(seg) - 1)) / 1e3
    enc_end = max((e for n, s, e in seg if "layernorm" in n), default=seg[-1][2])
    idle_after = (rows[b][1] - seg[-1][2]) / 1e3
    big_gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][-40:], seg[i + 1][0][-40:]) for i in range(len(seg) - 1))[-3:]
    calls.append((has_decode, len(seg), busy, span, gaps, idle_after, (enc_end - seg[0][1]) / 1e3, big_gaps))
for kind in (False, True):
    sel = [c for c in calls if c[0] == kind][5:]
    if not sel:
        continue
    med = lambda i: float(np.median([c[i] for c in sel]))
    print(f"{'process + compute_mask' if kind else 'process only':24s}: {len(sel)} calls, {int(med(1))} kernels, GPU busy {med(2):7.1f} us, first start to last end "
          f"{med(3):7.1f} us (gaps inside {med(4):6.1f} us), encoder part {med(6):7.1f} us, GPU idle until the next call's first kernel {med(5):6.1f} us "
          f"=> {med(3) + med(5):7.1f} us per call")
    print("   largest gaps of the last such call:", [(round(g, 1), a, b) for g, a, b in sel[-1][7]])
