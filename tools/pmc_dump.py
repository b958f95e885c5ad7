"""Sums of the counters of a rocprofv3 --pmc run per kernel: python tools/pmc_dump.py DB [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for name, disp, counter, value, dur in db.execute("select kernel_name, dispatch_id, counter_name, value, (end - start) from counters_collection"):
    if sub not in name:
        continue
    key = name.split("(")[0][:60]
    a = acc.setdefault(key, {"n": set(), "ns": {}})
    a["n"].add(disp); a["ns"][disp] = dur
    a[counter] = a.get(counter, 0.0) + value
for k, a in acc.items():
    n = len(a["n"])
    print(k, "launches", n, "avg_us %.1f" % (sum(a["ns"].values()) / n / 1e3))
    for c, v in sorted(a.items()):
        if c in ("n", "ns"):
            continue
        print("   %-34s %16.0f per launch" % (c, v / n))
