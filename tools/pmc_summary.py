#!/usr/bin/env python3
"""Per-kernel mean of PMC counters from a rocprofv3 --pmc run (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name" \
    if "kernel_name" in cols else None
if q is None:
    print(cols); sys.exit(1)
agg = {}
for k, c, v, n in db.execute(q):
    agg.setdefault(k.split("(")[0][:40], {})[c] = (v, n)
names = sorted({c for d in agg.values() for c in d})
print("%-40s %6s " % ("kernel", "n") + " ".join("%16s" % c[:16] for c in names))
for k, d in sorted(agg.items()):
    n = max(v[1] for v in d.values())
    print("%-40s %6d " % (k, n) + " ".join("%16.0f" % d.get(c, (0, 0))[0] for c in names))
