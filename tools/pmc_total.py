import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
tot = {}
per = {}
for k, c, v, n in rows:
    if k.startswith("void at::") or "elementwise" in k or "distribution" in k: continue
    tot[c] = tot.get(c, 0) + v
    per.setdefault(k.split("(")[0][:34], {})[c] = (v, n)
nf = int(sys.argv[2])
print({c: round(v / nf) for c, v in tot.items()}, "per forward")
for k, d in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1].values()))[:14]:
    print("%-36s" % k, {c: round(v[0] / nf) for c, v in d.items()})
