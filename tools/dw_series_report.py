import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
d = [(r[2]-r[1])/1e3 for r in rows if 'dw3_stream' in r[0]]
print(len(d), "launches; us:", " ".join("%.0f" % v for v in d))
g = [(rows[i+1][1]-rows[i][2])/1e3 for i in range(len(rows)-1) if 'dw3_stream' in rows[i][0] and 'dw3_stream' in rows[i+1][0]]
print("gaps us:", " ".join("%.0f" % v for v in g[:40]))
