"""A stretch of the kernel timeline of a rocprofv3 --kernel-trace database: start offset, duration and gap to the previous kernel (us)."""
import sqlite3, sys
db, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith('kernels')][0] if any(t.startswith('kernels') for t in tabs) else None
rows = cur.execute("select name, start, end from %s order by start limit %d offset %d" % (kd, count, first)).fetchall()
prev_end = None
for name, st, en in rows:
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print('%-50s dur %8.1f us   gap %6.1f us' % (name[:50], (en - st) / 1e3, gap))
    prev_end = en
