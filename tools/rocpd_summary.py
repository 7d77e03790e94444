"""Summarise a rocprofv3 rocpd database (kernel trace) as text: per-kernel calls / avg / min / max duration, registers, LDS."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
print('%-90s %6s %12s %12s %12s %5s %5s %8s %7s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'vgpr', 'sgpr', 'lds_B', 'scratch'))
rows = cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), "
                   "max(sgpr_count), max(lds_size), max(scratch_size), sum(end-start) from kernels group by name order by 10 desc")
for r in rows:
    print('%-90s %6d %12.3f %12.3f %12.3f %5d %5d %8d %7d' % (r[0][:90], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
