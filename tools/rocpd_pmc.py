"""Per-kernel PMC counter totals from a rocprofv3 rocpd database."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
want = sys.argv[2] if len(sys.argv) > 2 else None
try:
    cols = [d[0] for d in cur.execute('select * from counters_collection limit 1').description]
    print('columns:', cols)
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                       "group by kernel_name, counter_name order by 4 desc").fetchall()
    for r in rows[:12]:
        last = [v[0] for v in cur.execute("select value from counters_collection where kernel_name = ? and counter_name = ? "
                                          "order by start", (r[0], r[1])).fetchall()][-20:]
        print('%-70s %-12s dispatches %5d  sum %.6g  avg/dispatch %.6g  avg of the last %d dispatches %.6g' % (
            r[0][:70], r[1], r[2], r[3], r[4], len(last), sum(last) / max(len(last), 1)))
except Exception as e:
    print('counters_collection query failed:', e)
    for t in ('pmc_events', 'rocpd_pmc_event', 'pmc_info'):
        try:
            print(t, [d[0] for d in cur.execute('select * from %s limit 1' % t).description])
            for r in cur.execute('select * from %s limit 3' % t): print('  ', r)
        except Exception as e2:
            print(t, 'failed', e2)
