"""Concurrency summary of a rocprofv3 rocpd .db (kernel trace): how busy the device was between the first and the last kernel.
    python tools/dev/prof_timeline.py <db> [out.txt]
union = time with at least one kernel running; sum = sum of kernel durations (sum / union = mean number of kernels in flight)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if 'kernel_dispatch' in t]
out = ['tables: %d; kernel-dispatch tables: %s' % (len(tabs), disp)]
for t in disp:
    cols = [r[1] for r in db.execute('pragma table_info(%s)' % t)]
    if 'start' not in cols or 'end' not in cols:
        continue
    rows = db.execute('select start, end from %s order by start' % t).fetchall()
    if not rows:
        continue
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    s = sum(e - b for b, e in rows)
    union, cur_b, cur_e = 0, rows[0][0], rows[0][1]
    for b, e in rows[1:]:
        if b > cur_e:
            union += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_b
    out.append('%s: %d dispatches, span %.1f ms, union busy %.1f ms (%.1f %%), sum of durations %.1f ms (%.2f kernels in flight while busy)'
               % (t, len(rows), (t1 - t0) / 1e6, union / 1e6, 100.0 * union / (t1 - t0), s / 1e6, s / union))
    # the densest second: where the timed legs are
    break
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'a').write(txt + '\n')
