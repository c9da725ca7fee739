"""Turns a rocprofv3 rocpd .db (kernel trace) into a text summary: python scratch/prof_summary.py <db> [out.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
lines = ['%-70s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct')]
for n, c, t, a, p in rows:
    n = n.replace('(anonymous namespace)::', '').split('(')[0][-70:] if not n.startswith('void at::') else ('torch:' + n.split('<')[0].split('::')[-1] + ' ' + n[n.find('<'):][:40])[:70]
    lines.append('%-70s %8d %12.1f %10.2f %6.2f%%' % (n, c, t / 1e3 if t > 1e7 else t, a / 1e3 if t > 1e7 else a, p))
lines.append('TOTAL kernel time: %.1f (same unit as total_us)' % (tot / 1e3 if tot > 1e7 else tot))
txt = '\n'.join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
