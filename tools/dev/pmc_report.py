import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from counters_collection order by dispatch_id").fetchall()
d = collections.OrderedDict()
for disp, kn, gs, ws, cn, val, dur in rows:
    e = d.setdefault(disp, {'k': kn.replace('(anonymous namespace)::', '').split('(')[0][-46:], 'grid': gs, 'wg': ws, 'dur': dur})
    e[cn] = e.get(cn, 0) + val
names = sorted({r[4] for r in rows})
seen = collections.Counter()
print('%-46s %9s %8s ' % ('kernel', 'grid', 'dur_us') + ' '.join('%14s' % n.replace('_sum', '')[-14:] for n in names))
for disp, e in d.items():
    if e['k'].startswith('void at::') or 'rocclr' in e['k'] or 'elementwise' in e['k'] or 'distribution' in e['k']: continue
    seen[(e['k'], e['grid'])] += 1
    if seen[(e['k'], e['grid'])] != 3: continue      # third (warm) launch of each
    print('%-46s %9d %8.1f ' % (e['k'], e['grid'], e['dur'] / 1e3) + ' '.join('%14.4g' % e.get(n, float('nan')) for n in names))
