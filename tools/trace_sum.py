import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    import re
    m = re.search(r'(\w+_body)', n[n.find('ZNS_'):] if 'ZNS_' in n else n)
    k = (m.group(1) if m else n[:30], r.get('Grid_Size', r.get('Grid_Size_X')))
    d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    if len(v) > 20:
        ev, od = v[0::2], v[1::2]
        print('%-28s grid %9s n %4d mean %8.1f us | even %8.1f odd %8.1f' % (k[0][-28:], k[1], len(v), sum(v)/len(v), sum(ev)/len(ev), sum(od)/len(od)))
