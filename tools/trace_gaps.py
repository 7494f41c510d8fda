"""GPU idle time between kernels of a `rocprofv3 --kernel-trace` run.

usage: python tools/trace_gaps.py <rocprof output dir> [min kernels per burst]
Prints, for the steady part of the trace (after the first 20 % of the kernels), the wall span, the sum of kernel
durations, the idle share, and the histogram of the gaps in front of each kernel name.
"""
import csv, sys, glob, re, collections

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 5:]


def short(n):
    m = re.search(r'(\w+_body)', n[n.find('ZNS_'):] if 'ZNS_' in n else n)
    return m.group(1) if m else n[:30]


busy = 0
gaps = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if prev_end is not None:
        gaps[short(r['Kernel_Name'])].append((s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels %d span %.2f ms busy %.2f ms idle %.1f %%' % (len(rows), span / 1e6, busy / 1e6, 100 * (1 - busy / span)))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-28s n %5d gap before: mean %7.1f us median %7.1f p90 %7.1f total %8.2f ms' % (k[-28:], len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)], sum(v) / 1e3))
