"""summarise a rocprofv3 kernel_trace.csv: per-kernel totals for the LAST evaluation in the trace"""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
with open(files[0]) as f:
    for d in csv.DictReader(f):
        rows.append((d['Kernel_Name'].split('(')[0].replace('mfgp::', '').replace('void ', ''), int(d['Start_Timestamp']),
                     int(d['End_Timestamp']), int(d['Grid_Size_X']) // int(d['Workgroup_Size_X'])))
rows.sort(key=lambda r: r[1])
idx = [i for i, x in enumerate(rows) if ('kbuild_' in x[0] and '<0>' in x[0])]
s, e = idx[-2], idx[-1]
ev = rows[s:e]
print(len(ev), 'dispatches; span ms %.3f' % ((ev[-1][2] - ev[0][1]) / 1e6))
tot, cnt = {}, {}
gap = 0
for i, x in enumerate(ev):
    tot[x[0]] = tot.get(x[0], 0) + (x[2] - x[1]); cnt[x[0]] = cnt.get(x[0], 0) + 1
    if i > 0:
        gap += max(0, x[1] - ev[i - 1][2])
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-32s n=%4d total %.3f ms avg %.1f us" % (k, cnt[k], tot[k] / 1e6, tot[k] / cnt[k] / 1e3))
print('gaps ms %.3f' % (gap / 1e6))
if len(sys.argv) > 2:
    for x in ev:
        print(x[0], x[3], round((x[2] - x[1]) / 1e3, 1))
