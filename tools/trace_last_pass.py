"""timeline of the LAST evaluation pass in a rocprofv3 kernel trace (from the last group of K-build launches to the end):
trace_last_pass.py <trace dir> [max rows]"""
import csv
import glob
import sys

path = sys.argv[1]
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 400
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
with open(files[0]) as f:
    for d in csv.DictReader(f):
        rows.append((d['Kernel_Name'].split('(')[0].replace('mfgp::', '').replace('void ', ''), int(d['Start_Timestamp']),
                     int(d['End_Timestamp']), int(d['Grid_Size_X']) // int(d['Workgroup_Size_X']) * max(1, int(d.get('Grid_Size_Y', 1) or 1)),
                     d['Queue_Id']))
rows.sort(key=lambda r: r[1])
last = max(i for i, x in enumerate(rows) if 'kbuild_' in x[0])
first = last
while first > 0 and 'kbuild_' in rows[first - 1][0]:
    first -= 1
ev = rows[first:]
t0 = ev[0][1]
busy = {}
for x in ev:
    busy[x[0]] = busy.get(x[0], 0.0) + (x[2] - x[1]) / 1e3
print("# pass: %d launches, %.1f us from first start to last end" % (len(ev), (max(x[2] for x in ev) - t0) / 1e3))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("#   %-34s %9.1f us in %d launches" % (k, v, sum(1 for x in ev if x[0] == k)))
for x in ev[:nmax]:
    print("%-30s q%s blocks %6d  start %9.1f  end %9.1f  dur %7.1f us" % (x[0], x[4], x[3], (x[1] - t0) / 1e3, (x[2] - t0) / 1e3, (x[2] - x[1]) / 1e3))
