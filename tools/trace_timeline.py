import csv, sys, glob
path = sys.argv[1]; n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0; n1 = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
with open(files[0]) as f:
    for d in csv.DictReader(f):
        rows.append((d['Kernel_Name'].split('(')[0].replace('mfgp::', '').replace('void ', '').replace(', ', ','), int(d['Start_Timestamp']),
                     int(d['End_Timestamp']), int(d['Grid_Size_X']) // int(d['Workgroup_Size_X']), d['Queue_Id'], d.get('Stream_Id', '')))
rows.sort(key=lambda r: r[1])
idx = [i for i, x in enumerate(rows) if ('kbuild_' in x[0] and '<0>' in x[0])]
ev = rows[idx[-2]:idx[-1]]
t0 = ev[0][1]
for x in ev[n0:n1]:
    print("%-28s q%s s%s blocks %5d  start %9.1f  end %9.1f  dur %7.1f us" % (x[0], x[4], x[5], x[3], (x[1] - t0) / 1e3, (x[2] - t0) / 1e3, (x[2] - x[1]) / 1e3))
