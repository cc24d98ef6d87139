"""Account for the Cholesky phase of one evaluation in a rocprofv3 kernel trace: time on the main stream spent in
leaves, in chain GEMMs, idle between them; busy time of the bulk streams.  usage: chain_account.py <trace dir>"""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
with open(files[0]) as f:
    for d in csv.DictReader(f):
        rows.append((d['Kernel_Name'].split('(')[0].replace('mfgp::', '').replace('void ', ''), int(d['Start_Timestamp']),
                     int(d['End_Timestamp']), int(d['Grid_Size_X']) // int(d['Workgroup_Size_X']), d['Queue_Id']))
rows.sort(key=lambda r: r[1])
idx = [i for i, x in enumerate(rows) if ('kbuild_' in x[0] and '<0>' in x[0])]
ev = rows[idx[-2]:idx[-1]]
leaves = [i for i, x in enumerate(ev) if 'leaf' in x[0]]
ev = ev[leaves[0]:leaves[-1] + 1]       # potrf phase: first leaf .. last leaf
mainq = ev[0][4]
t0, t1 = ev[0][1], ev[-1][2]
leaf = sum(x[2] - x[1] for x in ev if 'leaf' in x[0])
leaf_n = sum(1 for x in ev if 'leaf' in x[0])
slow_leaf = sum(max(0, (x[2] - x[1]) - 47000) for x in ev if 'leaf' in x[0])
chain = [x for x in ev if x[4] == mainq and 'leaf' not in x[0]]
chain_t = sum(x[2] - x[1] for x in chain)
main = [x for x in ev if x[4] == mainq]
idle = sum(max(0, b[1] - a[2]) for a, b in zip(main[:-1], main[1:]))
print("potrf phase %.0f us: %d leaves %.0f us (of which waiting for a CU ~%.0f), chain GEMMs %d launches %.0f us, main-stream idle %.0f us" % (
    (t1 - t0) / 1e3, leaf_n, leaf / 1e3, slow_leaf / 1e3, len(chain), chain_t / 1e3, idle / 1e3))
for q in sorted(set(x[4] for x in ev)):
    if q == mainq:
        continue
    ks = [x for x in ev if x[4] == q]
    print("  queue %s: %d launches, busy %.0f us, %d tiles" % (q, len(ks), sum(x[2] - x[1] for x in ks) / 1e3, sum(x[3] for x in ks)))
big = sorted(((b[1] - a[2]) / 1e3, (a[2] - t0) / 1e3, a[0], b[0]) for a, b in zip(main[:-1], main[1:]))[-8:]
print("  largest main-stream gaps (us, at, after, before):", [(round(g), round(at), x[:12], y[:12]) for g, at, x, y in big])
