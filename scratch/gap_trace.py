"""inter-kernel gaps of one EM iteration from a rocprofv3 kernel trace (run under
rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --only-headline)
usage: gap_trace.py DIR"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:36]))
rows.sort()
gaps = collections.defaultdict(list)
durs = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if s1 - e0 < 200000:                       # same burst of launches
        gaps[(n0, n1)].append(s1 - e0)
    durs[n0].append(e0 - s0)
print("gap after -> before: n, mean us")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    print("  %-38s -> %-38s n=%4d  %.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
