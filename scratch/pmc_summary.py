"""average PMC counters per kernel from rocprofv3 counter_collection CSVs"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-40s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
