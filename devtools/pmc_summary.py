"""dev helper: average PMC counters per kernel from rocprofv3 --pmc csv output."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "k_knn_tile"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if pat in k:
        # (launches enqueued behind the end of an alignment exit within microseconds: not part of the averages)
        if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < 20e3:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        v2 = v[len(v)//2:]  # later (converged) launches
        print("   %-24s n=%d mean=%.4g  late-mean=%.4g" % (c, len(v), sum(v)/len(v), sum(v2)/len(v2)))
