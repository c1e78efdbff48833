import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last N seconds: find the last ~2500 kernels (the ragged case)
tail = rows[-int(sys.argv[2]):]
agg = collections.defaultdict(lambda: [0, 0])
for r in tail:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"][:60] + " q" + r.get("Queue_Id", "?")
    agg[k][0] += d; agg[k][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%9.1f us total %5d calls %8.1f avg  %s" % (v[0] / 1e3, v[1], v[0] / 1e3 / v[1], k))
print("span ms", (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e6, "queues", sorted(set(r.get("Queue_Id") for r in tail)))
