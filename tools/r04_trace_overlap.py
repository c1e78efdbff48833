"""Reads a rocprofv3 kernel trace (dir, kernel-name substrings a b): average durations and how much of the a-kernels' time
overlaps b-kernels running on another queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
A = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if sys.argv[2] in r["Kernel_Name"]]
B = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if sys.argv[3] in r["Kernel_Name"]]
A, B = A[len(A) // 2:], B[len(B) // 2:]          # the later (timed) half
print("%s: %d launches, avg %.1f us, queues %s" % (sys.argv[2], len(A), sum(e - s for s, e, _ in A) / len(A) / 1e3, sorted(set(q for _, _, q in A))))
print("%s: %d launches, avg %.1f us, queues %s" % (sys.argv[3], len(B), sum(e - s for s, e, _ in B) / len(B) / 1e3, sorted(set(q for _, _, q in B))))
ov = 0
for s, e, q in A:
    for s2, e2, q2 in B:
        if q2 != q and s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
print("overlap of %s with %s on another queue: %.1f %% of its time" % (sys.argv[2], sys.argv[3], 100.0 * ov / sum(e - s for s, e, _ in A)))
t0, t1 = min(s for s, _, _ in A + B), max(e for _, e, _ in A + B)
print("span %.3f ms for %d + %d launches" % ((t1 - t0) / 1e6, len(A), len(B)))
