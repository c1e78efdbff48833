"""Summarise which (queue, stream) each kernel of a rocprofv3 --kernel-trace results.db ran on, the busy time and
span of the last training step (between the last two adadelta launches), and overlap between queues."""
import collections
import glob
import sqlite3
import sys


def main(path):
    db = glob.glob(path + "/**/*_results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    names = {r[0]: r[1] for r in c.execute("select id, kernel_name from %s" % sym)}
    rows = list(c.execute("select start,end,kernel_id,queue_id,stream_id from %s order by start" % kt))
    idx = [i for i, r in enumerate(rows) if "adadelta" in names[r[2]]]
    step = rows[idx[-2] + 1: idx[-1] + 1]
    span = (max(r[1] for r in step) - step[0][0]) / 1e6
    busy = sum(r[1] - r[0] for r in step) / 1e6
    print("%s: last step %d launches, span %.2f ms, summed kernel time %.2f ms" % (db, len(step), span, busy))
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        k = (r[3], r[4])
        per[k][0] += 1
        per[k][1] += (r[1] - r[0]) / 1e6
    for k, v in sorted(per.items()):
        print("   queue %d stream %d: %d launches, %.2f ms" % (k[0], k[1], v[0], v[1]))


if __name__ == "__main__":
    main(sys.argv[1])
