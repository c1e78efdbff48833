"""Per-(kernel, grid) launch statistics from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the text cell and
the photo cell launch the same LSTM kernel symbols with different grids, which --stats averages together.
  python tools/trace_by_grid.py gpurun_out/r01e/ks2/ks_kernel_trace.csv [substring] > profiles/...json"""
import collections
import csv
import json
import sys


def main(path, needle="lstm_"):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            key = "%s grid(%s,%s,%s)" % (r["Kernel_Name"].split("(")[0], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {k: dict(launches=len(v), avg_us=round(sum(v) / len(v), 1), min_us=round(min(v), 1), max_us=round(max(v), 1))
           for k, v in sorted(d.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
