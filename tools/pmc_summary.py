"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files -> JSON (profiles/*_pmc.json).
usage: pmc_summary.py out.json dir1 dir2 ...   (each dir holds *_counter_collection.csv of one pass)
FETCH_SIZE / WRITE_SIZE are KB per dispatch summed over the 8 XCDs; on gfx950 FETCH_SIZE tallies a wide coalesced
read at half its bytes (MI355X_MICROARCH.md), so hbm_read_bytes_corrected = 2 * FETCH_SIZE * 1024."""
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[k] += float(r["Counter_Value"])   # one row per XCD/instance: sum
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, cname), v in per_dispatch.items():
            kn = names[disp]
            if "fvta::" not in kn:
                continue
            acc[kn.split("(")[0]][cname].append(v)
out = {}
for kn, cs in acc.items():
    out[kn] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[kn]["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in out[kn]:
        out[kn]["hbm_read_bytes_corrected"] = 2 * out[kn]["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in out[kn]:
        out[kn]["hbm_write_bytes"] = out[kn]["WRITE_SIZE"] * 1024
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for kn in sorted(out):
    print(kn[:70].ljust(70), {k: round(v / 1e6, 2) for k, v in out[kn].items() if k.startswith("hbm")}, out[kn]["dispatches"])
