#!/bin/bash
# tools/kstats.sh <tag> <python script and args...>: rocprofv3 --kernel-trace --stats of one command, top kernels printed
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run through gpurun}" || exit 1
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/ks -o ks -- python3 "$@" > gpurun_out/$tag/run.log 2> gpurun_out/$tag/run.err
f=$(find gpurun_out/$tag/ks -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$tag/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(60)]:
    print("%-90s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 gpurun_out/$tag/run.log
