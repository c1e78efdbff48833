#!/bin/bash
# Container side of the round's measurement set: REFUSES a dirty tree (every figure in profiles/r06_* must belong to a commit),
# stamps HEAD for the GPU side, runs tools/collect_profiles_r06.sh through gpurun, copies the results into profiles/r06_* and
# records date + commit of every counter file in profiles/pmc_meta.json.   tools/collect_r06.sh [full]
set -eu
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain)" ]; then echo "collect_r06: the tree is dirty -- commit first" >&2; git status --short >&2; exit 1; fi
head=$(git rev-parse --short HEAD)
echo "$head" > tools/.collect_head        # (git-ignored; travels to the box with the snapshot)
/usr/local/graft/bin/gpurun --timeout 3000 -- "tools/collect_profiles_r06.sh r06 ${1:-}"
rm -f tools/.collect_head
[ "$(cat gpurun_out/r06/head.txt)" = "$head" ] || { echo "collect_r06: the box ran another tree" >&2; exit 1; }
for f in gpurun_out/r06/*.json gpurun_out/r06/*.csv gpurun_out/r06/*.txt; do [ -f "$f" ] && cp "$f" "profiles/r06_$(basename "$f")"; done
python3 - "$head" <<'PY'
import datetime, glob, json, os, sys
p = "profiles/pmc_meta.json"
meta = json.load(open(p)) if os.path.exists(p) else {}
for f in glob.glob("profiles/r06_*_pmc.json") + ["profiles/r06_kernel_stats.csv", "profiles/r06_bench_metric.json"]:
    if os.path.exists(f):
        meta[os.path.basename(f)] = dict(date=str(datetime.date.today()), commit=sys.argv[1])
json.dump(meta, open(p, "w"), indent=1)
PY
echo "collected at $head"
