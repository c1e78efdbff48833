#!/bin/bash
# kernel-level breakdown of the train step entered with token ids (bench.py --front-end)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_frontend
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --front-end --steps 5 --warmup 2 --no-cpu-baseline > $out/ks_bench.json 2> $out/ks.err
ls $out/ks
