#!/bin/bash
# round-2 set C: full GPU suite, default bench line, f32 train line, one-rank RCCL line (early gradient bucket path)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py > gpurun_out/r02c/bench_metric.json 2> gpurun_out/r02c/bench_metric.log; tail -2 gpurun_out/r02c/bench_metric.log
python bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02c/bench_f32_train.json 2>/dev/null
FVTA_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c/bench_rccl_one_rank.json 2> gpurun_out/r02c/rccl.log; tail -3 gpurun_out/r02c/rccl.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02c/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("ms_per_step_event_median"), d["kernel_ms_per_step"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
