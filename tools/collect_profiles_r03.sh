#!/bin/bash
# Round-3 measurement set (runs on the GPU box via gpurun): the default bench line, rocprofv3 kernel stats of the same
# command, the LSTM forward counter passes (tools/r03_pmc_wreg.sh) -- copied into profiles/r03_* afterwards
# (tools/collect_profiles_r03.sh <tag> full; then: for f in gpurun_out/<tag>/*.{json,csv,txt}; do cp $f profiles/r03_$(basename $f); done).
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > $out/bench_metric.json 2> $out/bench_metric.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/ks_bench.json 2> $out/ks.err
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
if [ "$2" = "full" ]; then
  bash tools/r03_pmc_wreg.sh > $out/pmc_wreg.log 2>&1
  cp gpurun_out/r03_pmc_wreg/summary.json $out/lstm_pmc.json 2>/dev/null
  cp gpurun_out/r03_pmc_bwd/summary.json $out/lstm_bwd_pmc.json 2>/dev/null
  timeout 600 python3 bench.py --variant ragged --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_ragged.json 2>/dev/null
  timeout 600 python3 bench.py --forward-only --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_fwd_f32.json 2>/dev/null
  timeout 600 python3 bench.py --forward-only --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fwd_bf16.json 2>/dev/null
  timeout 600 python3 bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_f32_train.json 2>/dev/null
  timeout 600 python3 bench.py --config long_album --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_long_train.json 2>/dev/null
  timeout 600 python3 bench.py --config plumbing --precision f32 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_plumbing.json 2>/dev/null
  timeout 600 python3 bench.py --front-end --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_frontend.json 2>/dev/null
  timeout 600 python3 bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_bf16x3_train.json 2>/dev/null
  # one rank through the launcher: the collective path (RCCL communicator, flat-gradient all-reduce) on the one GPU there is
  FVTA_DIST_FORCE=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_rccl_one_rank.json 2>/dev/null
  PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python3 tools/r03_frontend_ab.py 10 > $out/frontend_kernels.txt 2>/dev/null
fi
head -30 $out/kernel_stats.csv | cut -c1-150
