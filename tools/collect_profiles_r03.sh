#!/bin/bash
# Round-3 measurement set (runs on the GPU box via gpurun): the default bench line, rocprofv3 kernel stats of the same
# command, the LSTM forward counter passes (tools/r03_pmc_wreg.sh) -- copied into profiles/r03_* afterwards.
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
  timeout 600 python3 bench.py --variant ragged --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_ragged.json 2>/dev/null
  timeout 600 python3 bench.py --forward-only --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_fwd_f32.json 2>/dev/null
  timeout 600 python3 bench.py --forward-only --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fwd_bf16.json 2>/dev/null
  timeout 600 python3 bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_f32_train.json 2>/dev/null
  timeout 600 python3 bench.py --config long_album --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_long_train.json 2>/dev/null
  timeout 600 python3 bench.py --config plumbing --precision f32 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_plumbing.json 2>/dev/null
  timeout 600 python3 bench.py --front-end --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_frontend.json 2>/dev/null
fi
head -30 $out/kernel_stats.csv | cut -c1-150
