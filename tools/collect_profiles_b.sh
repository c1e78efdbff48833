#!/bin/bash
# Second measurement set of round 1 (gpurun): bench lines for every BASELINE.json config + the kernel trace of the
# headline run.  Output under gpurun_out/$1/; the files quoted in DESIGN.md are copied to profiles/ by hand.
tag=${1:-r01b}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --steps 50 --warmup 10 > $out/bench_metric.json 2> $out/bench_metric.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/ks_bench.json 2> $out/ks.err
timeout 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --variant ragged > $out/bench_ragged.json 2> $out/bench_ragged.err
timeout 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --forward-only --precision f32 > $out/bench_fwd_f32.json 2> $out/bench_fwd_f32.err
timeout 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --forward-only > $out/bench_fwd_bf16.json 2> $out/bench_fwd_bf16.err
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --front-end > $out/bench_frontend.json 2> $out/bench_frontend.err
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config long_album > $out/bench_long_train.json 2> $out/bench_long_train.err
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config long_album --forward-only > $out/bench_long_fwd.json 2> $out/bench_long_fwd.err
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --config plumbing --precision f32 > $out/bench_plumbing.json 2> $out/bench_plumbing.err
for f in $out/bench_*.json; do echo "== $f"; cut -c1-260 $f; done
find $out -name "*stats*.csv" | head
