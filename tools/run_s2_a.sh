#!/bin/bash
# session-2 check: full GPU suite, then the never-yet-run shapes (long album fwd / train, ragged metric)
out=gpurun_out/s2a
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/metric.json 2> $out/metric.err
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant ragged > $out/ragged.json 2> $out/ragged.err
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --forward-only > $out/metric_fwd.json 2> $out/metric_fwd.err
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --forward-only --precision f32 > $out/metric_fwd_f32.json 2> $out/metric_fwd_f32.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config long_album --forward-only > $out/long_fwd.json 2> $out/long_fwd.err
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config long_album > $out/long_train.json 2> $out/long_train.err
tail -3 $out/tests.log; for f in metric ragged metric_fwd metric_fwd_f32 long_fwd long_train; do echo "== $f"; cut -c1-400 $out/$f.json; tail -2 $out/$f.err; done
