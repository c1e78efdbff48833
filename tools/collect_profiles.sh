#!/bin/bash
# Runs on the GPU box (via gpurun): the measurements DESIGN.md / profiles/ quote.  Output under gpurun_out/$1/.
# rocprofv3: the program itself follows `--`; PMC passes are separate from the kernel trace (and from each other:
# FETCH_SIZE and WRITE_SIZE do not fit one pass).
tag=${1:-r01_final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/ks_bench.json 2> $out/ks.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_attn_$c -o p -- python3 tools/bench_attn.py bwd > /dev/null 2> $out/pmc_attn_$c.err
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_lstm_$c -o p -- python3 tools/bench_lstm.py bf16 bwd > /dev/null 2> $out/pmc_lstm_$c.err
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_attn_sq -o p -- python3 tools/bench_attn.py bwd > /dev/null 2> $out/pmc_attn_sq.err
find $out -name "*.csv" | head -40
tail -c 600 $out/bench.json
