#!/bin/bash
# Round-2 measurement set (runs on the GPU box via gpurun): bench lines, rocprofv3 kernel stats of the same command,
# counter passes for the attention kernels (FETCH / WRITE / SQ) -- the LSTM counter passes are tools/r02_pmc_lstm.sh.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > $out/bench_metric.json 2> $out/bench_metric.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/ks_bench.json 2> $out/ks.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_attn_$c -o p -- python3 tools/bench_attn.py bwd 2 > $out/pmc_attn_$c.log 2> $out/pmc_attn_$c.err
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_attn_sq -o p -- python3 tools/bench_attn.py bwd 2 > /dev/null 2> $out/pmc_attn_sq.err
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_attn_tcc -o p -- python3 tools/bench_attn.py bwd 2 > /dev/null 2> $out/pmc_attn_tcc.err
python3 tools/pmc_summary.py $out/attention_pmc.json $out/pmc_attn_FETCH_SIZE $out/pmc_attn_WRITE_SIZE $out/pmc_attn_sq $out/pmc_attn_tcc
timeout 600 python3 bench.py --variant ragged --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_ragged.json 2>/dev/null
timeout 600 python3 bench.py --forward-only --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_fwd_f32.json 2>/dev/null
timeout 600 python3 bench.py --forward-only --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fwd_bf16.json 2>/dev/null
timeout 600 python3 bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_f32_train.json 2>/dev/null
timeout 600 python3 bench.py --config long_album --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_long_train.json 2>/dev/null
timeout 600 python3 bench.py --config plumbing --precision f32 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_plumbing.json 2>/dev/null
timeout 600 python3 bench.py --front-end --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_frontend.json 2>/dev/null
find $out -name "*stats*.csv" | head; ls $out
