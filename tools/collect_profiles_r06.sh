#!/bin/bash
# Round-6 measurement set, GPU side (started by tools/collect_r06.sh through gpurun; writes gpurun_out/<tag>/):
# the default bench line (with its `also` block), rocprofv3 kernel stats of the same command, and -- with `full` -- the counter
# passes (separate passes per counter group: FETCH_SIZE and WRITE_SIZE never together, never beside a trace) of the bi-LSTM
# forward / backward / attention kernels, the split engine's kernels, and the one-rank RCCL line.
set -u
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run through gpurun}" || exit 1
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cp tools/.collect_head "$out/head.txt" 2>/dev/null || echo "unknown" > "$out/head.txt"
timeout 1200 python3 bench.py > $out/bench_metric.json 2> $out/bench_metric.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --also off > $out/ks_bench.json 2> $out/ks.err
cp "$(find $out/ks -name "*kernel_stats.csv" | head -1)" $out/kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_ragged -o ks -- python3 bench.py --variant ragged --steps 5 --warmup 2 --no-cpu-baseline --also off > $out/ks_ragged_bench.json 2> $out/ks_ragged.err
cp "$(find $out/ks_ragged -name "*kernel_stats.csv" | head -1)" $out/kernel_stats_ragged.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_x3 -o ks -- python3 bench.py --precision bf16x3 --steps 3 --warmup 2 --no-cpu-baseline --also off > $out/ks_x3_bench.json 2> $out/ks_x3.err
cp "$(find $out/ks_x3 -name "*kernel_stats.csv" | head -1)" $out/kernel_stats_bf16x3.csv 2>/dev/null
pmc() {  # pmc <dir> <script and args...>
  d=$1; shift; mkdir -p $d
  pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $d/$name -o p -- python3 $CMD > $d/$name.log 2> $d/$name.err; }
  CMD="$*"
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  python3 tools/pmc_summary.py $d/summary.json $d/sq $d/sq2 $d/fetch $d/write > $d/summary.txt
}
if [ "${2:-}" = "full" ]; then
  # (FVTA_AB_SKIP=1: the regime of the step -- the context sequences' fp32 rows are not stored, shadow rows)
  export FVTA_AB_SKIP=1; pmc $out/pmc_fwd tools/lstm_fwd_ab.py; unset FVTA_AB_SKIP;  cp $out/pmc_fwd/summary.json $out/lstm_pmc.json
  export FVTA_AB_DXOW=1
  pmc $out/pmc_bwd tools/lstm_bwd_ab.py;          cp $out/pmc_bwd/summary.json $out/lstm_bwd_pmc.json
  pmc $out/pmc_bwd_ragged tools/lstm_bwd_ab.py 12864 30 200 512 ragged; cp $out/pmc_bwd_ragged/summary.json $out/lstm_bwd_ragged_pmc.json
  export FVTA_AB_PREC=2
  pmc $out/pmc_x3_fwd tools/lstm_fwd_ab.py;       cp $out/pmc_x3_fwd/summary.json $out/lstm_bf16x3_pmc.json
  pmc $out/pmc_x3_bwd tools/lstm_bwd_ab.py;       cp $out/pmc_x3_bwd/summary.json $out/lstm_bf16x3_bwd_pmc.json
  unset FVTA_AB_PREC FVTA_AB_DXOW
  pmc $out/pmc_attn tools/bench_attn_shadow.py;   cp $out/pmc_attn/summary.json $out/attention_pmc.json   # fp32-row and shadow-row kernels
  pmc $out/pmc_attn_wide tools/bench_attn_wide.py; cp $out/pmc_attn_wide/summary.json $out/attention_wide_pmc.json
  # one rank through the launcher: the collective path (RCCL communicator, flat-gradient all-reduce) on the one GPU there is
  FVTA_DIST_FORCE=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --also off > $out/bench_rccl_one_rank.json 2>/dev/null
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --also off > $out/bench_plain_same_box.json 2>/dev/null
fi
head -30 $out/kernel_stats.csv | cut -c1-150
