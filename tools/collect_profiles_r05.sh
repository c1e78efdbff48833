#!/bin/bash
# Round-5 measurement set (runs on the GPU box via gpurun): the default bench line (with its `also` block), rocprofv3
# kernel stats of the same command, counter passes of the bi-LSTM forward / backward / attention kernels as they are
# now, the one-rank RCCL line -- copied into profiles/r05_* afterwards:
#   gpurun -- tools/collect_profiles_r05.sh r05 full;  for f in gpurun_out/r05/*.{json,csv,txt}; do cp $f profiles/r05_$(basename $f); done
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > $out/bench_metric.json 2> $out/bench_metric.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --also off > $out/ks_bench.json 2> $out/ks.err
cp $(find $out/ks -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
pmc() {  # pmc <dir> <script and args...>: separate passes per counter group (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE never together)
  d=$1; shift; mkdir -p $d
  pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $d/$name -o p -- python3 $CMD > $d/$name.log 2> $d/$name.err; }
  CMD="$*"
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  python3 tools/pmc_summary.py $d/summary.json $d/sq $d/sq2 $d/fetch $d/write > $d/summary.txt
}
if [ "$2" = "full" ]; then
  # (FVTA_AB_SKIP=1: the regime of the step -- the context sequences' fp32 rows are not stored, shadow rows)
  export FVTA_AB_SKIP=1; pmc $out/pmc_fwd tools/lstm_fwd_ab.py; unset FVTA_AB_SKIP;  cp $out/pmc_fwd/summary.json $out/lstm_pmc.json
  pmc $out/pmc_bwd tools/lstm_bwd_ab.py;          cp $out/pmc_bwd/summary.json $out/lstm_bwd_pmc.json
  pmc $out/pmc_bwd_ragged tools/lstm_bwd_ab.py 12864 30 200 512 ragged; cp $out/pmc_bwd_ragged/summary.json $out/lstm_bwd_ragged_pmc.json
  pmc $out/pmc_attn tools/bench_attn_shadow.py;   cp $out/pmc_attn/summary.json $out/attention_pmc.json   # fp32-row and shadow-row kernels
  pmc $out/pmc_attn_wide tools/bench_attn_wide.py; cp $out/pmc_attn_wide/summary.json $out/attention_wide_pmc.json
  timeout 300 python3 tools/bench_attn_wide.py > $out/attention_wide.txt 2>/dev/null; timeout 300 python3 tools/bench_attn_wide.py exact >> $out/attention_wide.txt 2>/dev/null
  # one rank through the launcher: the collective path (RCCL communicator, flat-gradient all-reduce) on the one GPU there is
  FVTA_DIST_FORCE=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --also off > $out/bench_rccl_one_rank.json 2>/dev/null
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --also off > $out/bench_plain_same_box.json 2>/dev/null
  FVTA_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 timeout 600 python3 tools/rccl_probe.py > $out/rccl_probe.txt 2>/dev/null
fi
head -30 $out/kernel_stats.csv | cut -c1-150
