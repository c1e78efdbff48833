#!/bin/bash
# round 3: counter passes over the text-cell bi-LSTM FORWARD (weights-in-registers step kernel) at the metric shape.
# Separate passes per counter group (SQ: 8 slots; TCC: 4; FETCH_SIZE and WRITE_SIZE never together).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_pmc_wreg
mkdir -p $out
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/r03_wreg_ab.py > $out/$name.log 2> $out/$name.err; tail -1 $out/$name.log; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_summary.py $out/summary.json $out/sq $out/sq2 $out/tcc $out/fetch $out/write > /dev/null
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_pmc_wreg/summary.json"))
for k, v in sorted(d.items()):
    if "lstm_fwd_wreg" not in k: continue
    print(k[:70])
    for c, x in sorted(v.items()): print("    %-32s %14.1f" % (c, x))
PY
# ---- the backward kernels of the text cell (fused step, dx, dW): tools/bench_lstm.py bf16 bwd, FETCH / WRITE / SQ passes
outb=gpurun_out/r03_pmc_bwd
mkdir -p $outb
passb() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $outb/$name -o p -- python3 tools/bench_lstm.py bf16 bwd > $outb/$name.log 2> $outb/$name.err; tail -1 $outb/$name.log; }
passb sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
passb fetch FETCH_SIZE
passb write WRITE_SIZE
python3 tools/pmc_summary.py $outb/summary.json $outb/sq $outb/fetch $outb/write > /dev/null
