#!/bin/bash
# which per-CU memory-pipeline unit the LSTM forward step kernel saturates: TA / TCP / TD / LDS counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_pmc_ta
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(TA_|TCP_|TD_|SQ_LDS|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_INST_CYCLES_VMEM|SQ_INSTS_VMEM|SQ_ACTIVE_INST_VMEM|SQ_WAIT_INST|GRBM_GUI|SQ_BUSY_CU)" | head -150 > $out/counters.txt
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA|TCP|TD)_[A-Z0-9_]+" | sort -u | head -200 > $out/names.txt
wc -l $out/names.txt; head -100 $out/names.txt | tr '\n' ' '
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/bench_lstm.py bf16 > $out/$name.log 2> $out/$name.err; tail -1 $out/$name.log; grep -i -E "error|invalid|not" $out/$name.err | head -3; }
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_STORE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
FVTA_LSTM_WIDE_TILE=1 pass ta_wide TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_STORE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
FVTA_LSTM_WIDE_TILE=1 pass sq2_wide SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
for d in ta tcp sq2 ta_wide sq2_wide; do python3 tools/pmc_summary.py $out/$d.json $out/$d > /dev/null 2>&1; python3 - <<PY
import json
try:
    d=json.load(open("$out/$d.json"))
    for k,v in d.items():
        if "lstm_step_fwd" in k: print("$d", k[:50], {c: round(x,1) for c,x in v.items()})
except Exception as e: print("$d ERR", e)
PY
done
