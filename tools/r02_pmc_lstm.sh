#!/bin/bash
# round-2: counter passes over the text-cell bi-LSTM kernels at the metric shape (tools/bench_lstm.py bf16 bwd).
# Separate passes per counter group (SQ: 8 slots; TCC: 4; FETCH_SIZE and WRITE_SIZE never together).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_pmc_lstm
mkdir -p $out
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/bench_lstm.py bf16 bwd > $out/$name.log 2> $out/$name.err; tail -2 $out/$name.log; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_summary.py $out/summary.json $out/sq $out/tcc $out/fetch $out/write
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r02_pmc_lstm/summary.json"))
for k, v in sorted(d.items()):
    if "lstm" not in k: continue
    print(k[:60])
    print("   ", {c: round(x, 1) for c, x in v.items()})
PY
