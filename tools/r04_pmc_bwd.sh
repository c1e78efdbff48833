#!/bin/bash
# round 4: counter passes over the text-cell bi-LSTM BACKWARD at the metric shape (tools/r04_ring_ab.py): the step kernel
# (FVTA_LSTM_WREG picks it: 7 lstm_bwd_ring_bf16, 3 lstm_bwd_fused_bf16), dx, dW.  Separate passes per counter group.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r04_pmc_bwd}
mkdir -p $out
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/r04_ring_ab.py > $out/$name.log 2> $out/$name.err; tail -1 $out/$name.log; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/pmc_summary.py $out/summary.json $out/sq $out/sq2 $out/tcc $out/fetch $out/write > /dev/null
python3 - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary.json"))
for k, v in sorted(d.items()):
    if "lstm_bwd" not in k and "lstm_dx" not in k and "lstm_dw_bf16" not in k: continue
    print(k[:90])
    for c, x in sorted(v.items()): print("    %-32s %14.1f" % (c, x))
PY
