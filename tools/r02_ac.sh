#!/bin/bash
# fused backward step epilogue: rows whose loads are in flight together (-DFVTA_BWD_EPI_ROWS=2 / 4 (product) / 8), dense + ragged
cd "$GRAFT_REPO_ROOT"
L=fvta_memexqa_amd/csrc
cp $L/libfvta_hip.so /tmp/libfvta_hip_product.so
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', d['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'], 'dw', k['lstm_dw'])"; }
for v in rb2 product rb8 rb2 product rb8; do
  if [ $v = product ]; then cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so; else cp $L/libfvta_hip_$v.so $L/libfvta_hip.so; fi
  run "dense $v" ""; run "ragged $v" "--variant ragged"
done
cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so
