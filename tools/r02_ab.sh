#!/bin/bash
# timing experiment: the fused backward step's A operand (dz of step t+1) addressed as if stored k-tile-major
# (a library built with -DFVTA_BWD_FAKE_BLOCKED beside the product one; numerically garbage)
cd "$GRAFT_REPO_ROOT"
L=fvta_memexqa_amd/csrc
cp $L/libfvta_hip.so /tmp/libfvta_hip_product.so
run() { python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', d['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'], 'dw', k['lstm_dw'])"; }
for v in product fakeblk product fakeblk; do
  if [ $v = product ]; then cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so; else cp $L/libfvta_hip_$v.so $L/libfvta_hip.so; fi
  run "dense $v" ""
done
cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so
