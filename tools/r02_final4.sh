#!/bin/bash
# final measurement set of round 2 (after the last kernel change): stress loops, then collect_profiles_r02.sh r02c
cd "$GRAFT_REPO_ROOT"
python tools/stress_attn.py 2>&1 | tail -1 > gpurun_out/r02_final4_stress.log
python tools/stress_forward.py 2>&1 | tail -1 >> gpurun_out/r02_final4_stress.log
bash tools/collect_profiles_r02.sh r02c > gpurun_out/r02_final4_collect.log 2>&1
