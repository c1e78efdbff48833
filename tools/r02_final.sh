#!/bin/bash
# final round-2 set: full GPU suite, then the measurement collection and the LSTM counter passes
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/collect_profiles_r02.sh r02 > gpurun_out/r02_collect.log 2>&1
bash tools/r02_pmc_lstm.sh > gpurun_out/r02_pmc_lstm.log 2>&1
tail -3 gpurun_out/r02_collect.log
