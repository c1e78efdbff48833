#!/bin/bash
# full -m gpu suite, smoke, then the final round-2 measurement set
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_final2_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r02_final2_tests.log 2>&1
bash tools/collect_profiles_r02.sh r02c > gpurun_out/r02_final2_collect.log 2>&1
