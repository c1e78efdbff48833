#!/bin/bash
# last check of the round: full -m gpu suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02_final3_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r02_final3_tests.log
timeout 900 python bench.py > gpurun_out/r02_final3_bench.json 2> gpurun_out/r02_final3_bench.err
