#!/bin/bash
# round-2 measurement set A: overlapped LSTM backward tails vs serial
cd "$GRAFT_REPO_ROOT"
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -8
for v in "" "--no-overlap"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms_per_step'])"
done
