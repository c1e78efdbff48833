#!/bin/bash
# char-CNN forward of the default shape (5 x 8) on the matrix-pipe kernel: FVTA_EMBED_MFMA_SMALL
cd "$GRAFT_REPO_ROOT"
FVTA_EMBED_MFMA_SMALL=1 python -m pytest tests/test_gpu_embed.py tests/test_gpu_feed.py -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --front-end --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_event_median'])"; }
FVTA_EMBED_MFMA_SMALL=0 run "small_mfma=0"
FVTA_EMBED_MFMA_SMALL=1 run "small_mfma=1"
FVTA_EMBED_MFMA_SMALL=0 run "small_mfma=0"
FVTA_EMBED_MFMA_SMALL=1 run "small_mfma=1"
