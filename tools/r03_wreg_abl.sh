#!/bin/bash
# ablations of the weights-in-registers forward step (DIAG library): FVTA_DEBUG_SKIP bits 1 no gate stages, 2 no MFMAs,
# 4 no activation DMA, 8 no weight load, 16 no stores
for m in 0 1 16 2 4 8 3 7 15 31; do
  echo "== FVTA_DEBUG_SKIP=$m"; FVTA_DEBUG_SKIP=$m python tools/r03_wreg_ab.py 2>&1 | grep fwd
done
