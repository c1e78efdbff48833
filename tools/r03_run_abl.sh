#!/bin/bash
# times the text-cell forward with each diag/libfvta_hip_abl<bits>.so given (plus the product library first)
python tools/r03_wreg_ab.py 2>&1 | grep fwd | sed 's/^/product: /'
for b in "$@"; do
  FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_wreg_abl$b.so python tools/r03_wreg_ab.py 2>&1 | grep fwd | sed "s/^/abl $b: /"
done
for nt in 1 3; do FVTA_LSTM_NT=$nt python tools/r03_wreg_ab.py 2>&1 | grep fwd | sed "s/^/product FVTA_LSTM_NT=$nt: /"; done
