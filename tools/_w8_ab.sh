# correctness (fp32 rows stored) then timing of ablations
export FVTA_AB_SAVE=1
for m in 0 7 15; do FVTA_LSTM_WREG=$m timeout 300 python tools/lstm_fwd_ab.py 13120 30 200 512 2>&1 | tail -2; done
for m in 0 15; do FVTA_LSTM_WREG=$m timeout 300 python tools/lstm_fwd_ab.py 3000 20 100 512 ragged 2>&1 | tail -2; done
export FVTA_AB_SKIP=1
for b in 1 2 4 16 3; do echo "abl $b"; FVTA_LIB_PATH=fvta_memexqa_amd/csrc/diag/libfvta_hip_lstm_wreg8_abl$b.so FVTA_LSTM_WREG=15 timeout 300 python tools/lstm_fwd_ab.py 13120 30 200 512 2>&1 | grep fwd; done
