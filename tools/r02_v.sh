#!/bin/bash
# attn_fwd_pair16 ablations (compile-time, -DFVTA_PAIR_ABL=bits; libraries built beside the product one):
# 1 no tile loads after an item's first, 2 no score MFMA loop, 4 no weighted sum (refill only), 8 no pair hand-shake
cd "$GRAFT_REPO_ROOT"
L=fvta_memexqa_amd/csrc
cp $L/libfvta_hip.so /tmp/libfvta_hip_product.so
sed -i 's/n=5)/n=50)/' tools/bench_attn.py
for v in ${VARIANTS:-product abl6 abl22 abl16 product abl6 abl22 abl16}; do
  if [ $v = product ]; then cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so; else cp $L/libfvta_hip_$v.so $L/libfvta_hip.so; fi
  echo -n "$v: "; python tools/bench_attn.py 2>&1 | grep "attn fwd"
done
cp /tmp/libfvta_hip_product.so $L/libfvta_hip.so
