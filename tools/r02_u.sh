#!/bin/bash
# attn_fwd_pair16 phase timeline (diagnostics library built beside the product one: csrc/libfvta_hip_diag.so)
cd "$GRAFT_REPO_ROOT"
cp fvta_memexqa_amd/csrc/libfvta_hip_diag.so fvta_memexqa_amd/csrc/libfvta_hip.so
for wv in 0 1 4; do python tools/attn_phases_pair.py $wv 2>&1 | grep -v amdgpu.ids; done
