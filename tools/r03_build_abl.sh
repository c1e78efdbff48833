#!/bin/bash
# builds diag/libfvta_hip_abl<bits>.so variants of the library with lstm_wreg.hip compiled -DFVTA_WREG_ABL=<bits>
cd "$(dirname "$0")/../fvta_memexqa_amd/csrc" && mkdir -p diag
for b in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DFVTA_WREG_ABL=$b -c lstm_wreg.hip -o diag/lstm_wreg_abl$b.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o diag/libfvta_hip_abl$b.so $(ls *.o | grep -v '^lstm_wreg.o$') diag/lstm_wreg_abl$b.o ) &
done
wait
ls -la diag/*.so
