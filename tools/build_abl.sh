#!/bin/bash
# builds diag/libfvta_hip_<src>_abl<bits>.so variants of the library with one source compiled -D<MACRO>=<bits>
#   tools/build_abl.sh 1 2 4                                    (lstm_wreg.hip, -DFVTA_WREG_ABL)
#   SRC=embed MACRO=FVTA_EMB_ABL tools/build_abl.sh 1 2 4       (embed.hip)
SRC=${SRC:-lstm_wreg}; MACRO=${MACRO:-FVTA_WREG_ABL}
cd "$(dirname "$0")/../fvta_memexqa_amd/csrc" && mkdir -p diag
for b in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -D$MACRO=$b -c $SRC.hip -o diag/${SRC}_abl$b.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o diag/libfvta_hip_${SRC}_abl$b.so $(ls *.o | grep -v "^$SRC.o\$") diag/${SRC}_abl$b.o ) &
done
wait
ls -la diag/*.so
