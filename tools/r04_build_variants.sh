#!/bin/bash
# builds diag/libfvta_hip_<name>.so variants of the library with ONE source compiled with extra -D flags:
#   SRC=lstm_wreg_bwd tools/r04_build_variants.sh abl1:-DFVTA_RING_ABL=1 look4:-DFVTA_RING_LOOK=4 ...
SRC=${SRC:-lstm_wreg_bwd}
cd "$(dirname "$0")/../fvta_memexqa_amd/csrc" && mkdir -p diag
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $flags -c $SRC.hip -o diag/${SRC}_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o diag/libfvta_hip_$name.so $(ls *.o | grep -v "^$SRC.o\$") diag/${SRC}_$name.o ) &
done
wait
ls diag/*.so
