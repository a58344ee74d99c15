#!/bin/bash
# Development aid: builds tools/ubench/libvts_fstamp.so = the library with the fused forward kernel's phase stamps
# compiled in (extra -D flags are passed through).  Needs the objects of a finished `make` in csrc/.
set -e
cd "$(dirname "$0")/../../vae-time-series_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -DVTS_F_STAMP "$@" -c lstm_rec_fused.hip -o /tmp/fstamp.o
objs=$(ls *.o | grep -v lstm_rec_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/libvts_fstamp.so $objs /tmp/fstamp.o
