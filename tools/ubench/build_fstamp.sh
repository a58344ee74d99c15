#!/bin/bash
# Development aid: builds tools/ubench/libvts_fstamp${TAG:+_$TAG}.so = the library with the fused forward kernel's
# phase stamps compiled in (extra -D flags are passed through; NOSTAMP=1 leaves the stamps out).  Needs the objects
# of a finished `make` in csrc/.   TAG=x NOSTAMP=1 tools/ubench/build_fstamp.sh -DVTS_F_NOSTORE
set -e
cd "$(dirname "$0")/../../vae-time-series_amd/csrc"
tag=${TAG:+_$TAG}
stamp=-DVTS_F_STAMP
[ -n "$NOSTAMP" ] && stamp=
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm $stamp "$@" -c lstm_rec_fused.hip -o /tmp/fstamp$tag.o
objs=$(ls *.o | grep -v lstm_rec_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/libvts_fstamp$tag.so $objs /tmp/fstamp$tag.o
