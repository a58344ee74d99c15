#!/bin/bash
# Development aid: builds tools/ubench/libvts_<name>.so = the library with one source file recompiled under extra
# -D flags.  Needs the objects of a finished `make` in csrc/.
#   bash tools/ubench/build_variant.sh <name> <source.hip> [-DFLAG ...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../vae-time-series_amd/csrc"
obj=${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src -o /tmp/variant_$name.o 2>/dev/null
objs=$(ls *.o | grep -v "^$obj$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/libvts_$name.so $objs /tmp/variant_$name.o
echo built tools/ubench/libvts_$name.so
