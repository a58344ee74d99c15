#!/bin/bash
# Development aid: builds tools/ubench/libvts_cstamp.so = the library with the cluster recurrence's phase stamps
# compiled in (extra -D flags are passed through).  Needs the objects of a finished `make` in csrc/.
set -e
cd "$(dirname "$0")/../../vae-time-series_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVTS_C_STAMP -DVTS_CB_STAMP "$@" -c lstm_rec_cluster.hip -o /tmp/cstamp_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/${CSTAMP_OUT:-libvts_cstamp.so} vts_api.o gemm_f32.o lstm_rec_f32.o lstm_rec_bf16x3.o lstm_rec_pair.o lstm_rec_fused.o /tmp/cstamp_$$.o lstm_planes.o mlp_chain.o elementwise.o attention.o ffn_fused.o windows.o mixture3.o
