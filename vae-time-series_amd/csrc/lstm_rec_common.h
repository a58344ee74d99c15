// Pieces shared by the persistent recurrent LSTM kernels (fp32 and bf16x3 variants).
#pragma once
#include "vts_common.h"

namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store (vmcnt(0)), which would drain the prefetched W_hh fragments
// and the whole step's output stores at every time step.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int H>
struct RecCfg {
  static constexpr int JT = H / 16;                 // hidden tiles
  static constexpr int NW = JT >= 4 ? 4 : JT;       // waves per workgroup
  static constexpr int JPW = JT / NW;               // hidden tiles per wave
  static constexpr int KQ_F = H / 16;               // k-quads (16 k each) in the forward product
  static constexpr int KQ_B = (4 * H) / 16;         // k-quads in the backward product
  static constexpr int LDA = 4 * H + 4;             // LDS row stride of the dA tile
};

// Streaming (read-once / write-once) tile16 traffic goes around the L2's normal replacement so that
// the W_hh image every CU of the XCD re-reads each step stays resident.
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
__device__ __forceinline__ void st_stream(float* p, f32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}

// ---- activations --------------------------------------------------------------------------------
// One v_exp_f32 + one v_rcp_f32 each (about 1 ulp apiece): with one wave per SIMD every VALU
// instruction of the cell update sits on the step's critical path, so libm-grade expf/tanhf
// (tens of instructions) would cost a quarter of the step.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}

}  // namespace
