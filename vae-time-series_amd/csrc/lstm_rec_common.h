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

// Tail of the weight-packing kernels of the cluster and pair recurrences: the launch's hand-off words (n16 pieces of
// 16 bytes) set to `pattern` and its status word cleared, so that no memset launch stands in front of the recurrence.
__device__ __forceinline__ void reset_handoff(void* words, long n16, unsigned pattern, unsigned* status) {
  typedef unsigned rh4 __attribute__((ext_vector_type(4)));
  rh4* p = reinterpret_cast<rh4*>(words);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
    p[i] = rh4{pattern, pattern, pattern, pattern};
  if (status != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *status = 0u;
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- W_hh register ring ---------------------------------------------------------------------------
// The recurrent kernels consume the packed W_hh image in units of UT 16-byte fragments per lane
// (1 KiB per fragment and wave).  A ring of four register buffers keeps three units in flight.
// The loads are issued as inline asm and waited for with
// hand-counted s_waitcnt: left to the compiler, the wait at the loop back-edge is vmcnt(0), which
// drains the ring once per iteration and exposes a full L2 round trip each time.  Loads return in
// order, so "at most N younger VMEM operations outstanding" proves a buffer has landed; operations
// the compiler issues in between (gate loads, the previous step's stores) only make a given N
// stricter.  The waits carry the buffer as an in/out operand so that no use can move above them,
// and so do the loads: every definition of a ring buffer is tied to the previous one, which keeps
// each buffer in the same physical registers around the loops.  (A fresh output register per
// load lets the allocator pick different registers per iteration and reconcile them with moves
// at the back-edge - moves that would copy registers whose data is still in flight.)
// tools/check_ring_registers.py verifies on the generated ISA that nothing but these loads, the
// LDS reads of the resident part and the MFMAs ever touches the ring registers.
// AG selects the register file the buffer is pinned to (false: VGPRs, true: AGPRs - both are legal
// load destinations and MFMA B operands; splitting the ring over the two files keeps the arch
// VGPRs free for the cell update's VALU work).
#define VTS_W_LOAD(CONS, REG, VOFF, K)                                                              \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : CONS(REG) : "v"(VOFF), "s"(sbase), "n"((K) * 1024))
template <int UT, bool AG = false>
__device__ __forceinline__ void w_issue(u32x4 (&w)[UT], const void* sbase, unsigned voff) {
  static_assert(UT == 2 || UT == 4 || UT == 8, "half-unit of 2, 4 or 8 fragments");
#pragma unroll
  for (int i = 0; i < UT; ++i) {
    const unsigned vo = i < 4 ? voff : voff + 4096u;
    if constexpr (AG) VTS_W_LOAD("+a", w[i], vo, i & 3);
    else VTS_W_LOAD("+v", w[i], vo, i & 3);
  }
}
#undef VTS_W_LOAD
template <int N, int UT, bool AG = false>
__device__ __forceinline__ void w_wait(u32x4 (&w)[UT]) {
  if constexpr (AG) {
    if constexpr (UT == 8)
      asm volatile("s_waitcnt vmcnt(%8)" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]), "+a"(w[4]), "+a"(w[5]),
                   "+a"(w[6]), "+a"(w[7]) : "n"(N));
    else if constexpr (UT == 4)
      asm volatile("s_waitcnt vmcnt(%4)" : "+a"(w[0]), "+a"(w[1]), "+a"(w[2]), "+a"(w[3]) : "n"(N));
    else
      asm volatile("s_waitcnt vmcnt(%2)" : "+a"(w[0]), "+a"(w[1]) : "n"(N));
  } else {
    if constexpr (UT == 8)
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]),
                   "+v"(w[6]), "+v"(w[7]) : "n"(N));
    else if constexpr (UT == 4)
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
    else
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N));
  }
}

}  // namespace
