// Shared declarations for the vts HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#define VTS_OK 0
#define VTS_ERR_ARG 1      // bad shape / pointer / alignment
#define VTS_ERR_HIP 2      // a HIP runtime call failed
#define VTS_ERR_UNSUPPORTED 3

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Thread-local message for vts_last_error().
void vts_set_error(const char* fmt, ...);

#define VTS_REQUIRE(cond, ...)                    \
  do {                                            \
    if (!(cond)) {                                \
      vts_set_error(__VA_ARGS__);                 \
      return VTS_ERR_ARG;                         \
    }                                             \
  } while (0)

#define VTS_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      vts_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));  \
      return VTS_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

// Kernels that use more than 64 KiB of dynamic LDS must opt in with hipFuncSetAttribute, which
// applies to the current device only: `done` (one static atomic word per call site) remembers which
// devices of this process have been configured.  Calls come from any host thread (autograd runs
// backward on its own), hence the atomic; setting the attribute twice is harmless, so no lock.
typedef std::atomic<unsigned long long> vts_lds_mask;
static inline int vts_reserve_lds(const void* kernel, size_t bytes, vts_lds_mask* done, const char* what) {
  if (bytes <= 64 * 1024) return VTS_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done->load(std::memory_order_acquire) & bit) return VTS_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    vts_set_error("%s: cannot reserve %zu B of LDS: %s", what, bytes, hipGetErrorString(e));
    return VTS_ERR_HIP;
  }
  done->fetch_or(bit, std::memory_order_release);
  return VTS_OK;
}

// A stream that is recording a graph takes no cooperative launch; the launchers of the cooperative kernels ask
// first and report VTS_ERR_UNSUPPORTED with nothing enqueued, so that their callers take the plain kernels.
static inline bool vts_stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs != hipStreamCaptureStatusNone;
}

static inline bool vts_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int vts_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// "tile16" layout of a logical [M', N] fp32 matrix (M' and N multiples of 16): 16x16 blocks,
// block (mt, nt) at ((mt * N/16) + nt) * 256 floats; inside a block element (r, c) sits at
// ((r >> 2) * 16 + c) * 4 + (r & 3).  That is exactly the C/D register image of
// v_mfma_f32_16x16x4_f32 (lane l = (r>>2)*16 + c holds rows 4*(l>>4)..+3 of column l&15), so a
// wave stores or loads one block as 64 lanes x 16 B, fully coalesced, and the same float4 is
// also a ready-made A fragment (4 consecutive k for one column) for a product that contracts
// over the block's row index.
__host__ __device__ static inline long vts_tile16_offset(long m, long n, long N) {
  return ((m >> 4) * (N >> 4) + (n >> 4)) * 256 + ((((m & 15) >> 2) * 16 + (n & 15)) << 2) + (m & 3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store (vmcnt(0)), which drains prefetches issued before it.
static __device__ __forceinline__ void vts_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- bf16x3 operand split ----------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// x -> (hi, lo) for two values at a time: one packed round-to-nearest conversion gives both hi
// halves, two integer ops widen them back to fp32, one packed conversion rounds the residuals
// (6 VALU instructions per pair; v_cvt_pk_bf16_f32 has no builtin, a plain cast costs ~9).
static __device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
static __device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  lo = cvt_pk_bf16(x0 - h0, x1 - h1);
}
static __device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo) {
  split2(x0, x1, hi.x, lo.x);
  split2(x2, x3, hi.y, lo.y);
}

// ---- counter-based dropout ----------------------------------------------------------------------
// One hash serves two elements: element `idx` of a tensor whose mask was drawn with `seed` takes the low
// (idx even) or the high (idx odd) 16 bits of vts_hash32(seed, idx >> 1) and is kept iff they are >= thresh,
// thresh = round(p * 2^16) (the drop probability is p to within 2^-17).  A pure function of (seed, idx): the
// backward pass and any number of kernels regenerate the same mask without storing it.  The hash is one round of a
// multiply/xor-shift mixer over the low index word, with the high word and the second seed word folded in; its two
// 32-bit multiplies run at a quarter of the vector rate, which is why the kernels that draw a mask per matrix-core
// operand (the feed-forward block, the attention weights) share it between neighbours.
__host__ __device__ static inline unsigned vts_hash32(unsigned long long seed, unsigned long long idx) {
  unsigned x = (unsigned)idx ^ (unsigned)seed;
  x += (unsigned)(idx >> 32) * 0x9E3779B9u + (unsigned)(seed >> 32);
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ static inline bool vts_drop_keep(unsigned long long seed, unsigned long long idx, unsigned thresh) {
  const unsigned h = vts_hash32(seed, idx >> 1);
  return ((idx & 1) ? h >> 16 : h & 0xffffu) >= thresh;
}
// keep decisions of elements idx and idx + 1 for an EVEN idx: one hash
__host__ __device__ static inline void vts_drop_keep2(unsigned long long seed, unsigned long long idx_even, unsigned thresh,
                                                      bool& k0, bool& k1) {
  const unsigned h = vts_hash32(seed, idx_even >> 1);
  k0 = (h & 0xffffu) >= thresh;
  k1 = (h >> 16) >= thresh;
}
// ... of idx .. idx + 3: two hashes when idx is even, the general route otherwise
__host__ __device__ static inline void vts_drop_keep4(unsigned long long seed, unsigned long long idx, unsigned thresh,
                                                      bool (&k)[4]) {
  if ((idx & 1) == 0) {
    vts_drop_keep2(seed, idx, thresh, k[0], k[1]);
    vts_drop_keep2(seed, idx + 2, thresh, k[2], k[3]);
  } else {
    for (int e = 0; e < 4; ++e) k[e] = vts_drop_keep(seed, idx + e, thresh);
  }
}
static inline unsigned vts_drop_thresh(float p) {
  if (!(p > 0.f)) return 0u;
  const double t = (double)p * 65536.0 + 0.5;
  return t >= 65535.0 ? 65535u : (t < 1.0 ? 1u : (unsigned)t);
}

// How a dropout site gets its mask: explicit bytes (tests, replay of recorded draws), the counter
// hash (training), or nothing (eval / p == 0).  mult(idx) is the factor applied to element idx.
struct VtsDrop {
  const unsigned char* keep;   // [n] 0/1 or null
  unsigned long long seed;
  unsigned thresh;             // 0 = no hashed dropout
  float scale;                 // 1 / (1 - p)
  __device__ __forceinline__ float mult(unsigned long long idx) const {
    if (keep) return keep[idx] ? scale : 0.f;
    if (thresh) return vts_drop_keep(seed, idx, thresh) ? scale : 0.f;
    return 1.f;
  }
  // the factors of elements idx and idx + 1, idx EVEN (one hash for the two)
  __device__ __forceinline__ void mult2(unsigned long long idx_even, float& m0, float& m1) const {
    if (keep) {
      m0 = keep[idx_even] ? scale : 0.f;
      m1 = keep[idx_even + 1] ? scale : 0.f;
    } else if (thresh) {
      bool k0, k1;
      vts_drop_keep2(seed, idx_even, thresh, k0, k1);
      m0 = k0 ? scale : 0.f;
      m1 = k1 ? scale : 0.f;
    } else {
      m0 = m1 = 1.f;
    }
  }
};
static inline VtsDrop vts_make_drop(const unsigned char* keep, float p, unsigned long long seed) {
  VtsDrop d{};
  if (p > 0.f && (keep || seed)) {
    d.keep = keep;
    d.seed = seed;
    d.thresh = keep ? 0u : vts_drop_thresh(p);
    d.scale = 1.f / (1.f - p);
  } else {
    d.scale = 1.f;
  }
  return d;
}
