// Feed-forward block of an encoder layer as two back-to-back products in one kernel:
//   forward : hidden = dropout(relu(x . W1^T + b1)),  y  = hidden . W2^T + b2
//   backward: dpre   = mask * (dy . W2),              dx = dpre . W1        (+ column sums of dpre = db1)
// for model width d = 64 and a hidden width F that is a multiple of 64 (2048 in
// nn.TransformerEncoderLayer's default, vae-models/mixture-vae-transformer.py:39).
//
// Why fuse: each product alone is short (contraction 64) or narrow (64 outputs), so as separate
// GEMMs one workgroup's life is a load, a couple of dozen MFMAs and a long epilogue, and the
// [tokens, F] intermediate crosses HBM between them.  Here a workgroup owns 128 tokens for the
// whole block.  Their 64 input features are split into bf16 head/tail fragments once and stay in
// registers; the weights stream through LDS in chunks of 64 hidden units (both matrices are
// L2-resident, 512 KiB each); a chunk of the intermediate goes from the first product's
// accumulators through the epilogue straight into LDS as the second product's operand, and is
// written to HBM exactly once on the way (backward needs it).  The first product is computed
// transposed (hidden x tokens), which leaves every lane with four consecutive hidden units of one
// token: a ready 8-byte piece of the second product's A operand and a float4 of the HBM row.
//
// Arithmetic is the split-bf16 scheme of the GEMMs (x = hi + lo, three bf16 MFMAs per product, fp32
// accumulation).  8 waves per workgroup, wave w owns tokens 16w .. 16w+15 in both products, so the
// intermediate never crosses waves and only the weight chunks need workgroup barriers.
#include "vts_common.h"

namespace {

constexpr int D = 64;            // model width (contraction of product 1, output width of product 2)
constexpr int HC = 64;           // hidden units per chunk
constexpr int TOK = 128;         // tokens per workgroup
constexpr int ROWB = 64;         // LDS row pitch: 32 bf16
// LDS map (bytes).  A "plane" holds rows x 32 k-values of bf16; [kstep][hi|lo] planes per operand.
constexpr int W_PLANE = 64 * ROWB;               // 4 KiB: 64 weight rows
constexpr int S_PLANE = TOK * ROWB;              // 8 KiB: 128 token rows
constexpr int OFF_W1 = 0;                        // [2][2] W_PLANE  = 16 KiB   rows = hidden, k = d
constexpr int OFF_W2 = OFF_W1 + 4 * W_PLANE;     // [2][2] W_PLANE  = 16 KiB   rows = out,    k = hidden
constexpr int OFF_S = OFF_W2 + 4 * W_PLANE;      // [2][2] S_PLANE  = 32 KiB   rows = token,  k = hidden (also stages x)
constexpr int OFF_CS = OFF_S + 4 * S_PLANE;      // [2][8][64] floats = 4 KiB  column sums per wave, two chunks in flight (backward)
constexpr int OFF_B1 = OFF_CS + 2 * 8 * 64 * 4;  // [64] floats: the chunk's slice of b1 (forward)
constexpr int LDS_BYTES = OFF_B1 + 64 * 4;

struct FfnParams {
  const float* x;        // [M][ldx]   forward: layer input; backward: dy
  long ldx;
  const float* wa;       // [F][64]    product 1 weights, row = hidden unit: forward W1, backward W2^T
  const float* wb;       // [64][F]    product 2 weights, row = output:      forward W2, backward W1^T
  const float* b1;       // [F]  forward
  const float* b2;       // [64] forward
  const float* keep;     // [M][F] float 0/1 or null (forward)
  unsigned long long seed;
  unsigned thresh;       // hashed dropout when keep == null and thresh != 0
  float scale;           // 1/(1-p) applied to kept / unmasked elements
  float* mid;            // [M][F]  forward: hidden (out); backward: dpre (out)
  const float* hidden;   // backward: forward's hidden (mask source when bits == null)
  unsigned long long* bits_out;        // forward: [M][F/64] mask words or null
  const unsigned long long* bits_in;   // backward
  float* out;            // [M][ldo]  y / dx
  long ldo;
  float* colpart;        // backward: [ceil(M/128)][F] column sums of dpre
  int M, F;
};

__device__ __forceinline__ int sw_off(int row, int kbyte) {   // bank swizzle shared by every plane
  const int prow = (row & ~3) | ((row & 3) ^ ((row >> 4) & 3));
  return prow * ROWB + ((((kbyte >> 4) ^ ((row >> 2) & 3)) << 4) | (kbyte & 15));
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// 8 consecutive floats -> 16 bytes of bf16 heads and 16 bytes of tails
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, uint4& hi, uint4& lo) {
  uint2 h0, l0, h1, l1;
  split4(a[0], a[1], a[2], a[3], h0, l0);
  split4(b[0], b[1], b[2], b[3], h1, l1);
  hi = uint4{h0.x, h0.y, h1.x, h1.y};
  lo = uint4{l0.x, l0.y, l1.x, l1.y};
}

// DROP (forward): 0 no dropout, 1 float keep mask, 2 counter hash.  BITS: forward writes / backward reads
// the packed mask (backward without it compares the saved hidden activations with 0).
template <bool BWD, int DROP, bool BITS>
__global__ __launch_bounds__(512, 2) void ffn_b2b_kernel(const FfnParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * TOK;
  const int nchunks = P.F / HC;

  // ---- x tile -> LDS (S region) -> this wave's B fragments of product 1, kept for the whole loop
  {
    const int row = tid >> 2, kq = tid & 3;            // 16 floats: k = 16*kq ..
    f32x4 v[4];
    const bool ok = row0 + row < P.M;
    const float* src = P.x + (long)(row0 + row) * P.ldx + kq * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ok ? *reinterpret_cast<const f32x4*>(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    char* base = lds + OFF_S + (kq >> 1) * 2 * S_PLANE;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 hi, lo;
      split8(v[2 * half], v[2 * half + 1], hi, lo);
      const int off = sw_off(row, (kq & 1) * 32 + half * 16);
      *reinterpret_cast<uint4*>(base + off) = hi;
      *reinterpret_cast<uint4*>(base + S_PLANE + off) = lo;
    }
  }
  __syncthreads();
  bf16x8 xh[2], xl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int off = sw_off(w * 16 + r, g * 16);
    xh[ks] = *reinterpret_cast<const bf16x8*>(lds + OFF_S + ks * 2 * S_PLANE + off);
    xl[ks] = *reinterpret_cast<const bf16x8*>(lds + OFF_S + ks * 2 * S_PLANE + S_PLANE + off);
  }

  f32x4 acc2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weight chunk staging: thread -> (row tid/8, 8 floats at k = 8*(tid%8)) of both matrices
  const int wrow = tid >> 3, wkq = tid & 7;
  f32x4 wa0, wa1, wb0, wb1;
  float bias_next = 0.f;
  // everything a chunk reads from global memory, fetched one chunk ahead
  auto fetch = [&](int c) {
    if constexpr (!BWD) bias_next = P.b1[c * HC + (tid & (HC - 1))];
    const float* pa = P.wa + (long)(c * HC + wrow) * D + wkq * 8;
    const float* pb = P.wb + (long)wrow * P.F + c * HC + wkq * 8;
    wa0 = *reinterpret_cast<const f32x4*>(pa); wa1 = *reinterpret_cast<const f32x4*>(pa + 4);
    wb0 = *reinterpret_cast<const f32x4*>(pb); wb1 = *reinterpret_cast<const f32x4*>(pb + 4);
  };
  auto stash = [&]() {
    if (!BWD && tid < HC) reinterpret_cast<float*>(lds + OFF_B1)[tid] = bias_next;
    uint4 hi, lo;
    const int off = sw_off(wrow, (wkq & 3) * 16);
    split8(wa0, wa1, hi, lo);
    *reinterpret_cast<uint4*>(lds + OFF_W1 + (wkq >> 2) * 2 * W_PLANE + off) = hi;
    *reinterpret_cast<uint4*>(lds + OFF_W1 + (wkq >> 2) * 2 * W_PLANE + W_PLANE + off) = lo;
    split8(wb0, wb1, hi, lo);
    *reinterpret_cast<uint4*>(lds + OFF_W2 + (wkq >> 2) * 2 * W_PLANE + off) = hi;
    *reinterpret_cast<uint4*>(lds + OFF_W2 + (wkq >> 2) * 2 * W_PLANE + W_PLANE + off) = lo;
  };
  fetch(0);

  const int token = row0 + w * 16 + r;                 // this lane's token in product 1's result
  const bool tok_ok = token < P.M;
  float* cs = reinterpret_cast<float*>(lds + OFF_CS);
  // mask words are read one chunk ahead (an HBM round trip is ~the length of a chunk)
  const unsigned long long* bits_row = (BWD && BITS) ? P.bits_in + (long)token * nchunks : nullptr;   // rows are padded to 128
  unsigned long long word_next = bits_row ? bits_row[0] : 0ull;
  float* mid_row = P.mid + (long)token * P.F + 4 * g;          // + c*HC + hb*16 per element group
  stash();                                             // chunk 0 (the x fragments were read before the barrier above)
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    // Next chunk's global reads, in flight during this chunk's arithmetic and waited for at the bottom of
    // the iteration.  The loop body has no data-dependent branch around a global load or store (the last chunk
    // re-fetches itself, rows past M land in the padding rows of the [tokens, F] buffers) and every load of
    // an iteration is issued before its stores: vmcnt counts loads and stores together and retires in order,
    // so the wait for these loads can leave exactly this iteration's stores outstanding instead of draining
    // them (an HBM round trip per chunk otherwise).
    const int cn = c + 1 < nchunks ? c + 1 : c;
    fetch(cn);
    const unsigned long long word = word_next;
    if constexpr (BWD && BITS) word_next = bits_row[cn];
    if (BWD && c > 0 && w == 0) {                      // column sums of the previous chunk: fold the 8 waves
      const float* prev = cs + ((c - 1) & 1) * 512;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += prev[q * 64 + lane];
      P.colpart[(long)blockIdx.x * P.F + (c - 1) * HC + lane] = s;
    }

    // ---- product 1, transposed: S^T[hidden 64][token 16 of this wave], contraction d = 64
    f32x4 acc1[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) acc1[hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        const int off = sw_off(hb * 16 + r, g * 16);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + OFF_W1 + ks * 2 * W_PLANE + off);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + OFF_W1 + ks * 2 * W_PLANE + W_PLANE + off);
        acc1[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[ks], acc1[hb], 0, 0, 0);
        acc1[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[ks], acc1[hb], 0, 0, 0);
        acc1[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[ks], acc1[hb], 0, 0, 0);
      }
    }

    // ---- epilogue of product 1: lane holds hidden units hb*16 + 4g .. +3 of chunk c for its token
    const long mid_off = (long)token * P.F + c * HC;
    unsigned nzbits = 0;                                // bit 4*hb + e
    f32x4 colv[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const int hcol = hb * 16 + 4 * g;
      f32x4 v = acc1[hb];
      if constexpr (!BWD) {
        v += *reinterpret_cast<const f32x4*>(lds + OFF_B1 + hcol * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if constexpr (DROP == 1) {
          if (tok_ok) v = v * *reinterpret_cast<const f32x4*>(P.keep + mid_off + hcol) * P.scale;
        } else if constexpr (DROP == 2) {
          const unsigned long long idx = (unsigned long long)mid_off + hcol;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vts_hash32(P.seed, idx + e) >= P.thresh ? v[e] * P.scale : 0.f;
        }
      } else {
        if constexpr (BITS) {
          const unsigned nib = (unsigned)(word >> (16 * g + 4 * hb)) & 15u;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((nib >> e) & 1u) ? v[e] * P.scale : 0.f;
        } else {
          f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (tok_ok) hv = *reinterpret_cast<const f32x4*>(P.hidden + mid_off + hcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (hv[e] != 0.f) ? v[e] * P.scale : 0.f;
        }
      }
      if (!tok_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(mid_row + c * HC + hb * 16));   // streamed: not read again by this kernel
      if constexpr (!BWD && BITS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) nzbits |= (v[e] != 0.f ? 1u : 0u) << (4 * hb + e);
      }
      colv[hb] = v;
      // second product's A operand: token row, k = hidden (hb>>1 picks the 32-wide k step)
      uint2 hi, lo;
      split4(v[0], v[1], v[2], v[3], hi, lo);
      const int off = sw_off(w * 16 + r, ((hb & 1) * 16 + 4 * g) * 2);
      *reinterpret_cast<uint2*>(lds + OFF_S + (hb >> 1) * 2 * S_PLANE + off) = hi;
      *reinterpret_cast<uint2*>(lds + OFF_S + (hb >> 1) * 2 * S_PLANE + S_PLANE + off) = lo;
    }
    if constexpr (!BWD) {
      if constexpr (BITS) {
        // mask word of (token, 64-unit block): 16-bit field g holds this lane's bits, bit 4*hb + e <-> unit
        // hb*16 + 4*g + e.  The four lanes of a token (g = 0..3) sit 16 lanes apart.
        // All four lanes of the token assemble and store the same word (an unconditional store, see above).
        const unsigned partner = __shfl_xor(nzbits, 16);
        const unsigned pair = (g & 1) ? (partner | (nzbits << 16)) : (nzbits | (partner << 16));   // fields (0,1) or (2,3)
        const unsigned other = __shfl_xor(pair, 32);
        const unsigned long long full = (g & 2) ? (((unsigned long long)pair << 32) | other) : (((unsigned long long)other << 32) | pair);
        P.bits_out[(long)token * nchunks + c] = full;
      }
    } else {
      // column sums over this wave's 16 tokens; lanes r == 0 publish 4 hidden units per block
#pragma unroll
      for (int hb = 0; hb < 4; ++hb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = colv[hb][e];
          s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
          if (r == 0) cs[(c & 1) * 512 + w * 64 + hb * 16 + 4 * g + e] = s;
        }
    }
    wave_lds_sync();                                   // S rows of this wave are written and read by this wave only

    // ---- product 2: out[token 16 of this wave][64], contraction over the chunk's 64 hidden units
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int offa = sw_off(w * 16 + r, g * 16);
      const bf16x8 sh = *reinterpret_cast<const bf16x8*>(lds + OFF_S + ks * 2 * S_PLANE + offa);
      const bf16x8 sl = *reinterpret_cast<const bf16x8*>(lds + OFF_S + ks * 2 * S_PLANE + S_PLANE + offa);
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const int off = sw_off(ob * 16 + r, g * 16);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds + OFF_W2 + ks * 2 * W_PLANE + off);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(lds + OFF_W2 + ks * 2 * W_PLANE + W_PLANE + off);
        acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, bl, acc2[ob], 0, 0, 0);
        acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sl, bh, acc2[ob], 0, 0, 0);
        acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, bh, acc2[ob], 0, 0, 0);
      }
    }
    __syncthreads();                                   // every wave is done with this chunk's weights (and its S rows)
    stash();                                           // next chunk's weights: waits for the fetch above only
    __syncthreads();
  }
  if constexpr (BWD) {
    if (w == 0) {
      const float* prev = cs + ((nchunks - 1) & 1) * 512;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += prev[q * 64 + lane];
      P.colpart[(long)blockIdx.x * P.F + (nchunks - 1) * HC + lane] = s;
    }
  }
  // ---- out tile: acc2[ob] = C[token 4g+reg][output ob*16 + r]
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    const int col = ob * 16 + r;
    const float bias = (!BWD && P.b2) ? P.b2[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + w * 16 + 4 * g + reg;
      if (row < P.M) P.out[(long)row * P.ldo + col] = acc2[ob][reg] + bias;
    }
  }
}

// out[c][r] = in[r][c]  (the backward kernel wants both weight matrices the other way round; 512 KiB each)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int rr = r0 + i, cc = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (rr < R && cc < C) ? in[(long)rr * C + cc] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int cc = c0 + i, rr = r0 + threadIdx.x;
    if (rr < R && cc < C) out[(long)cc * R + rr] = tile[threadIdx.x][i];
  }
}

template <bool BWD, int DROP, bool BITS>
int launch_b2b(const FfnParams& P, hipStream_t st) {
  static vts_lds_mask done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(ffn_b2b_kernel<BWD, DROP, BITS>), LDS_BYTES, &done, "ffn_b2b"))
    return rc;
  hipLaunchKernelGGL((ffn_b2b_kernel<BWD, DROP, BITS>), dim3((unsigned)((P.M + TOK - 1) / TOK)), dim3(512), LDS_BYTES, st, P);
  VTS_CHECK_LAUNCH("ffn_b2b");
  return VTS_OK;
}

}  // namespace

bool vts_ffn_fused_supported(long M, long d, long F, long ldx, const void* x, const void* keep) {
  return d == D && F % HC == 0 && F >= HC && (ldx % 4) == 0 && vts_aligned16(x) && (!keep || vts_aligned16(keep)) && M > 0;
}

int vts_ffn_fused_fwd(const float* x, long ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* keep, unsigned long long seed, unsigned thresh, float scale, float* hidden,
                      unsigned long long* bits, float* y, int M, int F, hipStream_t st) {
  FfnParams P{};
  P.x = x; P.ldx = ldx; P.wa = w1; P.wb = w2; P.b1 = b1; P.b2 = b2;
  P.keep = keep; P.seed = seed; P.thresh = thresh; P.scale = scale;
  P.mid = hidden; P.bits_out = bits; P.out = y; P.ldo = D; P.M = M; P.F = F;
  const int drop = keep ? 1 : (thresh ? 2 : 0);
  if (bits) {
    if (drop == 2) return launch_b2b<false, 2, true>(P, st);
    if (drop == 1) return launch_b2b<false, 1, true>(P, st);
    return launch_b2b<false, 0, true>(P, st);
  }
  if (drop == 2) return launch_b2b<false, 2, false>(P, st);
  if (drop == 1) return launch_b2b<false, 1, false>(P, st);
  return launch_b2b<false, 0, false>(P, st);
}

// w2t [F][64] and w1t [64][F] are scratch for the transposed weights; colpart [ceil(M/128)][F].
int vts_ffn_fused_bwd_dx(const float* dy, const float* w1, const float* w2, float* w1t, float* w2t,
                         const float* hidden, const unsigned long long* bits, float scale, float* dpre, float* dx,
                         long lddx, float* colpart, int M, int F, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((F + 31) / 32, (D + 31) / 32), dim3(32, 8), 0, st, w2, w2t, D, F);   // W2 [64][F] -> [F][64]
  VTS_CHECK_LAUNCH("transpose");
  hipLaunchKernelGGL(transpose_kernel, dim3((D + 31) / 32, (F + 31) / 32), dim3(32, 8), 0, st, w1, w1t, F, D);   // W1 [F][64] -> [64][F]
  VTS_CHECK_LAUNCH("transpose");
  FfnParams P{};
  P.x = dy; P.ldx = D; P.wa = w2t; P.wb = w1t;
  P.scale = scale; P.mid = dpre; P.hidden = hidden; P.bits_in = bits;
  P.out = dx; P.ldo = lddx; P.colpart = colpart; P.M = M; P.F = F;
  return bits ? launch_b2b<true, 0, true>(P, st) : launch_b2b<true, 0, false>(P, st);
}
