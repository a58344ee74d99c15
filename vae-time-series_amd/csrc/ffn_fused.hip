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
// the packed mask (backward without it compares the saved hidden activations with 0).  MID: the [tokens, F]
// intermediate (forward: hidden, backward: dpre) is written to memory; without it the weight gradients come from
// ffn_wgrad_recompute_kernel below, which works both intermediates out again from x, dy and the packed mask.
template <bool BWD, int DROP, bool BITS, bool MID = true>
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
          bool kp[4];
          vts_drop_keep4(P.seed, idx, P.thresh, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = kp[e] ? v[e] * P.scale : 0.f;
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
      if constexpr (MID)
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


// ---- the block without its [tokens, F] intermediate in memory: weights as a packed image, the chunk in registers ---
// Used when nothing but the packed mask is kept for backward.  Per chunk of 64 hidden units the two weight slices
// are read from a bf16 head/tail IMAGE that a small kernel writes once per call (the kernel above splits the fp32
// weights again in every workgroup and chunk) and that arrives in LDS by DMA, no registers or VALU involved:
//   WA [unit 64][k 64]     A operand of the first product, computed transposed: [unit][token] = WA . in^T
//   WB [col 64][unit 64]   B operand of the second: out[token][col] += mid[token][unit] . WB^T
//   b1 [unit 64] fp32      (forward)
// forward: in = x, WA = W1 rows, WB = W2 rows; backward: in = dy, WA = W2^T rows, WB = W1^T rows.
// The accumulator image of the transposed first product - lane (g, r): units 4g..4g+3 of block hb for token r - is,
// after the epilogue and the head/tail split, an A operand of the second product as it stands: row = token r,
// k-slots 0..3 = units 4g..4g+3 of block 2kk, 4..7 = those of block 2kk+1; the B operand is read from WB in the same
// unit order (two 8-byte pieces).  So the intermediate never touches LDS, and a wave (32 tokens, two per SIMD)
// runs 96 MFMAs per chunk against 16 + 32 fragment reads.  Column sums of the gradient (db1) are left to the
// weight-gradient kernel, which recomputes the same tile anyway.
constexpr int CP = 144;                        // row pitch of an image plane: 64 bf16 + 16 B (spreads the banks)
constexpr int C_PLANE = 64 * CP;
constexpr int C_WA = 0, C_WB = 2 * C_PLANE, C_B1 = 4 * C_PLANE;
constexpr int C_IMG = 4 * C_PLANE + 1024;      // 37 KiB per chunk: WA hi, WA lo, WB hi, WB lo, b1 (+ padding)
constexpr int C_LDS = 2 * C_IMG;
constexpr int C_TOK = 256;                     // tokens per workgroup: eight waves x 32

// image[c] for every chunk c; one thread per (chunk, WA | WB, row, pair of columns)
__global__ void ffn_pack_image_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b1,
                                      char* __restrict__ image, int F, int bwd) {
  const long total = (long)(F / HC) * 2 * 64 * 32;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cp = (int)(idx & 31), row = (int)((idx >> 5) & 63), which = (int)((idx >> 11) & 1);
    const long c = idx >> 12;
    float v0, v1;
    if (which == 0) {       // WA[row = unit][col = k]
      if (!bwd) { v0 = w1[(c * HC + row) * D + 2 * cp]; v1 = w1[(c * HC + row) * D + 2 * cp + 1]; }
      else { v0 = w2[(long)(2 * cp) * F + c * HC + row]; v1 = w2[(long)(2 * cp + 1) * F + c * HC + row]; }
    } else {                // WB[row = out column][col = unit]
      if (!bwd) { v0 = w2[(long)row * F + c * HC + 2 * cp]; v1 = w2[(long)row * F + c * HC + 2 * cp + 1]; }
      else { v0 = w1[(c * HC + 2 * cp) * D + row]; v1 = w1[(c * HC + 2 * cp + 1) * D + row]; }
    }
    unsigned hi, lo;
    split2(v0, v1, hi, lo);
    char* dst = image + c * C_IMG + (which ? C_WB : C_WA) + row * CP + cp * 4;
    *reinterpret_cast<unsigned*>(dst) = hi;
    *reinterpret_cast<unsigned*>(dst + C_PLANE) = lo;
    if (which == 0 && cp == 0) reinterpret_cast<float*>(image + c * C_IMG + C_B1)[row] = b1 ? b1[c * HC + row] : 0.f;
  }
}

struct FfnRegsParams {
  const float* x; long ldx;        // forward: layer input; backward: dy
  const char* image;               // [F / 64][C_IMG]
  const float* b2;                 // forward
  const float* keep;               // forward, DROP 1: [M][F] float 0/1
  unsigned long long seed;
  unsigned thresh;
  float scale;
  unsigned long long* bits_out;    // forward:  [rows][F / 64]
  const unsigned long long* bits_in;   // backward
  float* out; long ldo;            // y / dx
  int M, F;
};

__device__ __forceinline__ unsigned ffn_lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

template <bool BWD, int DROP>
__global__ __launch_bounds__(512) void ffn_regs_kernel(const FfnRegsParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * C_TOK + w * 32;
  const int nchunks = P.F / HC;

  // ---- the wave's 32 tokens as B fragments of the first product: lane (g, r) = token r of tile T, k = 8g.. of k-step ks
  uint4 xh[2][2], xl[2][2];
  bool ok[2];
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int token = row0 + 16 * T + r;
    ok[T] = token < P.M;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* src = P.x + (long)token * P.ldx + 32 * ks + 8 * g;
      const f32x4 a = ok[T] ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 b = ok[T] ? *reinterpret_cast<const f32x4*>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      split8(a, b, xh[T][ks], xl[T][ks]);
    }
  }
  // ---- chunk images: 37 one-KiB pieces, piece q by wave q % 8, straight into LDS
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(ffn_lds_addr(lds));
  auto issue = [&](int c, int buf) {
    const char* src = P.image + (long)c * C_IMG;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int q = w + 8 * i;
      if (q < C_IMG / 1024) {
        const unsigned dst = lds0 + buf * C_IMG + q * 1024;
        const unsigned vo = lane * 16 + q * 1024;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(vo), "s"(src) : "memory", "m0");
      }
    }
  };
  issue(0, 0);
  f32x4 acc2[2][4];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc2[T][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long* bits_row[2] = {nullptr, nullptr};
  unsigned long long word_next[2] = {0ull, 0ull};
  if constexpr (BWD) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      bits_row[T] = P.bits_in + (long)(row0 + 16 * T + r) * nchunks;
      word_next[T] = ok[T] ? bits_row[T][0] : 0ull;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const char* img = lds + (c & 1) * C_IMG;
    if (c + 1 < nchunks) issue(c + 1, (c & 1) ^ 1);      // the other buffer: every wave left it a barrier ago
    unsigned long long word[2] = {word_next[0], word_next[1]};
    if constexpr (BWD) {
      const int cn = c + 1 < nchunks ? c + 1 : c;
#pragma unroll
      for (int T = 0; T < 2; ++T) word_next[T] = ok[T] ? bits_row[T][cn] : 0ull;
    }
    // ---- first product, transposed: [unit 64][token 16] per tile
    f32x4 a1[2][4];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) a1[T][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        const char* pa = img + C_WA + (16 * hb + r) * CP + 64 * ks + 16 * g;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(pa), al = *reinterpret_cast<const bf16x8*>(pa + C_PLANE);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          a1[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, xl[T][ks]), a1[T][hb], 0, 0, 0);
          a1[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, xh[T][ks]), a1[T][hb], 0, 0, 0);
          a1[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, xh[T][ks]), a1[T][hb], 0, 0, 0);
        }
      }
    // ---- epilogue in registers: lane (g, r) holds units 16 hb + 4g .. + 3 of chunk c for token r of tile T
    uint4 sh[2][2], sl[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const long token = row0 + 16 * T + r;
      unsigned nz = 0;
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        f32x4 v = a1[T][hb];
        if constexpr (!BWD) {
          v += *reinterpret_cast<const f32x4*>(img + C_B1 + (16 * hb + 4 * g) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          const long idx = token * P.F + (long)c * HC + 16 * hb + 4 * g;
          if constexpr (DROP == 1) {
            if (ok[T]) v = v * *reinterpret_cast<const f32x4*>(P.keep + idx) * P.scale;
          } else if constexpr (DROP == 2) {
            bool kp[4];
            vts_drop_keep4(P.seed, (unsigned long long)idx, P.thresh, kp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = kp[e] ? v[e] * P.scale : 0.f;
          }
          if (!ok[T]) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) nz |= (v[e] != 0.f ? 1u : 0u) << (4 * hb + e);
        } else {
          const unsigned nib = (unsigned)(word[T] >> (16 * g + 4 * hb)) & 15u;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((nib >> e) & 1u) ? v[e] * P.scale : 0.f;
        }
        a1[T][hb] = v;
      }
      split8(a1[T][0], a1[T][1], sh[T][0], sl[T][0]);
      split8(a1[T][2], a1[T][3], sh[T][1], sl[T][1]);
      if constexpr (!BWD) {
        // mask word of (token, chunk): 16-bit field g holds this lane's bits (bit 4 hb + e <-> unit 16 hb + 4 g + e);
        // the four lanes of a token sit 16 lanes apart
        const unsigned partner = __shfl_xor(nz, 16);
        const unsigned pair = (g & 1) ? (partner | (nz << 16)) : (nz | (partner << 16));
        const unsigned other = __shfl_xor(pair, 32);
        const unsigned long long full = (g & 2) ? (((unsigned long long)pair << 32) | other) : (((unsigned long long)other << 32) | pair);
        if (g == 0 && ok[T]) P.bits_out[token * nchunks + c] = full;
      }
    }
    // ---- second product: out[token][col] += mid[token][unit] . WB[col][unit]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const char* pb = img + C_WB + (16 * ob + r) * CP + 64 * kk + 8 * g;
        auto frag = [&](const char* q) -> bf16x8 {
          const uint2 lo = *reinterpret_cast<const uint2*>(q), hi = *reinterpret_cast<const uint2*>(q + 32);
          return __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
        };
        const bf16x8 bh = frag(pb), bl = frag(pb + C_PLANE);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          acc2[T][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sh[T][kk]), bl, acc2[T][ob], 0, 0, 0);
          acc2[T][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sl[T][kk]), bh, acc2[T][ob], 0, 0, 0);
          acc2[T][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sh[T][kk]), bh, acc2[T][ob], 0, 0, 0);
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the next image have landed
    __syncthreads();                                    // ... and everybody's; everybody is done with this one
  }
  // ---- out tile: acc2[T][ob] = C[token 16 T + 4g + reg][col 16 ob + r]
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const int col = ob * 16 + r;
      const float bias = (!BWD && P.b2) ? P.b2[col] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = row0 + 16 * T + 4 * g + reg;
        if (row < P.M) P.out[(long)row * P.ldo + col] = acc2[T][ob][reg] + bias;
      }
    }
}

// ---- weight gradients of the block without the [tokens, F] intermediates in memory -----------------------------
//   dW2[o][u] = sum_t dy[t][o] * hidden[t][u],   dW1[u][k] = sum_t dpre[t][u] * x[t][k]
// with hidden = bit ? (x . W1^T + b1) * scale : 0 and dpre = bit ? (dy . W2) * scale : 0 worked out again per tile
// (bit = the packed relu/dropout mask the forward pass wrote): four products per (token tile, 64 hidden units)
// instead of two, against 3 x 4 bytes per hidden element that no longer cross HBM.
//
// A workgroup (eight waves, two per SIMD: one's mask / split arithmetic runs under the other's MFMAs) owns a slab of
// tokens and four chunks of 64 hidden units; two waves share a chunk, one recomputes hidden and accumulates dW2,
// the other recomputes dpre and accumulates dW1 (as its transpose, which makes the two waves' code the same up to
// which planes and which image they read).  A wave keeps its matrix's fragments (B operand of its recomputed
// product) and the sixteen accumulator tiles of its 64 x 64 gradient block in registers for the whole slab.  Tokens come in
// tiles of 64: x and dy are split into bf16 heads and tails once per tile and written to LDS twice - token-major
// (A operand of the recomputed products, one ds_read_b128 per fragment) and feature-major (operands of the two
// gradient products, which contract over tokens: 8-byte reads).  The recomputed tiles never leave registers: the
// accumulator image of a [token][unit] product (lane (g, r): tokens 4g..4g+3, unit r) IS an MFMA operand whose
// contraction index is the token, so hidden goes into dW2's product as the B operand and dpre into dW1's as the A
// operand straight from the registers they were computed in.  A k-step of 32 tokens takes the two 16-token tiles
// T0, T1 of an iteration: k-slot 8g+e is token 4g+e of T0 (e < 4) or 4g+e-4 of T1; the operand read from LDS uses
// the same order, so the sum does not notice.  Slabs are summed by ffn_slab_reduce_kernel (fixed order).
constexpr int RT = 64;                     // tokens per staged tile
constexpr int RP = 144;                    // row pitch in bytes of a token-major plane row (64 bf16 + 16 B: spreads the banks)
constexpr int RPT = 136;                   // ... of a feature-major one (8-byte accesses only; 144 would put the staging
                                           // writes of rows four apart on the same banks, four deep)
constexpr int R_PLANE = 64 * RP;           // one plane: 64 rows
constexpr int R_XA = 0;                    // token-major planes: x hi, x lo, dy hi, dy lo
constexpr int R_XT = 4 * R_PLANE;          // feature-major planes, same order
constexpr int R_BITS = 8 * R_PLANE;        // [token 64][chunk of the group 4] mask words
constexpr int R_STAGE = R_BITS + RT * 32;  // 75 776 B
constexpr int R_LDS = 2 * R_STAGE;

struct FfnWgradParams {
  const float* x; long ldx;        // [M][ldx]
  const float* dy;                 // [M][64]
  const char* image_fwd;           // packed images of the forward and the backward chain kernels: their WA planes
  const char* image_bwd;           // are W1's and W2^T's rows of a chunk as bf16 heads / tails, b1 sits in the forward one
  const unsigned long long* bits;  // [rows padded to 128][F / 64]
  float scale;
  float* slab1;                    // [nslabs][F][64] partial dW1
  float* slab2;                    // [nslabs][64][F] partial dW2
  float* colpart;                  // [nslabs][F] partial column sums of dpre (db1)
  int M, F, nslabs, tiles_per_slab;
};

__global__ __launch_bounds__(512) void ffn_wgrad_recompute_kernel(const FfnWgradParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int slab = blockIdx.x % P.nslabs, cg = blockIdx.x / P.nslabs;   // a slab's chunk groups sit nslabs apart: one XCD when nslabs % 8 == 0
  const int nchunks = P.F / HC;
  const int c = cg * 4 + (w >> 1);                                      // this wave's chunk ...
  const int path = w & 1;                                               // ... and its half of the work (see above)
  const int tile0 = slab * P.tiles_per_slab;
  const int ntiles_all = (P.M + RT - 1) / RT;
  int ntiles = ntiles_all - tile0;
  if (ntiles > P.tiles_per_slab) ntiles = P.tiles_per_slab;
  if (ntiles < 0) ntiles = 0;
  // planes of the staged tile this wave reads: token-major operand of the recomputed product, feature-major operand
  // of the gradient product (the other tensor)
  const int pl_rec = path ? 2 * R_PLANE : 0, pl_grad = path ? 0 : 2 * R_PLANE;

  // ---- this wave's weight fragments: B operands, lane (g, r) = unit r of block hb, k = 8g.. of k-step ks
  uint4 wh[4][2], wl[4][2];
  float bias[4];
  {
    const char* img = (path ? P.image_bwd : P.image_fwd) + (long)c * C_IMG;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      bias[hb] = path ? 0.f : reinterpret_cast<const float*>(img + C_B1)[hb * 16 + r];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = C_WA + (16 * hb + r) * CP + 64 * ks + 16 * g;
        wh[hb][ks] = *reinterpret_cast<const uint4*>(img + off);
        wl[hb][ks] = *reinterpret_cast<const uint4*>(img + off + C_PLANE);
      }
    }
  }
  f32x4 G[4][4];                     // gradient block [feature tile j][unit block hb]: dW2[o][unit] or dW1[unit][k] transposed
  float cs[4] = {0.f, 0.f, 0.f, 0.f};      // column sums of the recomputed tile over this lane's tokens, per unit block
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) G[j][hb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging: thread = 4 tokens x 4 features of x (threads 0..255) or of dy; 128 threads also carry two mask words each
  const int tg = (tid >> 4) & 15, dg = tid & 15;
  const bool is_dy = tid >= 256;
  const float* sbase = is_dy ? P.dy : P.x;
  const long sld = is_dy ? D : P.ldx;
  f32x4 vx[4];
  uint4 vb = uint4{0u, 0u, 0u, 0u};
  auto fetch = [&](int tile) {
    const long tok = (long)tile * RT + 4 * tg;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      vx[i] = tok + i < P.M ? *reinterpret_cast<const f32x4*>(sbase + (tok + i) * sld + 4 * dg) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 128) {     // token tid >> 1, words 2 (tid & 1), +1 of the group's four (the mask rows are padded to 128 tokens)
      const long t = (long)tile * RT + (tid >> 1);
      vb = *reinterpret_cast<const uint4*>(P.bits + t * nchunks + cg * 4 + 2 * (tid & 1));
    }
  };
  auto stash = [&](int buf) {
    char* st = lds + buf * R_STAGE;
    char* pa = st + R_XA + (is_dy ? 2 * R_PLANE : 0);      // the hi plane of the pair; lo follows
    char* pt = st + R_XT + (is_dy ? 2 * R_PLANE : 0);
    uint2 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split4(vx[i][0], vx[i][1], vx[i][2], vx[i][3], hi[i], lo[i]);
      *reinterpret_cast<uint2*>(pa + (4 * tg + i) * RP + 8 * dg) = hi[i];
      *reinterpret_cast<uint2*>(pa + R_PLANE + (4 * tg + i) * RP + 8 * dg) = lo[i];
    }
    // 4 x 4 transpose of 16-bit elements: feature 4 dg + q of tokens 4 tg .. 4 tg + 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      auto col = [&](const uint2* m) -> uint2 {
        const unsigned a0 = q < 2 ? m[0].x : m[0].y, a1 = q < 2 ? m[1].x : m[1].y;
        const unsigned a2 = q < 2 ? m[2].x : m[2].y, a3 = q < 2 ? m[3].x : m[3].y;
        if (q & 1) return uint2{(a0 >> 16) | (a1 & 0xffff0000u), (a2 >> 16) | (a3 & 0xffff0000u)};
        return uint2{(a0 & 0xffffu) | (a1 << 16), (a2 & 0xffffu) | (a3 << 16)};
      };
      *reinterpret_cast<uint2*>(pt + (4 * dg + q) * RPT + 8 * tg) = col(hi);
      *reinterpret_cast<uint2*>(pt + R_PLANE + (4 * dg + q) * RPT + 8 * tg) = col(lo);
    }
    if (tid < 128) *reinterpret_cast<uint4*>(st + R_BITS + (tid >> 1) * 32 + (tid & 1) * 16) = vb;
  };

  if (ntiles > 0) {
    fetch(tile0);
    stash(0);
  }
  __syncthreads();
  const int bitpos = 16 * (r >> 2) + (r & 3);       // + 4 hb: this lane's unit inside a mask word
  for (int ti = 0; ti < ntiles; ++ti) {
    const int buf = ti & 1;
    const char* st = lds + buf * R_STAGE;
    if (ti + 1 < ntiles) fetch(tile0 + ti + 1);      // in flight during this tile's products
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      // ---- the recomputed product: [token][unit] tiles of this wave's chunk for tokens 32 it .. 32 it + 31
      f32x4 a[2][4];
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) a[T][hb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          const char* row = st + R_XA + pl_rec + (32 * it + 16 * T + r) * RP + 64 * ks + 16 * g;
          const bf16x8 xh = *reinterpret_cast<const bf16x8*>(row), xl = *reinterpret_cast<const bf16x8*>(row + R_PLANE);
#pragma unroll
          for (int hb = 0; hb < 4; ++hb)
            a[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, __builtin_bit_cast(bf16x8, wl[hb][ks]), a[T][hb], 0, 0, 0);
#pragma unroll
          for (int hb = 0; hb < 4; ++hb)
            a[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, __builtin_bit_cast(bf16x8, wh[hb][ks]), a[T][hb], 0, 0, 0);
#pragma unroll
          for (int hb = 0; hb < 4; ++hb)
            a[T][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, __builtin_bit_cast(bf16x8, wh[hb][ks]), a[T][hb], 0, 0, 0);
        }
      // ---- mask and bias (the dropout scale is applied to the finished sums); then head / tail fragments with the
      // token as contraction index
      unsigned wlo[2][4];      // this lane's units of token (T, e): bit 4 hb
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          wlo[T][e] = (unsigned)(*reinterpret_cast<const unsigned long long*>(st + R_BITS + (32 * it + 16 * T + 4 * g + e) * 32 + (w >> 1) * 8) >> bitpos);
      uint4 fh[4], fl[4];
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[T][hb][e] = (wlo[T][e] >> (4 * hb)) & 1u ? a[T][hb][e] + bias[hb] : 0.f;
            cs[hb] += a[T][hb][e];
          }
        split8(a[0][hb], a[1][hb], fh[hb], fl[hb]);
      }
      // ---- the gradient product: feature-major fragments of the other tensor as A, the recomputed tile as B
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* row = st + R_XT + pl_grad + (16 * j + r) * RPT + 64 * it + 8 * g;
        auto frag = [&](int plane) -> bf16x8 {
          const uint2 lo = *reinterpret_cast<const uint2*>(row + plane * R_PLANE);
          const uint2 hi = *reinterpret_cast<const uint2*>(row + plane * R_PLANE + 32);
          return __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
        };
        const bf16x8 th = frag(0), tl = frag(1);
#pragma unroll
        for (int hb = 0; hb < 4; ++hb)
          G[j][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, __builtin_bit_cast(bf16x8, fl[hb]), G[j][hb], 0, 0, 0);
#pragma unroll
        for (int hb = 0; hb < 4; ++hb)
          G[j][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl, __builtin_bit_cast(bf16x8, fh[hb]), G[j][hb], 0, 0, 0);
#pragma unroll
        for (int hb = 0; hb < 4; ++hb)
          G[j][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, __builtin_bit_cast(bf16x8, fh[hb]), G[j][hb], 0, 0, 0);
      }
    }
    if (ti + 1 < ntiles) stash(buf ^ 1);             // the other buffer: every wave left it a barrier ago
    __syncthreads();
  }
  // ---- this slab's partial blocks: lane (g, r) holds features 16 j + 4g .. + 3 of unit 16 hb + r
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) G[j][hb] *= P.scale;
  if (path == 0) {
    float* s2 = P.slab2 + (size_t)slab * D * P.F;            // dW2[o][unit]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb)
#pragma unroll
        for (int e = 0; e < 4; ++e) s2[(size_t)(16 * j + 4 * g + e) * P.F + (size_t)c * HC + hb * 16 + r] = G[j][hb][e];
  } else {
    float* s1 = P.slab1 + (size_t)slab * P.F * D;            // dW1[unit][k]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb)
        *reinterpret_cast<f32x4*>(s1 + ((size_t)c * HC + hb * 16 + r) * D + 16 * j + 4 * g) = G[j][hb];
    // column sums of dpre: the four lane groups of a unit hold disjoint tokens
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      float v = cs[hb];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (g == 0) P.colpart[(size_t)slab * P.F + (size_t)c * HC + hb * 16 + r] = v * P.scale;
    }
  }
}

// out[i] = sum over slabs of part[s][i], in slab order
__global__ void ffn_slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n4, int nslabs) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
    for (int k = 1; k < nslabs; ++k) s += reinterpret_cast<const f32x4*>(part)[(long)k * n4 + i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

template <bool BWD, int DROP, bool BITS, bool MID = true>
int launch_b2b(const FfnParams& P, hipStream_t st) {
  static vts_lds_mask done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(ffn_b2b_kernel<BWD, DROP, BITS, MID>), LDS_BYTES, &done, "ffn_b2b"))
    return rc;
  hipLaunchKernelGGL((ffn_b2b_kernel<BWD, DROP, BITS, MID>), dim3((unsigned)((P.M + TOK - 1) / TOK)), dim3(512), LDS_BYTES, st, P);
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
  if (!hidden) {      // nothing but the packed mask is kept for backward (vts_ffn_bwd_recompute)
    if (!bits) { vts_set_error("ffn_fwd: without a hidden buffer the packed mask is required"); return VTS_ERR_ARG; }
    if (drop == 2) return launch_b2b<false, 2, true, false>(P, st);
    if (drop == 1) return launch_b2b<false, 1, true, false>(P, st);
    return launch_b2b<false, 0, true, false>(P, st);
  }
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
  if (!dpre) {
    if (!bits) { vts_set_error("ffn_bwd: without a dpre buffer the packed mask is required"); return VTS_ERR_ARG; }
    return launch_b2b<true, 0, true, false>(P, st);
  }
  return bits ? launch_b2b<true, 0, true>(P, st) : launch_b2b<true, 0, false>(P, st);
}

// Weight gradients by recomputation (see ffn_wgrad_recompute_kernel).  slabs: vts_ffn_wgrad_recompute_slab_floats
// floats of scratch; colpart: [vts_ffn_wgrad_recompute_nslabs][F] floats, partial column sums of dpre; the images are
// those of vts_ffn_regs_fwd / vts_ffn_regs_bwd_dx.
int vts_ffn_wgrad_recompute_nslabs(long M, long F) {
  const long groups = F / (4 * HC), tiles = (M + RT - 1) / RT;
  long n = groups > 0 ? 256 / groups : 1;       // about one workgroup per compute unit
  if (n > 64) n = 64;
  if (n > tiles) n = tiles;
  return (int)(n < 1 ? 1 : n);
}
size_t vts_ffn_wgrad_recompute_slab_floats(long M, long F) { return (size_t)2 * vts_ffn_wgrad_recompute_nslabs(M, F) * F * D; }
bool vts_ffn_wgrad_recompute_supported(long M, long d, long F, long ldx, const void* x) {
  return d == D && F % (4 * HC) == 0 && F >= 4 * HC && (ldx % 4) == 0 && vts_aligned16(x) && M > 0;
}
int vts_ffn_wgrad_recompute(const float* x, long ldx, const float* dy, const void* image_fwd, const void* image_bwd,
                            const unsigned long long* bits, float scale, float* slabs, float* colpart, float* dw1, float* dw2,
                            int M, int F, hipStream_t st) {
  FfnWgradParams P{};
  P.x = x; P.ldx = ldx; P.dy = dy; P.bits = bits; P.scale = scale; P.colpart = colpart;
  P.image_fwd = reinterpret_cast<const char*>(image_fwd); P.image_bwd = reinterpret_cast<const char*>(image_bwd);
  P.M = M; P.F = F;
  P.nslabs = vts_ffn_wgrad_recompute_nslabs(M, F);
  const int tiles = (M + RT - 1) / RT;
  P.tiles_per_slab = (tiles + P.nslabs - 1) / P.nslabs;
  P.slab1 = slabs;
  P.slab2 = slabs + (size_t)P.nslabs * F * D;
  static vts_lds_mask done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(ffn_wgrad_recompute_kernel), R_LDS, &done, "ffn_wgrad_recompute"))
    return rc;
  hipLaunchKernelGGL(ffn_wgrad_recompute_kernel, dim3((unsigned)(P.nslabs * (F / (4 * HC)))), dim3(512), R_LDS, st, P);
  VTS_CHECK_LAUNCH("ffn_wgrad_recompute");
  const long n4 = (long)F * D / 4;
  hipLaunchKernelGGL(ffn_slab_reduce_kernel, dim3(128), dim3(256), 0, st, P.slab1, dw1, n4, P.nslabs);
  VTS_CHECK_LAUNCH("ffn_slab_reduce");
  hipLaunchKernelGGL(ffn_slab_reduce_kernel, dim3(128), dim3(256), 0, st, P.slab2, dw2, n4, P.nslabs);
  VTS_CHECK_LAUNCH("ffn_slab_reduce");
  return VTS_OK;
}

// The block through the register-chain kernel (ffn_regs_kernel).  image: vts_ffn_image_bytes(F) of scratch that the
// call fills (forward: kept by the caller for the weight-gradient kernel).
size_t vts_ffn_image_bytes(long F) { return (size_t)(F / HC) * C_IMG; }
template <bool BWD, int DROP>
static int launch_regs(const FfnRegsParams& P, hipStream_t st) {
  static vts_lds_mask done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(ffn_regs_kernel<BWD, DROP>), C_LDS, &done, "ffn_regs")) return rc;
  hipLaunchKernelGGL((ffn_regs_kernel<BWD, DROP>), dim3((unsigned)((P.M + C_TOK - 1) / C_TOK)), dim3(512), C_LDS, st, P);
  VTS_CHECK_LAUNCH("ffn_regs");
  return VTS_OK;
}
static int pack_image(const float* w1, const float* w2, const float* b1, void* image, int F, int bwd, hipStream_t st) {
  hipLaunchKernelGGL(ffn_pack_image_kernel, dim3((unsigned)((F / HC) * 16)), dim3(256), 0, st, w1, w2, b1, reinterpret_cast<char*>(image), F, bwd);
  VTS_CHECK_LAUNCH("ffn_pack_image");
  return VTS_OK;
}
int vts_ffn_pack_fwd_image(const float* w1, const float* b1, const float* w2, void* image, int F, hipStream_t st) {
  return pack_image(w1, w2, b1, image, F, 0, st);
}
int vts_ffn_regs_fwd(const float* x, long ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* keep, unsigned long long seed, unsigned thresh, float scale, unsigned long long* bits,
                     void* image, float* y, int M, int F, hipStream_t st) {
  if (int rc = pack_image(w1, w2, b1, image, F, 0, st)) return rc;
  FfnRegsParams P{};
  P.x = x; P.ldx = ldx; P.image = reinterpret_cast<const char*>(image); P.b2 = b2; P.keep = keep; P.seed = seed;
  P.thresh = thresh; P.scale = scale; P.bits_out = bits; P.out = y; P.ldo = D; P.M = M; P.F = F;
  const int drop = keep ? 1 : (thresh ? 2 : 0);
  if (drop == 2) return launch_regs<false, 2>(P, st);
  if (drop == 1) return launch_regs<false, 1>(P, st);
  return launch_regs<false, 0>(P, st);
}
int vts_ffn_regs_bwd_dx(const float* dy, const float* w1, const float* w2, const unsigned long long* bits, float scale,
                        void* image, float* dx, long lddx, int M, int F, hipStream_t st) {
  if (int rc = pack_image(w1, w2, nullptr, image, F, 1, st)) return rc;
  FfnRegsParams P{};
  P.x = dy; P.ldx = D; P.image = reinterpret_cast<const char*>(image); P.scale = scale; P.bits_in = bits;
  P.out = dx; P.ldo = lddx; P.M = M; P.F = F;
  return launch_regs<true, 0>(P, st);
}
