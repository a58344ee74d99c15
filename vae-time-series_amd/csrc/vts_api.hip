// extern "C" surface of libvts_hip.so (declared in include/vts.h): argument checks, workspace
// carving and kernel sequencing.  No device allocation, no synchronisation.
#include "vts_common.h"
#include <atomic>
#include "../../include/vts.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

// ---- internal entry points (gemm_f32.hip, lstm_rec_f32.hip, elementwise.hip) -------------------
int vts_ew_adam_counted(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        long long* state, float gscale, hipStream_t st);
int vts_ew_adam_guarded(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float bc1, float bc2_sqrt, float gscale, const int* guard, int n_guard, hipStream_t st);
int vts_ew_elbo_sums(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                     const float* scale, const float* shape, const float* klpart, int nkl, float thr, long B, int fs,
                     int L, float* part, float* sums, hipStream_t st);
int vts_ew_elbo_from_sums(int kind, const float* sums, const float* pi, float beta, float gscale, int fs, int L,
                          float* out10, hipStream_t st);
int vts_gemm_pick_ksplit(int M, int N, long K, int mode);
size_t vts_gemm_slab_bytes(int M, int N, int ksplit);
int vts_gemm_inproj(const float* X, const float* W, const float* b0, const float* b1, float* G, int Bwin, int T,
                    int Cin, int N4H, int mode, hipStream_t stream);
int vts_gemm_dgrad_t16(const float* dG, const float* W, float* dX, int Bwin, int T, int Cin, int N4H,
                       int accumulate, int mode, hipStream_t stream);
int vts_gemm_wgrad_t16(const float* dG, const float* X, float* dW, float* slabs, int ksplit, int Bwin, int T,
                       int F, long ldx, int N4H, int tshift, int accumulate, int mode, hipStream_t stream);
int vts_gemm_wgrad_cut(int M, int N, long K, int mode);
bool vts_gemm_wgrad_pair_usable(int N4H, int F, int Hh, long K, int mode);
size_t vts_gemm_wgrad_pair_slab_bytes(int N4H, int F, int Hh, long K, int mode);
int vts_gemm_wgrad_pair_t16(const float* dG, const float* X, long ldx, const float* Hs, float* dWih, float* dWhh,
                            float* slabs, int Bwin, int T, int F, int Hh, int N4H, int accumulate, int mode,
                            hipStream_t stream);
int vts_gemm_linear_fwd(const float* X, long ldx, const float* W, const float* bias, float* Y, long ldy, int M,
                        int N, int K, int act, const float* keep, float keep_scale, unsigned long long drop_seed,
                        unsigned drop_thresh, float* aux, unsigned long long* bits_out, int mode,
                        hipStream_t stream);
int vts_gemm_linear_dx(const float* dY, long lddy, const float* W, float* dX, long lddx, int M, int N, int K,
                       int accumulate, const float* mask_src, long ld_mask, const unsigned long long* mask_bits,
                       float mask_scale, float* colpart, int mode, hipStream_t stream);
int vts_gemm_linear_dw(const float* dY, long lddy, const float* X, long ldx, float* dW, float* slabs, int ksplit,
                       int M, int N, int K, int accumulate, int mode, hipStream_t stream);
bool vts_ffn_fused_supported(long M, long d, long F, long ldx, const void* x, const void* keep);
int vts_ffn_fused_fwd(const float* x, long ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* keep, unsigned long long seed, unsigned thresh, float scale, float* hidden,
                      unsigned long long* bits, float* y, int M, int F, hipStream_t st);
int vts_ffn_fused_bwd_dx(const float* dy, const float* w1, const float* w2, float* w1t, float* w2t,
                         const float* hidden, const unsigned long long* bits, float scale, float* dpre, float* dx,
                         long lddx, float* colpart, int M, int F, hipStream_t st);
int vts_ffn_wgrad_recompute_nslabs(long M, long F);
size_t vts_ffn_wgrad_recompute_slab_floats(long M, long F);
bool vts_ffn_wgrad_recompute_supported(long M, long d, long F, long ldx, const void* x);
int vts_ffn_wgrad_recompute(const float* x, long ldx, const float* dy, const void* image_fwd, const void* image_bwd,
                            const unsigned long long* bits, float scale, float* slabs, float* colpart, float* dw1, float* dw2,
                            int M, int F, hipStream_t st);
size_t vts_ffn_image_bytes(long F);
int vts_ffn_pack_fwd_image(const float* w1, const float* b1, const float* w2, void* image, int F, hipStream_t st);
int vts_ffn_regs_fwd(const float* x, long ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* keep, unsigned long long seed, unsigned thresh, float scale, unsigned long long* bits,
                     void* image, float* y, int M, int F, hipStream_t st);
int vts_ffn_regs_bwd_dx(const float* dy, const float* w1, const float* w2, const unsigned long long* bits, float scale,
                        void* image, float* dx, long lddx, int M, int F, hipStream_t st);
int vts_attn_mha(int bwd, const float* q, long ldq, const float* k, const float* v, long ldkv, float* out, long ldo,
                 float* lse, const unsigned char* keep, float drop_p, unsigned long long seed, const float* dout,
                 long lddo, float* dq, long lddq, float* dk, float* dv, long lddkv, int B, int T, int Tq, int nhead,
                 int dh, hipStream_t st);
int vts_attn_dropout_add(const float* residual, const float* x, const unsigned char* keep, float drop_p,
                         unsigned long long seed, float* out, long n, hipStream_t st);
int vts_attn_dropout_mask(unsigned long long seed, float drop_p, unsigned char* keep, long n, hipStream_t st);
bool vts_lstm_hidden_supported(int H);
struct vts_mlp_chain_desc;
int vts_mlp_chain_supported_impl(const vts_mlp_chain_desc* d, long B);
int vts_mlp_chain_fwd_impl(const vts_mlp_chain_desc* d, const float* x, long ldx, long B, const float* const* w,
                           const float* const* b, const float* const* keep, float* const* trunk_out,
                           float* const* head_out, float* const* head_aux, hipStream_t st);
size_t vts_mlp_chain_bwd_workspace_bytes_impl(const vts_mlp_chain_desc* d, long B);
int vts_mlp_chain_bwd_impl(const vts_mlp_chain_desc* d, const float* x, long ldx, long B, const float* const* w,
                           const float* const* keep, float* const* trunk_out, float* const* head_out,
                           float* const* head_aux, const float* const* d_head, float* dx, long lddx, float* const* dw,
                           float* const* db, void* workspace, size_t workspace_bytes, hipStream_t st);
bool vts_lstm_rec3_supported(int H);
void vts_lstm_rec3_set_staged_bwd(int on);
bool vts_lstm_rec3c_usable(int B, int T, int H);
bool vts_planes_usable(long B, long T, long H);
int vts_planes_fpad(long F);
size_t vts_planes_seq_bytes(long B, long T, long F, int with_zero_slot);
size_t vts_planes_weight_bytes(long rows, long K);
size_t vts_planes_wgrad_slab_bytes(long B, long T, long F, long H);
int vts_planes_from_rows(const float* x, void* planes, long B, long T, long C, hipStream_t st);
int vts_planes_zero_slot0(void* planes, long B, long T, long F, hipStream_t st);
int vts_planes_inproj(const void* s_planes, int a_zero_slot, const float* w_ih, const float* b_ih, const float* b_hh,
                      float* gates, void* wplanes, long B, long T, long Cin, long H, hipStream_t st);
int vts_planes_dgrad(const void* dg_planes, const float* w_ih, float* dx_t16, void* wplanes, long B, long T, long Cin,
                     long H, hipStream_t st);
int vts_planes_wgrad(const void* dg_planes, const void* s_planes, int x_zero_slot, const void* h_planes, float* dw_ih,
                     float* dw_hh, int accumulate, float* slabs, long B, long T, long F, long H, hipStream_t st);
int vts_lstm_rec3c_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp,
                       void* xws, void* fws, unsigned* err, int B, int T, void* h_planes, hipStream_t stream);
size_t vts_lstm_rec3c_bwd_flag_bytes(int B);
bool vts_lstm_fused_usable(long B, long T, long Cin, long H);
size_t vts_lstm_fused_image_bytes(long Cin);
size_t vts_lstm_fused_flag_bytes(long B);
void vts_lstm_fused_set_fault(int member);
int vts_lstm_fused_fwd(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih, const float* b_hh,
                       const float* w_hh, float* G, float* cseq, float* hseq, void* h_planes, float* hlast, float* clast,
                       void* image, void* flags, unsigned* err, long B, long T, long Cin, hipStream_t stream);
bool vts_lstm_rec3p_usable(int B, int T, int H);
size_t vts_lstm_rec3p_xchg_bytes(int B);
void vts_lstm_rec3p_set_fault(int member);
int vts_lstm_rec3p_bwd(const float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                       float* dbpart, const float* W_hh, float* Wp, void* dg_planes, void* xws, unsigned* err, int B, int T,
                       hipStream_t stream);
int vts_lstm_rec3c_bwd(const float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                       float* dbpart, const float* W_hh, float* Wp, void* dg_planes, void* fws, unsigned* err, int B, int T,
                       hipStream_t stream);
int vts_lstm_rec3_bwd(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                      float* dbpart, const float* W_hh, float* Wp, int B, int T, int H, void* dg_planes,
                      hipStream_t stream);
int vts_lstm_rec3_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp,
                      int B, int T, int H, void* h_planes, hipStream_t stream);
int vts_lstm_rec_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp, int B, int T,
                     int H, hipStream_t stream);
int vts_lstm_rec_bwd(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                     float* dbpart, const float* W_hh, float* Wp, int B, int T, int H, hipStream_t stream);
int vts_colsum_partials2(const float* part, float* out, float* out2, int nparts, int N, long pitch, int accumulate,
                         hipStream_t stream);
int vts_colsum_partials(const float* part, float* out, int nparts, int N, long pitch, int accumulate,
                        hipStream_t stream);
int vts_ew_reparam_fwd(int mode, const float* mean, const float* logvar, const float* scale, const float* shape,
                       const float* eps, const float* u, const float* choice, const float* pi, float* z, long B,
                       int L, hipStream_t st);
int vts_ew_reparam_bwd(int mode, const float* dz, const float* logvar, const float* scale, const float* shape,
                       const float* eps, const float* u, const float* choice, const float* pi, float* dmean,
                       float* dlogvar, float* dscale, float* dshape, long B, int L, hipStream_t st);
int vts_ew_reparam_kl_blocks(long B, int L);
int vts_ew_reparam_kl_fwd(int mode, const float* mean, const float* logvar, const float* scale, const float* shape,
                          const float* eps, const float* u, const float* choice, const float* pi, float thr, float* z,
                          float* part, long B, int L, hipStream_t st);
int vts_ew_reparam_kl_bwd(int mode, const float* dz, const float* dpart, const float* mean, const float* logvar,
                          const float* scale, const float* shape, const float* eps, const float* u,
                          const float* choice, const float* pi, float thr, float* dmean, float* dlogvar,
                          float* dscale, float* dshape, long B, int L, hipStream_t st);
int vts_ew_elbo_recon_fwd(int kind, const float* recon, const float* y, const float* klpart, int nkl,
                          const float* pi, float thr, float beta, long B, int fs, int L, float* part, float* out10,
                          hipStream_t st);
int vts_ew_elbo_recon_bwd(int kind, const float* recon, const float* y, const float* coef, const float* gout,
                          float thr, long B, int fs, float* drecon, float* dklpart, int nkl, float* dpi,
                          hipStream_t st);
int vts_ew_elbo_blocks(long nrec, long nlat);
int vts_ew_elbo_fwd(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                    const float* scale, const float* shape, const float* pi, float thr, float beta, long B, int fs,
                    int L, float* part, float* out10, hipStream_t st);
long vts_ew_elbo_static_max_elements();
int vts_ew_elbo_fwd_static(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                           const float* scale, const float* shape, const float* pi, float thr, float beta, long B,
                           int fs, int L, float* out10, hipStream_t st);
int vts_ew_elbo_bwd(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                    const float* scale, const float* shape, const float* coef, const float* gout, float thr, long B,
                    int fs, int L, float* drecon, float* dmean, float* dlogvar, float* dscale, float* dshape,
                    float* dpi, hipStream_t st);
int vts_ew_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, float* mean,
                         float* rstd, int M, int H, float eps, hipStream_t st);
int vts_ew_layernorm_bwd_blocks(int M);
int vts_ew_layernorm_bwd(const float* dy, const float* x, long ldx, const float* gamma, const float* mean,
                         const float* rstd, float* dx, long lddx, float* part, int M, int H, hipStream_t st);
int vts_ew_act_bwd(int kind, const float* dy, const float* y, const float* aux, float scale, float* dpre, long n,
                   hipStream_t st);
int vts_ew_colsum_chunks(int M);
int vts_ew_colsum_rows(const float* a, long lda, int M, int N, float* part, hipStream_t st);
int vts_ew_scale(const float* in, const float* s, float* out, long n, hipStream_t st);
int vts_ew_gather_chunks(const float* const* src, const long long* numel, const long long* dst_off, long count,
                         float* dst, hipStream_t st);
int vts_ew_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                float bc1, float bc2_sqrt, float gscale, hipStream_t st);
int vts_ew_tile16_convert(const float* rows, float* tiled, int B, int T, int N, int to_rows, hipStream_t st);

// ---- error text ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void vts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline size_t round256(size_t n) { return (n + 255) & ~(size_t)255; }
static inline long pad16(long b) { return (b + 15) / 16 * 16; }

#define VTS_MODE_CHECK(mode)                                                                        \
  do {                                                                                              \
    if ((mode) != 0 && (mode) != 1) {                                                               \
      vts_set_error("dtype_mode %d unknown (0 = f32 matrix cores, 1 = bf16x3 split GEMMs)", (int)(mode)); \
      return VTS_ERR_UNSUPPORTED;                                                                   \
    }                                                                                               \
  } while (0)

#define VTS_RC(call)        \
  do {                      \
    int rc__ = (call);      \
    if (rc__) return rc__;  \
  } while (0)

static bool fits_int(int64_t v) { return v >= 0 && v < (int64_t)2147483647; }

// Arithmetic of a Linear product: the split-bf16 kernels pay off once the product is large enough to
// fill the chip; the small heads and decoder layers stay on the f32 matrix cores.
static int linear_mode(int dtype_mode, int64_t M, int64_t N, int64_t K) {
  return (dtype_mode == 1 && M * N * K >= ((int64_t)1 << 26)) ? 1 : 0;
}

extern "C" {

const char* vts_version(void) { return "vts-hip 0.7 (gfx950, fp32 + bf16x3 mfma, lstm + transformer encoders, plane GEMMs, cluster and pair recurrences, fused chains, recomputing feed-forward backward)"; }
const char* vts_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ LSTM
size_t vts_lstm_gates_bytes(int64_t B, int64_t T, int64_t H) { return (size_t)pad16(B) * T * 4 * H * sizeof(float); }
size_t vts_lstm_cells_bytes(int64_t B, int64_t T, int64_t H) { return (size_t)pad16(B) * T * H * sizeof(float); }
size_t vts_lstm_seqgrad_bytes(int64_t B, int64_t T, int64_t F) { return (size_t)pad16(B) * T * F * sizeof(float); }

// workspace of the recurrent kernels: packed W_hh image + per-batch-tile bias partials
static size_t rec_ws_pack_bytes(int64_t H) { return round256((size_t)4 * H * H * sizeof(float)); }
static size_t rec_ws_dbpart_bytes(int64_t B, int64_t H) {
  return round256((size_t)(pad16(B) / 16) * 4 * H * sizeof(float));
}
static size_t rec_ws_seed_bytes(int64_t B, int64_t H) { return round256((size_t)pad16(B) * H * sizeof(float)); }
// [packed W_hh image][bias-gradient partials][dh_T in tile16 order][dc_T in tile16 order]
// ... [status word of the launch, 256 B][hand-off flags of the backward cluster kernel: four 128-byte lines per 16 windows]
static size_t rec_ws_bflag_bytes(int64_t B) { return round256((size_t)(pad16(B) / 16) * 512); }
// ... [exchange buffer of the backward pair kernel: hidden 256 and whole pairs of window tiles, up to the largest
// batch whose grid (one workgroup per 16 windows) is resident at once on the 256 CUs of the one device this library
// is built for; larger batches never take the pair kernel (vts_lstm_rec3p_usable) and carry no such region]
static size_t rec_ws_pair_bytes(int64_t B, int64_t H) {
  return (H == 256 && B % 32 == 0 && B <= 4096) ? round256(vts_lstm_rec3p_xchg_bytes((int)B)) : 0;
}
size_t vts_lstm_recurrent_workspace_bytes(int64_t B, int64_t T, int64_t H) {
  (void)T;
  return rec_ws_pack_bytes(H) + rec_ws_dbpart_bytes(B, H) + 2 * rec_ws_seed_bytes(B, H) + 256 + rec_ws_bflag_bytes(B) +
         rec_ws_pair_bytes(B, H);
}
static unsigned* rec_ws_status(void* workspace, int64_t B, int64_t H) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + rec_ws_pack_bytes(H) + rec_ws_dbpart_bytes(B, H) +
                                     2 * rec_ws_seed_bytes(B, H));
}
size_t vts_lstm_wgrad_workspace_bytes(int64_t B, int64_t T, int64_t F, int64_t H) {
  const int M = (int)(4 * H);
  const long K = pad16(B) * T;
  size_t need = 0;
  // either arithmetic mode may be asked for with this workspace; the bf16x3 plan may cut the columns in two
  for (int mode = 0; mode < 2; ++mode) {
    const int cut = vts_gemm_wgrad_cut(M, (int)F, K, mode);
    const int parts[2] = {cut ? cut : (int)F, cut ? (int)F - cut : 0};
    for (int n : parts)
      if (n > 0) {
        const size_t b = vts_gemm_slab_bytes(M, n, vts_gemm_pick_ksplit(M, n, K, mode));
        need = b > need ? b : need;
      }
  }
  if (vts_gemm_wgrad_pair_usable(M, (int)F, (int)H, K, 1)) {
    const size_t b = vts_gemm_wgrad_pair_slab_bytes(M, (int)F, (int)H, K, 1);
    need = b > need ? b : need;
  }
  return round256(need) + 256;
}
size_t vts_lstm_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H) {
  const size_t w1 = vts_lstm_wgrad_workspace_bytes(B, T, Cin, H), w2 = vts_lstm_wgrad_workspace_bytes(B, T, H, H);
  return vts_lstm_recurrent_workspace_bytes(B, T, H) + (w1 > w2 ? w1 : w2);
}

static int lstm_check(int64_t B, int64_t T, int64_t Cin, int64_t H) {
  VTS_REQUIRE(B > 0 && T > 0 && Cin > 0, "lstm: B, T, Cin must be positive (got %ld, %ld, %ld)", (long)B, (long)T,
              (long)Cin);
  if (!vts_lstm_hidden_supported((int)H)) {
    vts_set_error("lstm: hidden size %ld not supported (16, 32, 64, 128, 256)", (long)H);
    return VTS_ERR_UNSUPPORTED;
  }
  VTS_REQUIRE(fits_int(pad16(B) * T) && fits_int(pad16(B) * T * 4 * H / 16), "lstm: B*T too large");
  return VTS_OK;
}

int vts_lstm_inproj_fwd(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* gates,
                        int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, Cin, H));
  VTS_REQUIRE(x && w_ih && gates, "lstm_inproj_fwd: null pointer");
  VTS_REQUIRE(vts_aligned16(gates), "lstm_inproj_fwd: gates must be 16-byte aligned");
  return vts_gemm_inproj(x, w_ih, b_ih, b_hh, gates, (int)B, (int)T, (int)Cin, (int)(4 * H), dtype_mode,
                         reinterpret_cast<hipStream_t>(stream));
}

static int recurrent_fwd_impl(float* gates, float* cells, float* hseq, float* h_last, float* c_last, const float* w_hh,
                              void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H, int dtype_mode,
                              void* h_planes, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, 1, H));
  VTS_REQUIRE(gates && cells && (hseq || h_planes) && w_hh && workspace, "lstm_recurrent_fwd: null pointer");
  VTS_REQUIRE(vts_aligned16(gates) && vts_aligned16(cells) && (!hseq || vts_aligned16(hseq)) && vts_aligned16(workspace) &&
                  (!h_last || vts_aligned16(h_last)),
              "lstm_recurrent_fwd: gates, cells, hseq, h_last and workspace must be 16-byte aligned");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_recurrent_workspace_bytes(B, T, H),
              "lstm_recurrent_fwd: workspace %zu B too small", workspace_bytes);
  // bench-sized bf16x3 launches: W_hh stationary in clusters of eight workgroups.  Its exchange buffers and flags
  // live in the parts of the workspace that only the backward pass uses (gradient seeds, bias partials), which
  // are exactly large enough: 2 KiB and 256 B per window against 2 KiB and ~8 B needed.
  static const bool cluster_on = [] { const char* e = getenv("VTS_LSTM_CLUSTER"); return !(e && e[0] == '0'); }();
  // status word of this launch (vts_lstm_recurrent_fwd_status): only the cluster kernel can set it (its packing
  // kernel clears it; the other kernels' launches clear it here)
  unsigned* status = rec_ws_status(workspace, B, H);
  if (dtype_mode == 1 && cluster_on && vts_lstm_rec3c_usable((int)B, (int)T, (int)H)) {
    char* ws = reinterpret_cast<char*>(workspace);
    char* fws = ws + rec_ws_pack_bytes(H);
    char* xws = fws + rec_ws_dbpart_bytes(B, H);
    const int rc = vts_lstm_rec3c_fwd(gates, cells, hseq, h_last, c_last, w_hh, reinterpret_cast<float*>(workspace), xws, fws,
                                      status, (int)B, (int)T, h_planes, reinterpret_cast<hipStream_t>(stream));
    // the runtime would not guarantee a resident grid (at most the packing kernel ran): take the one-workgroup kernel
    if (rc != VTS_ERR_UNSUPPORTED) return rc;
  }
  if (hipMemsetAsync(status, 0, sizeof(unsigned), reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    vts_set_error("lstm_recurrent_fwd: clearing the status word failed");
    return VTS_ERR_HIP;
  }
  // bf16x3 recurrence for the hidden sizes it is built for; smaller ones stay on the fp32 kernel
  if (dtype_mode == 1 && vts_lstm_rec3_supported((int)H))
    return vts_lstm_rec3_fwd(gates, cells, hseq, h_last, c_last, w_hh, reinterpret_cast<float*>(workspace), (int)B,
                             (int)T, (int)H, h_planes, reinterpret_cast<hipStream_t>(stream));
  VTS_REQUIRE(!h_planes, "lstm_recurrent_fwd: plane output needs dtype_mode 1");
  return vts_lstm_rec_fwd(gates, cells, hseq, h_last, c_last, w_hh, reinterpret_cast<float*>(workspace), (int)B,
                          (int)T, (int)H, reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_recurrent_fwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H,
                                  int32_t* status, vts_stream_t stream) {
  VTS_RC(lstm_check(B, T, 1, H));
  VTS_REQUIRE(workspace && status, "lstm_recurrent_fwd_status: null pointer");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_recurrent_workspace_bytes(B, T, H),
              "lstm_recurrent_fwd_status: workspace %zu B too small", workspace_bytes);
  if (hipMemcpyAsync(status, rec_ws_status(const_cast<void*>(workspace), B, H), sizeof(int32_t), hipMemcpyDeviceToDevice,
                     reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    vts_set_error("lstm_recurrent_fwd_status: copy failed");
    return VTS_ERR_HIP;
  }
  return VTS_OK;
}

int vts_lstm_recurrent_bwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H,
                                  int32_t* status, vts_stream_t stream) {
  return vts_lstm_recurrent_fwd_status(workspace, workspace_bytes, B, T, H, status, stream);
}

int vts_lstm_recurrent_fwd(float* gates, float* cells, float* hseq, float* h_last, float* c_last, const float* w_hh,
                           void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H, int dtype_mode,
                           vts_stream_t stream) {
  VTS_REQUIRE(hseq, "lstm_recurrent_fwd: null pointer");
  return recurrent_fwd_impl(gates, cells, hseq, h_last, c_last, w_hh, workspace, workspace_bytes, B, T, H, dtype_mode,
                            nullptr, stream);
}

static std::atomic<int> g_cluster_bwd{[] { const char* e = getenv("VTS_LSTM_CLUSTER_BWD"); return (e && e[0] == '1') ? 1 : 0; }()};
void vts_debug_set_cluster_bwd(int on) { g_cluster_bwd.store(on ? 1 : 0); }
void vts_debug_set_staged_bwd(int on) { vts_lstm_rec3_set_staged_bwd(on); }
static std::atomic<int> g_pair_bwd{[] { const char* e = getenv("VTS_LSTM_PAIR_BWD"); return (e && e[0] == '0') ? 0 : 1; }()};
void vts_debug_set_pair_bwd(int on) { g_pair_bwd.store(on ? 1 : 0); }
void vts_debug_set_pair_fault(int member) { vts_lstm_rec3p_set_fault(member); }

static int recurrent_bwd_impl(float* gates, const float* cells, const float* dh_seq_t16, const float* dh_last,
                              const float* dc_last, const float* w_hh, float* db_ih, float* db_hh, int accumulate,
                              void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H, int dtype_mode,
                              void* dg_planes, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, 1, H));
  VTS_REQUIRE(gates && cells && w_hh && workspace, "lstm_recurrent_bwd: null pointer");
  VTS_REQUIRE(vts_aligned16(gates) && vts_aligned16(cells) && vts_aligned16(workspace) &&
                  (!dh_seq_t16 || vts_aligned16(dh_seq_t16)),
              "lstm_recurrent_bwd: tile16 buffers and workspace must be 16-byte aligned");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_recurrent_workspace_bytes(B, T, H),
              "lstm_recurrent_bwd: workspace %zu B too small", workspace_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* wpack = reinterpret_cast<float*>(workspace);
  float* dbpart = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + rec_ws_pack_bytes(H));
  const int nbt = (int)(pad16(B) / 16);
  // status word of this launch (vts_lstm_recurrent_bwd_status): the pair kernel's packing launch and the cluster
  // branch clear it themselves; for every other kernel - none of which can set it - it is cleared below, so that
  // the query never returns what an earlier launch or the allocator left in the workspace
  bool status_cleared = false;
  if (dtype_mode == 1 && vts_lstm_rec3_supported((int)H)) {
    // the bf16x3 kernel takes the gradients w.r.t. h_T / c_T as tile16 blocks (one coalesced load
    // per accumulator tile instead of per-lane row gathers inside a register-bound kernel)
    char* seed = reinterpret_cast<char*>(dbpart) + rec_ws_dbpart_bytes(B, H);
    float* seed_h = dh_last ? reinterpret_cast<float*>(seed) : nullptr;
    float* seed_c = dc_last ? reinterpret_cast<float*>(seed + rec_ws_seed_bytes(B, H)) : nullptr;
    if (seed_h) VTS_RC(vts_ew_tile16_convert(dh_last, seed_h, (int)B, 1, (int)H, 0, st));
    if (seed_c) VTS_RC(vts_ew_tile16_convert(dc_last, seed_c, (int)B, 1, (int)H, 0, st));
    // plane output at bench size: W_hh stationary in clusters of eight workgroups, as the forward pass; when the
    // runtime will not guarantee a resident grid (at most a packing kernel ran) the one-workgroup kernel takes over
    // (Off unless asked for - VTS_LSTM_CLUSTER_BWD=1 at start-up or vts_debug_set_cluster_bwd(1): measured at
    // 4096 x 96, hidden 256, it is 3 % slower than the one-workgroup kernel.  Its members exchange dA, four
    // times the forward kernel's h, and a CU's vector memory pipe moves the 610 KiB a step then needs no faster
    // than the 1 MiB weight image of the one-workgroup kernel.)
    int rc = VTS_ERR_UNSUPPORTED;
    // first choice at bench size: pairs of workgroups that split W_hh and the hidden units between them
    // (lstm_rec_pair.hip; VTS_LSTM_PAIR_BWD=0 at start-up or vts_debug_set_pair_bwd(0) turns it off)
    if (dg_planes && g_pair_bwd.load() > 0 && g_cluster_bwd.load() == 0 && rec_ws_pair_bytes(B, H) > 0 &&
        vts_lstm_rec3p_usable((int)B, (int)T, (int)H)) {
      unsigned* status = rec_ws_status(workspace, B, H);       // (cleared by the kernel that packs W_hh)
      rc = vts_lstm_rec3p_bwd(gates, cells, dh_seq_t16, seed_h, seed_c, dbpart, w_hh, wpack, dg_planes,
                              reinterpret_cast<char*>(status) + 256 + rec_ws_bflag_bytes(B), status, (int)B, (int)T, st);
      if (rc != VTS_OK && rc != VTS_ERR_UNSUPPORTED) return rc;
      status_cleared = rc == VTS_OK;
    }
    if (rc == VTS_ERR_UNSUPPORTED && dg_planes && g_cluster_bwd.load() > 0 && vts_lstm_rec3c_usable((int)B, (int)T, (int)H)) {
      unsigned* status = rec_ws_status(workspace, B, H);
      if (hipMemsetAsync(status, 0, sizeof(unsigned), st) != hipSuccess) {
        vts_set_error("lstm_recurrent_bwd: clearing the status word failed");
        return VTS_ERR_HIP;
      }
      rc = vts_lstm_rec3c_bwd(gates, cells, dh_seq_t16, seed_h, seed_c, dbpart, w_hh, wpack, dg_planes,
                              reinterpret_cast<char*>(status) + 256, status, (int)B, (int)T, st);
      if (rc != VTS_OK && rc != VTS_ERR_UNSUPPORTED) return rc;
      status_cleared = true;
    }
    if (rc == VTS_ERR_UNSUPPORTED)
    VTS_RC(vts_lstm_rec3_bwd(gates, cells, dh_seq_t16, seed_h, seed_c, dbpart, w_hh, wpack, (int)B, (int)T, (int)H,
                             dg_planes, st));
  } else {
    VTS_REQUIRE(!dg_planes, "lstm_recurrent_bwd: plane output needs dtype_mode 1 and a hidden size of 256");
    VTS_RC(vts_lstm_rec_bwd(gates, cells, dh_seq_t16, dh_last, dc_last, dbpart, w_hh, wpack, (int)B, (int)T, (int)H, st));
  }
  if (!status_cleared && hipMemsetAsync(rec_ws_status(workspace, B, H), 0, sizeof(unsigned), st) != hipSuccess) {
    vts_set_error("lstm_recurrent_bwd: clearing the status word failed");
    return VTS_ERR_HIP;
  }
  if (db_ih && db_hh && db_ih != db_hh) VTS_RC(vts_colsum_partials2(dbpart, db_ih, db_hh, nbt, (int)(4 * H), 4 * H, accumulate, st));
  else if (db_ih) VTS_RC(vts_colsum_partials(dbpart, db_ih, nbt, (int)(4 * H), 4 * H, accumulate, st));
  else if (db_hh) VTS_RC(vts_colsum_partials(dbpart, db_hh, nbt, (int)(4 * H), 4 * H, accumulate, st));
  return VTS_OK;
}

int vts_lstm_recurrent_bwd(float* gates, const float* cells, const float* dh_seq_t16, const float* dh_last,
                           const float* dc_last, const float* w_hh, float* db_ih, float* db_hh, int accumulate,
                           void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H, int dtype_mode,
                           vts_stream_t stream) {
  return recurrent_bwd_impl(gates, cells, dh_seq_t16, dh_last, dc_last, w_hh, db_ih, db_hh, accumulate, workspace,
                            workspace_bytes, B, T, H, dtype_mode, nullptr, stream);
}

// ---- the layer's products over bf16 plane operands (csrc/lstm_planes.hip, csrc/gemm_planes.h) ----
int vts_lstm_planes_supported(int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode) {
  static const bool on = [] { const char* e = getenv("VTS_LSTM_PLANES"); return !(e && e[0] == '0'); }();
  if (!on || dtype_mode != 1 || Cin <= 0 || !fits_int(pad16(B) * T * 4 * H / 16)) return 0;
  return vts_planes_usable((long)B, (long)T, (long)H) ? 1 : 0;
}
size_t vts_lstm_planes_bytes(int64_t B, int64_t T, int64_t F, int is_hseq) {
  return vts_planes_seq_bytes((long)B, (long)T, (long)F, is_hseq ? 1 : 0);
}
size_t vts_lstm_planes_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H) {
  const size_t w1 = vts_planes_weight_bytes(4 * H, Cin), w2 = vts_planes_weight_bytes(Cin, 4 * H);
  const size_t w = w1 > w2 ? w1 : w2, sl = vts_planes_wgrad_slab_bytes((long)B, (long)T, (long)Cin, (long)H);
  return round256(w > sl ? w : sl) + 256;
}
#define VTS_PLANES_CHECK(what)                                                                         \
  VTS_MODE_CHECK(dtype_mode);                                                                          \
  VTS_RC(lstm_check(B, T, Cin, H));                                                                    \
  if (!vts_lstm_planes_supported(B, T, Cin, H, dtype_mode)) {                                          \
    vts_set_error(what ": plane operands are not planned for B %ld, T %ld, H %ld, mode %d (see "       \
                  "vts_lstm_planes_supported)", (long)B, (long)T, (long)H, dtype_mode);                \
    return VTS_ERR_UNSUPPORTED;                                                                        \
  }

int vts_lstm_planes_from_rows(const float* x, void* x_planes, int64_t B, int64_t T, int64_t Cin, int64_t H,
                              int dtype_mode, vts_stream_t stream) {
  VTS_PLANES_CHECK("lstm_planes_from_rows");
  VTS_REQUIRE(x && x_planes && vts_aligned16(x_planes), "lstm_planes_from_rows: null or misaligned pointer");
  return vts_planes_from_rows(x, x_planes, (long)B, (long)T, (long)Cin, reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_inproj_fwd_planes(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih,
                               const float* b_hh, float* gates, void* workspace, size_t workspace_bytes, int64_t B,
                               int64_t T, int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_PLANES_CHECK("lstm_inproj_fwd_planes");
  VTS_REQUIRE(x_planes && w_ih && b_ih && b_hh && gates && workspace, "lstm_inproj_fwd_planes: null pointer");
  VTS_REQUIRE(vts_aligned16(x_planes) && vts_aligned16(gates) && vts_aligned16(workspace), "lstm_inproj_fwd_planes: 16-byte alignment");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_planes_workspace_bytes(B, T, Cin, H), "lstm_inproj_fwd_planes: workspace %zu B too small", workspace_bytes);
  return vts_planes_inproj(x_planes, x_is_hseq, w_ih, b_ih, b_hh, gates, workspace, (long)B, (long)T, (long)Cin, (long)H,
                           reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_recurrent_fwd_planes(float* gates, float* cells, float* hseq, void* h_planes, float* h_last, float* c_last,
                                  const float* w_hh, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                                  int64_t H, int dtype_mode, vts_stream_t stream) {
  const int64_t Cin = H;
  VTS_PLANES_CHECK("lstm_recurrent_fwd_planes");
  VTS_REQUIRE(h_planes && vts_aligned16(h_planes), "lstm_recurrent_fwd_planes: null or misaligned h_planes");
  return recurrent_fwd_impl(gates, cells, hseq, h_last, c_last, w_hh, workspace, workspace_bytes, B, T, H, dtype_mode,
                            h_planes, stream);
}

// ---- the layer's forward pass as one kernel (csrc/lstm_rec_fused.hip) ----
static std::atomic<int> g_fused_fwd{[] { const char* e = getenv("VTS_LSTM_FUSED_FWD"); return (e && e[0] == '1') ? 1 : 0; }()};
void vts_debug_set_fused_fwd(int on) { g_fused_fwd.store(on ? 1 : 0); }
void vts_debug_set_fused_fault(int member) { vts_lstm_fused_set_fault(member); }
int vts_lstm_fused_fwd_preferred(void) { return g_fused_fwd.load(); }
int vts_lstm_fused_fwd_supported(int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode) {
  if (!vts_lstm_planes_supported(B, T, Cin, H, dtype_mode)) return 0;
  return vts_lstm_fused_usable((long)B, (long)T, (long)Cin, (long)H) ? 1 : 0;
}
// [weight image][flags][status word, 256 B]
size_t vts_lstm_fused_fwd_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H) {
  (void)T; (void)H;
  return round256(vts_lstm_fused_image_bytes((long)Cin)) + round256(vts_lstm_fused_flag_bytes((long)B)) + 256;
}
static unsigned* fused_ws_status(void* workspace, int64_t B, int64_t Cin) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + round256(vts_lstm_fused_image_bytes((long)Cin)) +
                                     round256(vts_lstm_fused_flag_bytes((long)B)));
}
int vts_lstm_fused_fwd_planes(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih,
                              const float* b_hh, const float* w_hh, float* gates, float* cells, float* hseq,
                              void* h_planes, float* h_last, float* c_last, void* workspace, size_t workspace_bytes,
                              int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_PLANES_CHECK("lstm_fused_fwd_planes");
  if (!vts_lstm_fused_fwd_supported(B, T, Cin, H, dtype_mode)) {
    vts_set_error("lstm_fused_fwd_planes: not planned for B %ld, T %ld, Cin %ld (see vts_lstm_fused_fwd_supported)", (long)B,
                  (long)T, (long)Cin);
    return VTS_ERR_UNSUPPORTED;
  }
  VTS_REQUIRE(x_planes && w_ih && b_ih && b_hh && w_hh && gates && cells && h_planes && workspace, "lstm_fused_fwd_planes: null pointer");
  VTS_REQUIRE(vts_aligned16(x_planes) && vts_aligned16(gates) && vts_aligned16(cells) && vts_aligned16(h_planes) &&
                  vts_aligned16(workspace), "lstm_fused_fwd_planes: 16-byte alignment");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_fused_fwd_workspace_bytes(B, T, Cin, H), "lstm_fused_fwd_planes: workspace %zu B too small", workspace_bytes);
  char* ws = reinterpret_cast<char*>(workspace);
  return vts_lstm_fused_fwd(x_planes, x_is_hseq, w_ih, b_ih, b_hh, w_hh, gates, cells, hseq, h_planes, h_last, c_last, ws,
                            ws + round256(vts_lstm_fused_image_bytes((long)Cin)), fused_ws_status(workspace, B, Cin), (long)B,
                            (long)T, (long)Cin, reinterpret_cast<hipStream_t>(stream));
}
int vts_lstm_fused_fwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin,
                              int64_t H, int32_t* status, vts_stream_t stream) {
  VTS_RC(lstm_check(B, T, Cin, H));
  VTS_REQUIRE(workspace && status, "lstm_fused_fwd_status: null pointer");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_fused_fwd_workspace_bytes(B, T, Cin, H), "lstm_fused_fwd_status: workspace %zu B too small", workspace_bytes);
  if (hipMemcpyAsync(status, fused_ws_status(const_cast<void*>(workspace), B, Cin), sizeof(int32_t), hipMemcpyDeviceToDevice,
                     reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    vts_set_error("lstm_fused_fwd_status: copy failed");
    return VTS_ERR_HIP;
  }
  return VTS_OK;
}

int vts_lstm_recurrent_bwd_planes(const float* gates, const float* cells, const float* dh_seq_t16, const float* dh_last,
                                  const float* dc_last, const float* w_hh, void* dg_planes, float* db_ih, float* db_hh,
                                  int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                                  int64_t H, int dtype_mode, vts_stream_t stream) {
  const int64_t Cin = H;
  VTS_PLANES_CHECK("lstm_recurrent_bwd_planes");
  VTS_REQUIRE(dg_planes && vts_aligned16(dg_planes), "lstm_recurrent_bwd_planes: null or misaligned dg_planes");
  return recurrent_bwd_impl(const_cast<float*>(gates), cells, dh_seq_t16, dh_last, dc_last, w_hh, db_ih, db_hh, accumulate,
                            workspace, workspace_bytes, B, T, H, dtype_mode, dg_planes, stream);
}

int vts_lstm_dgrad_planes(const void* dg_planes, const float* w_ih, float* dx_t16, void* workspace,
                          size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode,
                          vts_stream_t stream) {
  VTS_PLANES_CHECK("lstm_dgrad_planes");
  VTS_REQUIRE(dg_planes && w_ih && dx_t16 && workspace, "lstm_dgrad_planes: null pointer");
  VTS_REQUIRE((Cin % 256) == 0, "lstm_dgrad_planes: needs Cin %% 256 == 0 (got %ld)", (long)Cin);
  VTS_REQUIRE(vts_aligned16(dg_planes) && vts_aligned16(dx_t16) && vts_aligned16(workspace), "lstm_dgrad_planes: 16-byte alignment");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_planes_workspace_bytes(B, T, Cin, H), "lstm_dgrad_planes: workspace %zu B too small", workspace_bytes);
  return vts_planes_dgrad(dg_planes, w_ih, dx_t16, workspace, (long)B, (long)T, (long)Cin, (long)H,
                          reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_wgrad_planes(const void* dg_planes, const void* x_planes, int x_is_hseq, const void* h_planes, float* dw_ih,
                          float* dw_hh, int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                          int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_PLANES_CHECK("lstm_wgrad_planes");
  VTS_REQUIRE(dg_planes && x_planes && h_planes && dw_ih && dw_hh && workspace, "lstm_wgrad_planes: null pointer");
  VTS_REQUIRE(vts_aligned16(dg_planes) && vts_aligned16(x_planes) && vts_aligned16(h_planes) && vts_aligned16(workspace),
              "lstm_wgrad_planes: 16-byte alignment");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_planes_workspace_bytes(B, T, Cin, H), "lstm_wgrad_planes: workspace %zu B too small", workspace_bytes);
  return vts_planes_wgrad(dg_planes, x_planes, x_is_hseq, h_planes, dw_ih, dw_hh, accumulate,
                          reinterpret_cast<float*>(workspace), (long)B, (long)T, (long)Cin, (long)H,
                          reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_dgrad(const float* dgates, const float* w_ih, float* dx_t16, int64_t B, int64_t T, int64_t Cin,
                   int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, Cin, H));
  VTS_REQUIRE(dgates && w_ih && dx_t16, "lstm_dgrad: null pointer");
  VTS_REQUIRE((Cin % 16) == 0, "lstm_dgrad: needs Cin %% 16 == 0 (got %ld)", (long)Cin);
  VTS_REQUIRE(vts_aligned16(dgates) && vts_aligned16(dx_t16) && vts_aligned16(w_ih), "lstm_dgrad: 16-byte alignment");
  return vts_gemm_dgrad_t16(dgates, w_ih, dx_t16, (int)B, (int)T, (int)Cin, (int)(4 * H), 0, dtype_mode,
                            reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_wgrad(const float* dgates, const float* x, int tshift, float* dw, int accumulate, void* workspace,
                   size_t workspace_bytes, int64_t B, int64_t T, int64_t F, int64_t H, int dtype_mode,
                   vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, F, H));
  VTS_REQUIRE(dgates && x && dw && workspace, "lstm_wgrad: null pointer");
  VTS_REQUIRE(tshift == 0 || tshift == -1, "lstm_wgrad: tshift must be 0 (input weights) or -1 (recurrent weights)");
  VTS_REQUIRE(vts_aligned16(dgates) && vts_aligned16(workspace), "lstm_wgrad: 16-byte alignment");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_wgrad_workspace_bytes(B, T, F, H), "lstm_wgrad: workspace %zu B too small",
              workspace_bytes);
  const int ks = vts_gemm_pick_ksplit((int)(4 * H), (int)F, pad16(B) * T, dtype_mode);
  return vts_gemm_wgrad_t16(dgates, x, dw, reinterpret_cast<float*>(workspace), ks, (int)B, (int)T, (int)F, F,
                            (int)(4 * H), tshift, accumulate, dtype_mode, reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_wgrad_pair_supported(int64_t B, int64_t T, int64_t F, int64_t H, int dtype_mode) {
  static const bool pair_on = [] { const char* e = getenv("VTS_LSTM_WGRAD_PAIR"); return !(e && e[0] == '0'); }();
  if (!pair_on || B <= 0 || T <= 0 || !fits_int(pad16(B) * T) || !fits_int(4 * H) || !fits_int(F)) return 0;
  return vts_gemm_wgrad_pair_usable((int)(4 * H), (int)F, (int)H, pad16(B) * T, dtype_mode) ? 1 : 0;
}

int vts_lstm_wgrad_pair(const float* dgates, const float* x, const float* hseq, float* dw_ih, float* dw_hh,
                        int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t F,
                        int64_t H, int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_RC(lstm_check(B, T, F, H));
  VTS_REQUIRE(dgates && x && hseq && dw_ih && dw_hh && workspace, "lstm_wgrad_pair: null pointer");
  VTS_REQUIRE(vts_aligned16(dgates) && vts_aligned16(workspace), "lstm_wgrad_pair: 16-byte alignment");
  if (!vts_lstm_wgrad_pair_supported(B, T, F, H, dtype_mode)) {
    vts_set_error("lstm_wgrad_pair: not planned for B %ld, T %ld, F %ld, H %ld, mode %d (see "
                  "vts_lstm_wgrad_pair_supported)", (long)B, (long)T, (long)F, (long)H, dtype_mode);
    return VTS_ERR_UNSUPPORTED;
  }
  VTS_REQUIRE(workspace_bytes >= vts_lstm_wgrad_workspace_bytes(B, T, F, H),
              "lstm_wgrad_pair: workspace %zu B too small", workspace_bytes);
  return vts_gemm_wgrad_pair_t16(dgates, x, F, hseq, dw_ih, dw_hh, reinterpret_cast<float*>(workspace), (int)B, (int)T,
                                 (int)F, (int)H, (int)(4 * H), accumulate, dtype_mode,
                                 reinterpret_cast<hipStream_t>(stream));
}

int vts_lstm_layer_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       float* gates, float* cells, float* hseq, float* h_last, float* c_last, void* workspace,
                       size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode,
                       vts_stream_t stream) {
  VTS_REQUIRE(workspace_bytes >= vts_lstm_workspace_bytes(B, T, Cin, H), "lstm_layer_fwd: workspace %zu B too small",
              workspace_bytes);
  VTS_RC(vts_lstm_inproj_fwd(x, w_ih, b_ih, b_hh, gates, B, T, Cin, H, dtype_mode, stream));
  return vts_lstm_recurrent_fwd(gates, cells, hseq, h_last, c_last, w_hh, workspace, workspace_bytes, B, T, H,
                                dtype_mode, stream);
}

int vts_lstm_layer_bwd(const float* x, const float* hseq, const float* w_ih, const float* w_hh, float* gates,
                       const float* cells, const float* dh_seq_t16, const float* dh_last, const float* dc_last,
                       float* dx_t16, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int accumulate,
                       void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin, int64_t H,
                       int dtype_mode, vts_stream_t stream) {
  VTS_REQUIRE(x && hseq && workspace, "lstm_layer_bwd: null pointer");
  VTS_REQUIRE(workspace_bytes >= vts_lstm_workspace_bytes(B, T, Cin, H), "lstm_layer_bwd: workspace %zu B too small",
              workspace_bytes);
  const size_t rec_bytes = vts_lstm_recurrent_workspace_bytes(B, T, H);
  void* wg_ws = reinterpret_cast<char*>(workspace) + rec_bytes;
  const size_t wg_bytes = workspace_bytes - rec_bytes;
  VTS_RC(vts_lstm_recurrent_bwd(gates, cells, dh_seq_t16, dh_last, dc_last, w_hh, db_ih, db_hh, accumulate, workspace,
                                workspace_bytes, B, T, H, dtype_mode, stream));
  if (dx_t16) VTS_RC(vts_lstm_dgrad(gates, w_ih, dx_t16, B, T, Cin, H, dtype_mode, stream));
  if (dw_ih && dw_hh && vts_lstm_wgrad_pair_supported(B, T, Cin, H, dtype_mode))
    return vts_lstm_wgrad_pair(gates, x, hseq, dw_ih, dw_hh, accumulate, wg_ws, wg_bytes, B, T, Cin, H, dtype_mode, stream);
  if (dw_ih) VTS_RC(vts_lstm_wgrad(gates, x, 0, dw_ih, accumulate, wg_ws, wg_bytes, B, T, Cin, H, dtype_mode, stream));
  if (dw_hh) VTS_RC(vts_lstm_wgrad(gates, hseq, -1, dw_hh, accumulate, wg_ws, wg_bytes, B, T, H, H, dtype_mode, stream));
  return VTS_OK;
}

int vts_tile16_convert(float* rows, float* tiled, int64_t B, int64_t T, int64_t N, int to_rows, vts_stream_t stream) {
  VTS_REQUIRE(rows && tiled && B > 0 && T > 0 && N > 0 && (N % 16) == 0, "tile16_convert: bad arguments");
  return vts_ew_tile16_convert(rows, tiled, (int)B, (int)T, (int)N, to_rows, reinterpret_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------- Linear
int vts_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t M, int64_t N,
                   int64_t K, int act, const float* keep, float keep_scale, float* aux, int dtype_mode,
                   vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0 && ldx >= K, "linear_fwd: bad arguments");
  VTS_REQUIRE(act >= 0 && act <= 4, "linear_fwd: unknown activation %d", act);
  VTS_REQUIRE(fits_int(M) && fits_int(N) && fits_int(K), "linear_fwd: dimension too large");
  return vts_gemm_linear_fwd(x, ldx, w, bias, y, N, (int)M, (int)N, (int)K, act, keep, keep_scale, 0ull, 0u, aux,
                             nullptr, linear_mode(dtype_mode, M, N, K), reinterpret_cast<hipStream_t>(stream));
}

int vts_linear_relu_dropout_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t M,
                                int64_t N, int64_t K, float drop_p, uint64_t seed, int dtype_mode,
                                vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0 && ldx >= K, "linear_relu_dropout_fwd: bad arguments");
  VTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_relu_dropout_fwd: drop_p %g outside [0, 1)", (double)drop_p);
  VTS_REQUIRE(fits_int(M) && fits_int(N) && fits_int(K), "linear_relu_dropout_fwd: dimension too large");
  const bool on = drop_p > 0.f && seed != 0;
  return vts_gemm_linear_fwd(x, ldx, w, bias, y, N, (int)M, (int)N, (int)K, VTS_ACT_RELU_DROPOUT, nullptr,
                             on ? 1.f / (1.f - drop_p) : 1.f, seed, on ? vts_drop_thresh(drop_p) : 0u, nullptr,
                             nullptr, linear_mode(dtype_mode, M, N, K), reinterpret_cast<hipStream_t>(stream));
}

struct LinWs {
  float* dpre;
  float* slabs;
  float* colpart;
  float* dxpart;     // [2 * ceil(M/128)][K] column sums of dx per half row tile
  int ksplit;
  size_t total;
};

static LinWs lin_ws_layout(void* base, int64_t M, int64_t N, int64_t K) {
  LinWs w{};
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  w.dpre = reinterpret_cast<float*>(p + off); off += round256((size_t)M * N * sizeof(float));
  w.ksplit = vts_gemm_pick_ksplit((int)N, (int)K, M, 0);
  w.slabs = reinterpret_cast<float*>(p + off); off += round256(vts_gemm_slab_bytes((int)N, (int)K, w.ksplit));
  w.colpart = reinterpret_cast<float*>(p + off); off += round256((size_t)vts_ew_colsum_chunks((int)M) * N * sizeof(float));
  w.dxpart = reinterpret_cast<float*>(p + off); off += round256((size_t)2 * ((M + 127) / 128) * K * sizeof(float));
  w.total = off;
  return w;
}

size_t vts_linear_bwd_workspace_bytes(int64_t M, int64_t N, int64_t K) { return lin_ws_layout(nullptr, M, N, K).total; }

static int vts_linear_bwd_ex(const float* dy, const float* y, const float* aux, int act, float keep_scale, const float* x,
                      int64_t ldx, const float* w, float* dx, int64_t lddx, int dx_accumulate, float* dw, float* db,
                      int param_accumulate, int x_act, float x_act_scale, float* dx_colsum, void* workspace,
                      size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(dy && x && w && workspace && M > 0 && N > 0 && K > 0, "linear_bwd: bad arguments");
  VTS_REQUIRE(act >= 0 && act <= 4, "linear_bwd: unknown activation %d", act);
  VTS_REQUIRE(act == 0 || y != nullptr, "linear_bwd: activation %d needs the forward output y", act);
  VTS_REQUIRE(act != 3 || aux != nullptr, "linear_bwd: clamp needs the aux mask written by linear_fwd");
  VTS_REQUIRE(x_act == VTS_ACT_NONE || x_act == VTS_ACT_RELU || x_act == VTS_ACT_RELU_DROPOUT,
              "linear_bwd: x_act must be none, relu or relu+dropout");
  VTS_REQUIRE(x_act == VTS_ACT_NONE || !dx_accumulate, "linear_bwd: x_act cannot be combined with dx_accumulate");
  VTS_REQUIRE(!dx_colsum || (dx && !dx_accumulate), "linear_bwd: dx_colsum needs dx, not accumulated");
  const LinWs ws = lin_ws_layout(workspace, M, N, K);
  VTS_REQUIRE(workspace_bytes >= ws.total, "linear_bwd: workspace %zu B < required %zu B", workspace_bytes, ws.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float* dpre = dy;
  if (act != 0) {
    VTS_RC(vts_ew_act_bwd(act, dy, y, aux, keep_scale, ws.dpre, M * N, st));
    dpre = ws.dpre;
  }
  const int mode = linear_mode(dtype_mode, M, N, K);
  if (dx) {
    const bool fuse = x_act != VTS_ACT_NONE;
    VTS_RC(vts_gemm_linear_dx(dpre, N, w, dx, lddx, (int)M, (int)N, (int)K, dx_accumulate, fuse ? x : nullptr, ldx,
                              nullptr, x_act == VTS_ACT_RELU_DROPOUT ? x_act_scale : 1.f,
                              dx_colsum ? ws.dxpart : nullptr, mode, st));
    if (dx_colsum)
      VTS_RC(vts_colsum_partials(ws.dxpart, dx_colsum, (int)(2 * ((M + 127) / 128)), (int)K, K, 0, st));
  }
  if (dw)
    VTS_RC(vts_gemm_linear_dw(dpre, N, x, ldx, dw, ws.slabs, ws.ksplit, (int)M, (int)N, (int)K, param_accumulate,
                              mode, st));
  if (db) {
    VTS_RC(vts_ew_colsum_rows(dpre, N, (int)M, (int)N, ws.colpart, st));
    VTS_RC(vts_colsum_partials(ws.colpart, db, vts_ew_colsum_chunks((int)M), (int)N, N, param_accumulate, st));
  }
  return VTS_OK;
}

int vts_linear_bwd(const float* dy, const float* y, const float* aux, int act, float keep_scale, const float* x,
                   int64_t ldx, const float* w, float* dx, int64_t lddx, int dx_accumulate, float* dw, float* db,
                   int param_accumulate, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K,
                   int dtype_mode, vts_stream_t stream) {
  return vts_linear_bwd_ex(dy, y, aux, act, keep_scale, x, ldx, w, dx, lddx, dx_accumulate, dw, db, param_accumulate,
                           VTS_ACT_NONE, 1.f, nullptr, workspace, workspace_bytes, M, N, K, dtype_mode, stream);
}

// ------------------------------------------------------------------------------------- LayerNorm
int vts_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, float* mean,
                      float* rstd, int64_t M, int64_t H, float eps, vts_stream_t stream) {
  VTS_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && H > 0 && ldx >= H, "layernorm_fwd: bad arguments");
  return vts_ew_layernorm_fwd(x, ldx, gamma, beta, y, mean, rstd, (int)M, (int)H, eps,
                              reinterpret_cast<hipStream_t>(stream));
}

size_t vts_layernorm_bwd_workspace_bytes(int64_t M, int64_t H) {
  return round256((size_t)vts_ew_layernorm_bwd_blocks((int)M) * 2 * H * sizeof(float));
}

int vts_layernorm_bwd(const float* dy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                      const float* rstd, float* dx, int64_t lddx, float* dgamma, float* dbeta, int param_accumulate,
                      void* workspace, size_t workspace_bytes, int64_t M, int64_t H, vts_stream_t stream) {
  VTS_REQUIRE(dy && x && gamma && mean && rstd && dx && workspace && M > 0 && H > 0, "layernorm_bwd: bad arguments");
  VTS_REQUIRE(H <= 4096, "layernorm_bwd: H %ld too large", (long)H);
  VTS_REQUIRE(workspace_bytes >= vts_layernorm_bwd_workspace_bytes(M, H), "layernorm_bwd: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  VTS_RC(vts_ew_layernorm_bwd(dy, x, ldx, gamma, mean, rstd, dx, lddx, part, (int)M, (int)H, st));
  const int blocks = vts_ew_layernorm_bwd_blocks((int)M);
  // partials are [blk][2][H]
  if (dgamma) VTS_RC(vts_colsum_partials(part, dgamma, blocks, (int)H, 2 * H, param_accumulate, st));
  if (dbeta) VTS_RC(vts_colsum_partials(part + H, dbeta, blocks, (int)H, 2 * H, param_accumulate, st));
  return VTS_OK;
}

// ----------------------------------------------------------------------------- reparam and ELBO
int vts_reparam_fwd(int family, const float* mean, const float* logvar, const float* scale, const float* shape,
                    const float* eps, const float* u, const float* choice, const float* pi, float* z, int64_t B,
                    int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && z && B > 0 && L > 0, "reparam_fwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar && eps), "reparam_fwd: gaussian inputs missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape && u), "reparam_fwd: gpd inputs missing");
  VTS_REQUIRE(family != VTS_MIXTURE || (choice && pi), "reparam_fwd: mixture needs choice and pi");
  return vts_ew_reparam_fwd(family, mean, logvar, scale, shape, eps, u, choice, pi, z, B, (int)L,
                            reinterpret_cast<hipStream_t>(stream));
}

int vts_reparam_bwd(int family, const float* dz, const float* logvar, const float* scale, const float* shape,
                    const float* eps, const float* u, const float* choice, const float* pi, float* dmean,
                    float* dlogvar, float* dscale, float* dshape, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && dz && B > 0 && L > 0, "reparam_bwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (logvar && eps && dmean && dlogvar), "reparam_bwd: gaussian buffers missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape && u && dscale && dshape), "reparam_bwd: gpd buffers missing");
  VTS_REQUIRE(family != VTS_MIXTURE || (choice && pi), "reparam_bwd: mixture needs choice and pi");
  return vts_ew_reparam_bwd(family, dz, logvar, scale, shape, eps, u, choice, pi, dmean, dlogvar, dscale, dshape, B,
                            (int)L, reinterpret_cast<hipStream_t>(stream));
}

size_t vts_elbo_workspace_bytes(int64_t B, int64_t fs, int64_t L) {
  return round256((size_t)vts_ew_elbo_blocks(B * fs, B * L) * 6 * sizeof(float));
}

int vts_elbo_fwd(int family, const float* recon, const float* y, const float* mean, const float* logvar,
                 const float* scale, const float* shape, const float* pi, float threshold, float beta, float* state,
                 void* workspace, size_t workspace_bytes, int64_t B, int64_t fs, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 3 && recon && y && state && workspace && B > 0 && fs > 0 && L > 0,
              "elbo_fwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar), "elbo_fwd: gaussian inputs missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape), "elbo_fwd: gpd inputs missing");
  VTS_REQUIRE(family < VTS_MIXTURE || pi, "elbo_fwd: mixture needs pi");
  VTS_REQUIRE(workspace_bytes >= vts_elbo_workspace_bytes(B, fs, L), "elbo_fwd: workspace too small");
  return vts_ew_elbo_fwd(family, recon, y, mean, logvar, scale, shape, pi, threshold, beta, B, (int)fs, (int)L,
                         reinterpret_cast<float*>(workspace), state, reinterpret_cast<hipStream_t>(stream));
}

int64_t vts_elbo_fwd_static_max_elements(void) { return vts_ew_elbo_static_max_elements(); }

int vts_elbo_fwd_static(int family, const float* recon, const float* y, const float* mean, const float* logvar,
                        const float* scale, const float* shape, const float* pi, float threshold, float beta,
                        float* state, int64_t B, int64_t fs, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 3 && recon && y && state && B > 0 && fs > 0 && L > 0,
              "elbo_fwd_static: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar), "elbo_fwd_static: gaussian inputs missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape), "elbo_fwd_static: gpd inputs missing");
  VTS_REQUIRE(family < VTS_MIXTURE || pi, "elbo_fwd_static: mixture needs pi");
  VTS_REQUIRE(fits_int(fs) && fits_int(L) && B <= ((int64_t)1 << 40), "elbo_fwd_static: dimension too large");
  return vts_ew_elbo_fwd_static(family, recon, y, mean, logvar, scale, shape, pi, threshold, beta, B, (int)fs, (int)L,
                                state, reinterpret_cast<hipStream_t>(stream));
}

int vts_elbo_bwd(int family, const float* recon, const float* y, const float* mean, const float* logvar,
                 const float* scale, const float* shape, const float* state, const float* grad_out, float threshold,
                 float* drecon, float* dmean, float* dlogvar, float* dscale, float* dshape, float* dpi, int64_t B,
                 int64_t fs, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 3 && recon && y && state && drecon && B > 0 && fs > 0 && L > 0,
              "elbo_bwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar && dmean && dlogvar), "elbo_bwd: gaussian buffers missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape && dscale && dshape), "elbo_bwd: gpd buffers missing");
  return vts_ew_elbo_bwd(family, recon, y, mean, logvar, scale, shape, state, grad_out, threshold, B, (int)fs, (int)L,
                         drecon, dmean, dlogvar, dscale, dshape, dpi, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------- fused sampling + KL, loss from sums
int64_t vts_reparam_kl_blocks(int64_t B, int64_t L) {
  if (B <= 0 || L <= 0) return 0;
  return vts_ew_reparam_kl_blocks(B, (int)L);
}

int vts_reparam_kl_fwd(int family, const float* mean, const float* logvar, const float* scale, const float* shape,
                       const float* eps, const float* u, const float* choice, const float* pi, float threshold,
                       float* z, float* klpart, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && z && klpart && B > 0 && L > 0, "reparam_kl_fwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar && eps), "reparam_kl_fwd: gaussian inputs missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape && u), "reparam_kl_fwd: gpd inputs missing");
  VTS_REQUIRE(family != VTS_MIXTURE || (choice && pi), "reparam_kl_fwd: mixture needs choice and pi");
  return vts_ew_reparam_kl_fwd(family, mean, logvar, scale, shape, eps, u, choice, pi, threshold, z, klpart, B,
                               (int)L, reinterpret_cast<hipStream_t>(stream));
}

int vts_reparam_kl_bwd(int family, const float* dz, const float* dklpart, const float* mean, const float* logvar,
                       const float* scale, const float* shape, const float* eps, const float* u, const float* choice,
                       const float* pi, float threshold, float* dmean, float* dlogvar, float* dscale, float* dshape,
                       int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && B > 0 && L > 0, "reparam_kl_bwd: bad arguments");
  VTS_REQUIRE(family == VTS_GPD || (mean && logvar && eps && dmean && dlogvar),
              "reparam_kl_bwd: gaussian buffers missing");
  VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape && u && dscale && dshape),
              "reparam_kl_bwd: gpd buffers missing");
  VTS_REQUIRE(family != VTS_MIXTURE || (choice && pi), "reparam_kl_bwd: mixture needs choice and pi");
  return vts_ew_reparam_kl_bwd(family, dz, dklpart, mean, logvar, scale, shape, eps, u, choice, pi, threshold, dmean,
                               dlogvar, dscale, dshape, B, (int)L, reinterpret_cast<hipStream_t>(stream));
}

int vts_elbo_recon_fwd(int family, const float* recon, const float* y, const float* klpart, int64_t nklpart,
                       const float* pi, float threshold, float beta, float* state, void* workspace,
                       size_t workspace_bytes, int64_t B, int64_t fs, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && recon && y && klpart && nklpart > 0 && state && workspace && B > 0 &&
                  fs > 0 && L > 0,
              "elbo_recon_fwd: bad arguments");
  VTS_REQUIRE(nklpart == vts_reparam_kl_blocks(B, L), "elbo_recon_fwd: nklpart is not vts_reparam_kl_blocks(B, L)");
  VTS_REQUIRE(family != VTS_MIXTURE || pi, "elbo_recon_fwd: mixture needs pi");
  VTS_REQUIRE(workspace_bytes >= vts_elbo_workspace_bytes(B, fs, L), "elbo_recon_fwd: workspace too small");
  return vts_ew_elbo_recon_fwd(family, recon, y, klpart, (int)nklpart, pi, threshold, beta, B, (int)fs, (int)L,
                               reinterpret_cast<float*>(workspace), state, reinterpret_cast<hipStream_t>(stream));
}

int vts_elbo_sums(int family, const float* recon, const float* y, const float* mean, const float* logvar,
                  const float* scale, const float* shape, const float* klpart, int64_t nklpart, float threshold,
                  float* sums, void* workspace, size_t workspace_bytes, int64_t B, int64_t fs, int64_t L,
                  vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 3 && recon && y && sums && workspace && B > 0 && fs > 0 && L > 0,
              "elbo_sums: bad arguments");
  if (klpart) {
    VTS_REQUIRE(family <= 2 && nklpart == vts_reparam_kl_blocks(B, L), "elbo_sums: nklpart is not vts_reparam_kl_blocks(B, L)");
  } else {
    VTS_REQUIRE(family == VTS_GPD || (mean && logvar), "elbo_sums: gaussian inputs missing");
    VTS_REQUIRE(family == VTS_GAUSSIAN || (scale && shape), "elbo_sums: gpd inputs missing");
  }
  VTS_REQUIRE(B < (1 << 24), "elbo_sums: the row count travels as a float (B < 2^24)");
  VTS_REQUIRE(workspace_bytes >= vts_elbo_workspace_bytes(B, fs, L), "elbo_sums: workspace too small");
  return vts_ew_elbo_sums(family, recon, y, mean, logvar, scale, shape, klpart, (int)nklpart, threshold, B, (int)fs,
                          (int)L, reinterpret_cast<float*>(workspace), sums, reinterpret_cast<hipStream_t>(stream));
}

int vts_elbo_from_sums(int family, const float* sums, const float* pi, float beta, float n_ranks, float* state,
                       int64_t fs, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 3 && sums && state && fs > 0 && L > 0, "elbo_from_sums: bad arguments");
  VTS_REQUIRE(family < VTS_MIXTURE || pi, "elbo_from_sums: mixture needs pi");
  return vts_ew_elbo_from_sums(family, sums, pi, beta, n_ranks, (int)fs, (int)L, state,
                               reinterpret_cast<hipStream_t>(stream));
}

int vts_elbo_recon_bwd(int family, const float* recon, const float* y, const float* state, const float* grad_out,
                       float threshold, float* drecon, float* dklpart, int64_t nklpart, float* dpi, int64_t B,
                       int64_t fs, vts_stream_t stream) {
  VTS_REQUIRE(family >= 0 && family <= 2 && recon && y && state && drecon && dklpart && nklpart > 0 && B > 0 &&
                  fs > 0,
              "elbo_recon_bwd: bad arguments");
  return vts_ew_elbo_recon_bwd(family, recon, y, state, grad_out, threshold, B, (int)fs, drecon, dklpart,
                               (int)nklpart, dpi, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------- feed-forward block
// The fused kernels work on whole 128-token tiles and store without bounds branches: the [tokens, F] buffers
// they write (hidden, mask, and dpre inside the workspace) carry rows up to the next multiple of 128.
static int64_t ffn_rows(int64_t M) { return (M + 127) / 128 * 128; }

struct FfnWs {
  float* dpre;       // [M][F] gradient w.r.t. linear1's pre-activation
  void* lin2;        // vts_linear_bwd workspace of linear2 (M, d, F)
  void* lin1;        // ... of linear1 (M, F, d)
  float* w1t;        // [d][F] and [F][d]: the weights transposed for the fused backward kernel
  float* w2t;
  size_t lin2_bytes, lin1_bytes, total;
};

static FfnWs ffn_ws_layout(void* base, int64_t M, int64_t d, int64_t F) {
  FfnWs w{};
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  w.dpre = reinterpret_cast<float*>(p + off); off += round256((size_t)ffn_rows(M) * F * sizeof(float));
  w.lin2_bytes = lin_ws_layout(nullptr, M, d, F).total;
  w.lin2 = p + off; off += w.lin2_bytes;
  w.lin1_bytes = lin_ws_layout(nullptr, M, F, d).total;
  w.lin1 = p + off; off += w.lin1_bytes;
  w.w1t = reinterpret_cast<float*>(p + off); off += round256((size_t)d * F * sizeof(float));
  w.w2t = reinterpret_cast<float*>(p + off); off += round256((size_t)d * F * sizeof(float));
  w.total = off;
  return w;
}

size_t vts_ffn_hidden_bytes(int64_t M, int64_t F) { return (size_t)ffn_rows(M) * F * sizeof(float); }
// [mask words, rows padded][the forward chain kernel's weight image: kept for the recomputing backward pass]
static size_t ffn_mask_words_bytes(int64_t M, int64_t F) { return round256((size_t)ffn_rows(M) * (F / 64) * 8); }
size_t vts_ffn_mask_bytes(int64_t M, int64_t F) {
  return (F % 64 == 0) ? ffn_mask_words_bytes(M, F) + round256(vts_ffn_image_bytes((long)F)) : 0;
}
size_t vts_ffn_bwd_workspace_bytes(int64_t M, int64_t d, int64_t F) { return ffn_ws_layout(nullptr, M, d, F).total; }

int vts_ffn_fwd(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                const float* keep, float drop_p, uint64_t seed, float* hidden, void* mask, float* y, int64_t M,
                int64_t d, int64_t F, int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(x && w1 && w2 && y && M > 0 && d > 0 && F > 0 && ldx >= d, "ffn_fwd: bad arguments");
  VTS_REQUIRE(hidden || (mask && b1 && b2 && vts_aligned16(mask) && (!keep || vts_aligned16(keep)) &&
                         vts_ffn_recompute_supported(x, ldx, M, d, F, dtype_mode)),
              "ffn_fwd: hidden may be NULL only with a 16-byte aligned mask (and keep) and where vts_ffn_recompute_supported() says so");
  VTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ffn_fwd: drop_p %g outside [0, 1)", (double)drop_p);
  VTS_REQUIRE(fits_int(M) && fits_int(d) && fits_int(F), "ffn_fwd: dimension too large");
  VTS_REQUIRE(!mask || (F % 64 == 0 && (reinterpret_cast<uintptr_t>(mask) & 7) == 0),
              "ffn_fwd: the packed mask needs F %% 64 == 0 and 8-byte alignment");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool hashed = !keep && drop_p > 0.f && seed != 0;
  const bool dropped = keep || hashed;
  if (!hidden)       // nothing but the mask is kept: the chunk stays in registers (vts_ffn_bwd_recompute is the way back)
    return vts_ffn_regs_fwd(x, ldx, w1, b1, w2, b2, keep, seed, hashed ? vts_drop_thresh(drop_p) : 0u,
                            dropped ? 1.f / (1.f - drop_p) : 1.f, reinterpret_cast<unsigned long long*>(mask),
                            reinterpret_cast<char*>(mask) + ffn_mask_words_bytes(M, F), y, (int)M, (int)F, st);
  // (a caller that keeps hidden may still go back through vts_ffn_bwd_recompute: leave the image behind the mask words)
  if (mask && b1 && b2 && vts_aligned16(mask) && vts_ffn_recompute_supported(x, ldx, M, d, F, dtype_mode))
    VTS_RC(vts_ffn_pack_fwd_image(w1, b1, w2, reinterpret_cast<char*>(mask) + ffn_mask_words_bytes(M, F), (int)F, st));
  if (linear_mode(dtype_mode, M, F, d) == 1 && b1 && b2 && vts_ffn_fused_supported(M, d, F, ldx, x, keep))
    return vts_ffn_fused_fwd(x, ldx, w1, b1, w2, b2, keep, seed, hashed ? vts_drop_thresh(drop_p) : 0u,
                             dropped ? 1.f / (1.f - drop_p) : 1.f, hidden, reinterpret_cast<unsigned long long*>(mask),
                             y, (int)M, (int)F, st);
  VTS_RC(vts_gemm_linear_fwd(x, ldx, w1, b1, hidden, F, (int)M, (int)F, (int)d, VTS_ACT_RELU_DROPOUT, keep,
                             dropped ? 1.f / (1.f - drop_p) : 1.f, seed, hashed ? vts_drop_thresh(drop_p) : 0u, nullptr,
                             reinterpret_cast<unsigned long long*>(mask), linear_mode(dtype_mode, M, F, d), st));
  return vts_gemm_linear_fwd(hidden, F, w2, b2, y, d, (int)M, (int)d, (int)F, VTS_ACT_NONE, nullptr, 1.f, 0ull, 0u,
                             nullptr, nullptr, linear_mode(dtype_mode, M, d, F), st);
}

int vts_ffn_bwd(const float* dy, const float* x, int64_t ldx, const float* w1, const float* w2, const float* hidden,
                const void* mask, float drop_scale, float* dx, int64_t lddx, float* dw1, float* db1, float* dw2,
                float* db2, void* workspace, size_t workspace_bytes, int64_t M, int64_t d, int64_t F, int dtype_mode,
                vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(dy && x && w1 && w2 && hidden && dw1 && db1 && dw2 && db2 && workspace && M > 0 && d > 0 && F > 0,
              "ffn_bwd: bad arguments");
  VTS_REQUIRE(!mask || F % 64 == 0, "ffn_bwd: the packed mask needs F %% 64 == 0");
  const FfnWs ws = ffn_ws_layout(workspace, M, d, F);
  VTS_REQUIRE(workspace_bytes >= ws.total, "ffn_bwd: workspace %zu B < required %zu B", workspace_bytes, ws.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const LinWs l2 = lin_ws_layout(ws.lin2, M, d, F);
  const LinWs l1 = lin_ws_layout(ws.lin1, M, F, d);
  const int mode2 = linear_mode(dtype_mode, M, d, F), mode1 = linear_mode(dtype_mode, M, F, d);
  if (mode2 == 1 && vts_ffn_fused_supported(M, d, F, d, dy, nullptr) && (!dx || (lddx % 4) == 0)) {
    // one kernel: dpre = mask * (dy . W2) written once, dx = dpre . W1 from the same registers, db1 partials
    float* dxbuf = dx ? dx : ws.dpre;   // dx not wanted: the tile is still produced, into scratch that is overwritten next
    VTS_REQUIRE(dx != nullptr, "ffn_bwd: the fused path produces dx; pass a buffer");
    VTS_RC(vts_ffn_fused_bwd_dx(dy, w1, w2, ws.w1t, ws.w2t, hidden, reinterpret_cast<const unsigned long long*>(mask),
                                drop_scale, ws.dpre, dxbuf, lddx, l2.dxpart, (int)M, (int)F, st));
    VTS_RC(vts_colsum_partials(l2.dxpart, db1, (int)((M + 127) / 128), (int)F, F, 0, st));
    VTS_RC(vts_gemm_linear_dw(dy, d, hidden, F, dw2, l2.slabs, l2.ksplit, (int)M, (int)d, (int)F, 0, mode2, st));
    VTS_RC(vts_ew_colsum_rows(dy, d, (int)M, (int)d, l2.colpart, st));
    VTS_RC(vts_colsum_partials(l2.colpart, db2, vts_ew_colsum_chunks((int)M), (int)d, d, 0, st));
    return vts_gemm_linear_dw(ws.dpre, F, x, ldx, dw1, l1.slabs, l1.ksplit, (int)M, (int)F, (int)d, 0, mode1, st);
  }
  // linear2: the data gradient lands as linear1's pre-activation gradient (relu and dropout derivative in the
  // epilogue, from the packed mask or from hidden != 0) together with its column sums = db1
  VTS_RC(vts_gemm_linear_dx(dy, d, w2, ws.dpre, F, (int)M, (int)d, (int)F, 0, hidden, F,
                            reinterpret_cast<const unsigned long long*>(mask), drop_scale, l2.dxpart, mode2, st));
  VTS_RC(vts_colsum_partials(l2.dxpart, db1, (int)(2 * ((M + 127) / 128)), (int)F, F, 0, st));
  VTS_RC(vts_gemm_linear_dw(dy, d, hidden, F, dw2, l2.slabs, l2.ksplit, (int)M, (int)d, (int)F, 0, mode2, st));
  VTS_RC(vts_ew_colsum_rows(dy, d, (int)M, (int)d, l2.colpart, st));
  VTS_RC(vts_colsum_partials(l2.colpart, db2, vts_ew_colsum_chunks((int)M), (int)d, d, 0, st));
  // linear1
  if (dx)
    VTS_RC(vts_gemm_linear_dx(ws.dpre, F, w1, dx, lddx, (int)M, (int)F, (int)d, 0, nullptr, 0, nullptr, 1.f, nullptr,
                              mode1, st));
  return vts_gemm_linear_dw(ws.dpre, F, x, ldx, dw1, l1.slabs, l1.ksplit, (int)M, (int)F, (int)d, 0, mode1, st);
}

// ---- backward of the block from x, dy and the packed mask alone (no [tokens, F] tensor in memory)
struct FfnRcWs {
  char* image;       // the backward chain kernel's weight image
  float* colpart1;   // [slabs][F] column sums of dpre
  float* colpart2;   // column-sum partials of dy
  float* slabs;      // per-slab partial weight gradients
  size_t total;
};
static FfnRcWs ffn_rc_layout(void* base, int64_t M, int64_t d, int64_t F) {
  FfnRcWs w{};
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  w.image = p + off; off += round256(vts_ffn_image_bytes((long)F));
  w.colpart1 = reinterpret_cast<float*>(p + off); off += round256((size_t)vts_ffn_wgrad_recompute_nslabs(M, F) * F * sizeof(float));
  w.colpart2 = reinterpret_cast<float*>(p + off); off += round256((size_t)vts_ew_colsum_chunks((int)M) * d * sizeof(float));
  w.slabs = reinterpret_cast<float*>(p + off); off += round256(vts_ffn_wgrad_recompute_slab_floats(M, F) * sizeof(float));
  w.total = off;
  return w;
}
int vts_ffn_recompute_supported(const float* x, int64_t ldx, int64_t M, int64_t d, int64_t F, int dtype_mode) {
  if (dtype_mode != 1 || M <= 0 || d <= 0 || F <= 0 || !fits_int(M) || !fits_int(F)) return 0;
  if (linear_mode(dtype_mode, M, d, F) != 1 || linear_mode(dtype_mode, M, F, d) != 1) return 0;
  return (vts_ffn_fused_supported(M, d, F, ldx, x, nullptr) && vts_ffn_wgrad_recompute_supported(M, d, F, ldx, x)) ? 1 : 0;
}
size_t vts_ffn_bwd_recompute_workspace_bytes(int64_t M, int64_t d, int64_t F) { return ffn_rc_layout(nullptr, M, d, F).total; }

int vts_ffn_bwd_recompute(const float* dy, const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2,
                          const void* mask, float drop_scale, float* dx, int64_t lddx, float* dw1, float* db1, float* dw2,
                          float* db2, void* workspace, size_t workspace_bytes, int64_t M, int64_t d, int64_t F,
                          int dtype_mode, vts_stream_t stream) {
  VTS_MODE_CHECK(dtype_mode);
  VTS_REQUIRE(dy && x && w1 && b1 && w2 && mask && dx && dw1 && db1 && dw2 && db2 && workspace,
              "ffn_bwd_recompute: null pointer");
  (void)b1;   // (its values travel in the forward image behind the mask words)
  VTS_REQUIRE(vts_ffn_recompute_supported(x, ldx, M, d, F, dtype_mode), "ffn_bwd_recompute: not planned for M %ld, d %ld, F %ld in mode %d (see vts_ffn_recompute_supported)",
              (long)M, (long)d, (long)F, dtype_mode);
  VTS_REQUIRE(vts_aligned16(dy) && vts_aligned16(w1) && vts_aligned16(mask) && vts_aligned16(workspace) && (lddx % 4) == 0,
              "ffn_bwd_recompute: dy, w1, mask and the workspace must be 16-byte aligned, lddx a multiple of 4");
  const FfnRcWs ws = ffn_rc_layout(workspace, M, d, F);
  VTS_REQUIRE(workspace_bytes >= ws.total, "ffn_bwd_recompute: workspace %zu B < required %zu B", workspace_bytes, ws.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned long long* bits = reinterpret_cast<const unsigned long long*>(mask);
  const char* image_fwd = reinterpret_cast<const char*>(mask) + ffn_mask_words_bytes(M, F);   // left there by vts_ffn_fwd
  // dx from one pass over the chunks, dpre in registers; then both weight gradients and db1 from the recomputed tiles
  VTS_RC(vts_ffn_regs_bwd_dx(dy, w1, w2, bits, drop_scale, ws.image, dx, lddx, (int)M, (int)F, st));
  VTS_RC(vts_ew_colsum_rows(dy, d, (int)M, (int)d, ws.colpart2, st));
  VTS_RC(vts_colsum_partials(ws.colpart2, db2, vts_ew_colsum_chunks((int)M), (int)d, d, 0, st));
  VTS_RC(vts_ffn_wgrad_recompute(x, ldx, dy, image_fwd, ws.image, bits, drop_scale, ws.slabs, ws.colpart1, dw1, dw2, (int)M,
                                 (int)F, st));
  return vts_colsum_partials(ws.colpart1, db1, vts_ffn_wgrad_recompute_nslabs(M, F), (int)F, F, 0, st);
}

// ------------------------------------------------------------------- transformer encoder pieces
static int mha_check(const char* what, int64_t B, int64_t T, int64_t Tq, int64_t nhead, int64_t dh, int64_t ldq,
                     int64_t ldkv, int64_t ldo, float drop_p) {
  VTS_REQUIRE(B > 0 && T > 0 && Tq > 0 && nhead > 0 && dh > 0, "%s: bad dimensions", what);
  VTS_REQUIRE(fits_int(B * nhead) && fits_int(T) && fits_int(Tq), "%s: dimension too large", what);
  VTS_REQUIRE(ldq >= nhead * dh && ldkv >= nhead * dh && ldo >= nhead * dh, "%s: row stride below nhead*dh", what);
  VTS_REQUIRE((ldq % 4) == 0 && (ldkv % 4) == 0 && (ldo % 4) == 0, "%s: row strides must be multiples of 4 floats", what);
  VTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: drop_p %g outside [0, 1)", what, (double)drop_p);
  return VTS_OK;
}

int vts_mha_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, int64_t ldo,
                float* lse, const uint8_t* keep, float drop_p, uint64_t seed, int64_t B, int64_t T, int64_t Tq,
                int64_t nhead, int64_t dh, vts_stream_t stream) {
  VTS_REQUIRE(q && k && v && out && lse, "mha_fwd: null pointer");
  VTS_RC(mha_check("mha_fwd", B, T, Tq, nhead, dh, ldq, ldkv, ldo, drop_p));
  VTS_REQUIRE(vts_aligned16(q) && vts_aligned16(k) && vts_aligned16(v) && vts_aligned16(out),
              "mha_fwd: q, k, v, out must be 16-byte aligned");
  return vts_attn_mha(0, q, ldq, k, v, ldkv, out, ldo, lse, keep, drop_p, seed, nullptr, 0, nullptr, 0, nullptr,
                      nullptr, 0, (int)B, (int)T, (int)Tq, (int)nhead, (int)dh, reinterpret_cast<hipStream_t>(stream));
}

int vts_mha_bwd(const float* dout, int64_t lddo, const float* q, int64_t ldq, const float* k, const float* v,
                int64_t ldkv, const float* out, int64_t ldo, const float* lse, const uint8_t* keep, float drop_p,
                uint64_t seed, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv, int64_t B, int64_t T,
                int64_t Tq, int64_t nhead, int64_t dh, vts_stream_t stream) {
  VTS_REQUIRE(dout && q && k && v && out && lse && dq && dk && dv, "mha_bwd: null pointer");
  VTS_RC(mha_check("mha_bwd", B, T, Tq, nhead, dh, ldq, ldkv, ldo, drop_p));
  VTS_REQUIRE(lddo >= nhead * dh && lddq >= nhead * dh && lddkv >= nhead * dh && (lddo % 4) == 0 && (lddq % 4) == 0 &&
                  (lddkv % 4) == 0,
              "mha_bwd: gradient row strides must be multiples of 4 floats and at least nhead*dh");
  VTS_REQUIRE(vts_aligned16(dout) && vts_aligned16(q) && vts_aligned16(k) && vts_aligned16(v) && vts_aligned16(out) &&
                  vts_aligned16(dq) && vts_aligned16(dk) && vts_aligned16(dv),
              "mha_bwd: tensors must be 16-byte aligned");
  return vts_attn_mha(1, q, ldq, k, v, ldkv, const_cast<float*>(out), ldo, const_cast<float*>(lse), keep, drop_p, seed,
                      dout, lddo, dq, lddq, dk, dv, lddkv, (int)B, (int)T, (int)Tq, (int)nhead, (int)dh,
                      reinterpret_cast<hipStream_t>(stream));
}

int vts_dropout_add_fwd(const float* residual, const float* x, const uint8_t* keep, float drop_p, uint64_t seed,
                        float* out, int64_t n, vts_stream_t stream) {
  VTS_REQUIRE(x && out && n > 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_add_fwd: bad arguments");
  return vts_attn_dropout_add(residual, x, keep, drop_p, seed, out, n, reinterpret_cast<hipStream_t>(stream));
}

int vts_dropout_bwd(const float* dout, const uint8_t* keep, float drop_p, uint64_t seed, float* dx, int64_t n,
                    vts_stream_t stream) {
  VTS_REQUIRE(dout && dx && n > 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_bwd: bad arguments");
  return vts_attn_dropout_add(nullptr, dout, keep, drop_p, seed, dx, n, reinterpret_cast<hipStream_t>(stream));
}

int vts_dropout_mask(uint64_t seed, float drop_p, uint8_t* keep, int64_t n, vts_stream_t stream) {
  VTS_REQUIRE(keep && n > 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_mask: bad arguments");
  return vts_attn_dropout_mask(seed, drop_p, keep, n, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------ Adam
int vts_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                  float lr, float beta1, float beta2, float eps, float grad_scale, vts_stream_t stream) {
  VTS_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  return vts_ew_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2),
                     grad_scale, reinterpret_cast<hipStream_t>(stream));
}

int vts_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                          float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* guard,
                          int64_t n_guard, vts_stream_t stream) {
  VTS_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step_guarded: bad arguments");
  VTS_REQUIRE(guard && n_guard >= 1 && n_guard <= 64 && (reinterpret_cast<uintptr_t>(guard) & 3u) == 0,
              "adam_step_guarded: 1 to 64 guard words, 4-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  return vts_ew_adam_guarded(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2),
                             grad_scale, reinterpret_cast<const int*>(guard), (int)n_guard, reinterpret_cast<hipStream_t>(stream));
}

int vts_adam_step_counted(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          int64_t* step_state, float lr, float beta1, float beta2, float eps, float grad_scale,
                          vts_stream_t stream) {
  VTS_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_state && n > 0, "adam_step_counted: bad arguments");
  VTS_REQUIRE((reinterpret_cast<uintptr_t>(step_state) & 7u) == 0, "adam_step_counted: step_state must be 8-byte aligned");
  return vts_ew_adam_counted(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                             reinterpret_cast<long long*>(step_state), grad_scale, reinterpret_cast<hipStream_t>(stream));
}

int vts_scale(const float* in, const float* scalar, float* out, int64_t n, vts_stream_t stream) {
  VTS_REQUIRE(in && scalar && out && n > 0, "scale: bad arguments");
  return vts_ew_scale(in, scalar, out, n, reinterpret_cast<hipStream_t>(stream));
}

int vts_gather_chunks(const float* const* src, const int64_t* numel, const int64_t* dst_offset, int64_t count,
                      float* dst, vts_stream_t stream) {
  VTS_REQUIRE(src && numel && dst_offset && dst && count >= 0, "gather_chunks: bad arguments");
  for (int64_t i = 0; i < count; ++i)
    VTS_REQUIRE(numel[i] >= 0 && dst_offset[i] >= 0, "gather_chunks: negative size or offset in chunk %lld", (long long)i);
  if (count == 0) return VTS_OK;
  return vts_ew_gather_chunks(src, reinterpret_cast<const long long*>(numel), reinterpret_cast<const long long*>(dst_offset),
                              (long)count, dst, reinterpret_cast<hipStream_t>(stream));
}

// ---- fused head / decoder chains (mlp_chain.hip); vts_mlp_chain and vts_mlp_chain_desc have the same layout
int vts_mlp_chain_supported(const vts_mlp_chain* chain, int64_t B) {
  return vts_mlp_chain_supported_impl(reinterpret_cast<const vts_mlp_chain_desc*>(chain), (long)B);
}
int vts_mlp_chain_fwd(const vts_mlp_chain* chain, const float* x, int64_t ldx, int64_t B, const float* const* w,
                      const float* const* b, const float* const* keep, float* const* trunk_out, float* const* head_out,
                      float* const* head_aux, vts_stream_t stream) {
  VTS_REQUIRE(chain != nullptr && w != nullptr, "vts_mlp_chain_fwd: missing argument");
  return vts_mlp_chain_fwd_impl(reinterpret_cast<const vts_mlp_chain_desc*>(chain), x, (long)ldx, (long)B, w, b, keep,
                                trunk_out, head_out, head_aux, (hipStream_t)stream);
}
size_t vts_mlp_chain_bwd_workspace_bytes(const vts_mlp_chain* chain, int64_t B) {
  return chain ? vts_mlp_chain_bwd_workspace_bytes_impl(reinterpret_cast<const vts_mlp_chain_desc*>(chain), (long)B) : 0;
}
int vts_mlp_chain_bwd(const vts_mlp_chain* chain, const float* x, int64_t ldx, int64_t B, const float* const* w,
                      const float* const* keep, float* const* trunk_out, float* const* head_out,
                      float* const* head_aux, const float* const* d_head, float* dx, int64_t lddx, float* const* dw,
                      float* const* db, void* workspace, size_t workspace_bytes, vts_stream_t stream) {
  VTS_REQUIRE(chain != nullptr && w != nullptr, "vts_mlp_chain_bwd: missing argument");
  VTS_REQUIRE(dx == nullptr || lddx >= chain->in_dim, "vts_mlp_chain_bwd: lddx too small");
  return vts_mlp_chain_bwd_impl(reinterpret_cast<const vts_mlp_chain_desc*>(chain), x, (long)ldx, (long)B, w, keep,
                                trunk_out, head_out, head_aux, d_head, dx, (long)lddx, dw, db, workspace, workspace_bytes,
                                (hipStream_t)stream);
}

}  // extern "C"
