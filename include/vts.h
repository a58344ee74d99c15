/* vts: C ABI of the MI355X (gfx950) LSTM-VAE train-step kernels.
 *
 * The reference (letrucnhi160703/vae-time-series) has no FFI: its hot path is the set of torch
 * call sites cited beside each entry point below (paths relative to vae-models/).  This header
 * is what a binding for that path binds: plain pointers and sizes, no torch types.
 *
 * Conventions
 *  - every pointer is DEVICE memory owned by the caller (fp32 unless noted), contiguous,
 *    16-byte aligned where stated; the library never allocates, frees or keeps device memory;
 *  - all work is enqueued on `stream`; no call synchronises, so calls compose with the caller's
 *    stream ordering and can be made from any host thread;
 *  - return 0 on success, otherwise VTS_ERR_* with a message in a thread-local buffer
 *    (vts_last_error()); no C++ exception crosses the boundary;
 *  - `dtype_mode`: 0 = fp32 operands on the f32 matrix cores (v_mfma_f32_16x16x4_f32);
 *    1 = "bf16x3": each fp32 operand is split into a bf16 head and a bf16 tail while it is staged,
 *    the product is head*head + head*tail + tail*head on the bf16 matrix cores with fp32
 *    accumulation (error ~2^-16 per product, inside the 1e-4 parity bound).  The recurrent
 *    entry points follow the mode for H >= 64 and use the fp32 kernels below that.  Other values
 *    are rejected.
 *
 * Layouts
 *  - sequences are batch-first row-major [B, T, F] exactly as the reference feeds nn.LSTM;
 *  - "tile16" buffers (gate activations, cell states, sequence gradients) are private to the
 *    LSTM entry points: a [Bp*T, N] matrix (Bp = B rounded up to 16) with logical row
 *    (b/16 * T + t) * 16 + b%16, cut into 16x16 blocks stored in MFMA accumulator order.  Callers
 *    only size them (vts_lstm_*_bytes) and pass them back unchanged from forward to backward;
 *    vts_tile16_convert() translates to/from row-major for callers that need the contents.
 */
#ifndef VTS_H
#define VTS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vts_stream_t; /* == hipStream_t */

#define VTS_OK 0
#define VTS_ERR_ARG 1
#define VTS_ERR_HIP 2
#define VTS_ERR_UNSUPPORTED 3

/* activation codes of vts_linear_* */
#define VTS_ACT_NONE 0
#define VTS_ACT_RELU 1
#define VTS_ACT_EXP 2          /* torch.exp(head)              gpd-lstm-DE.py:31, mixture-vae-DE.py:67 */
#define VTS_ACT_CLAMP_MIN 3    /* torch.clamp(head, min=1e-6)  gpd-lstm-DE.py:32, mixture-vae-DE.py:68 */
#define VTS_ACT_RELU_DROPOUT 4 /* dropout(relu(.)) with the keep mask fed in: gpd-lstm-DE.py:30,46 */

/* latent families of vts_reparam_* / vts_elbo_* */
#define VTS_GAUSSIAN 0
#define VTS_GPD 1
#define VTS_MIXTURE 2
#define VTS_MIXTURE_SPLIT 3    /* vts_elbo_* only: the transformer variant's loss, mixture-vae-transformer.py:100-113 */

const char* vts_version(void);
const char* vts_last_error(void);

/* ---- LSTM layer: nn.LSTM(input, H, num_layers, batch_first=True) with zero initial state ----
 * replaces gaussian-lstm-DE.py:16,28 ; gpd-lstm-DE.py:18,27 ; mixture-vae-DE.py:45,60 (one call
 * per stacked layer; layer k>0 takes layer k-1's hseq as x).  Gate order i,f,g,o; weights in
 * torch's shapes: w_ih [4H, Cin], w_hh [4H, H], b_ih/b_hh [4H].  H in {16, 32, 64, 128, 256}. */
size_t vts_lstm_gates_bytes(int64_t B, int64_t T, int64_t H);      /* tile16 [Bp*T, 4H] */
size_t vts_lstm_cells_bytes(int64_t B, int64_t T, int64_t H);      /* tile16 [Bp*T, H]  */
size_t vts_lstm_seqgrad_bytes(int64_t B, int64_t T, int64_t F);    /* tile16 [Bp*T, F]  */
size_t vts_lstm_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H);

/* forward: writes gates (activated i,f,g,o), cells (c_t), hseq [B,T,H]; h_last / c_last [B,H]
 * (the layer's slice of h_n / c_n) are optional. */
int vts_lstm_layer_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                       const float* b_hh, float* gates, float* cells, float* hseq, float* h_last,
                       float* c_last,
                       void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin,
                       int64_t H, int dtype_mode, vts_stream_t stream);

/* backward: consumes gates (overwritten with the gate pre-activation gradients).
 * Upstream gradients, each optional (NULL = zero): dh_seq_t16 tile16 [Bp*T, H] (what the layer
 * above wrote into its dx_t16), dh_last [B,H] and dc_last [B,H] row-major.
 * Outputs: dx_t16 tile16 [Bp*T, Cin] (NULL for the first layer: the input needs no gradient),
 * dw_ih, dw_hh, db_ih, db_hh in torch's shapes; accumulate != 0 adds into them. */
int vts_lstm_layer_bwd(const float* x, const float* hseq, const float* w_ih, const float* w_hh,
                       float* gates, const float* cells, const float* dh_seq_t16,
                       const float* dh_last, const float* dc_last, float* dx_t16, float* dw_ih,
                       float* dw_hh, float* db_ih, float* db_hh, int accumulate, void* workspace,
                       size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin, int64_t H,
                       int dtype_mode, vts_stream_t stream);

/* The stages of a layer as separate entry points (same buffers and conventions as above), for
 * callers that schedule or time them individually.  layer_fwd = inproj_fwd + recurrent_fwd;
 * layer_bwd = recurrent_bwd + dgrad + wgrad(x, tshift 0) + wgrad(hseq, tshift -1). */
size_t vts_lstm_recurrent_workspace_bytes(int64_t B, int64_t T, int64_t H);
size_t vts_lstm_wgrad_workspace_bytes(int64_t B, int64_t T, int64_t F, int64_t H);
/* gates <- x . w_ih^T + b_ih + b_hh for every step at once (one GEMM over B*T rows) */
int vts_lstm_inproj_fwd(const float* x, const float* w_ih, const float* b_ih, const float* b_hh,
                        float* gates, int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode,
                        vts_stream_t stream);
/* persistent recurrence: gates (pre-activations in, activations out), cells, hseq, h_last, c_last */
int vts_lstm_recurrent_fwd(float* gates, float* cells, float* hseq, float* h_last, float* c_last,
                           const float* w_hh, void* workspace, size_t workspace_bytes, int64_t B,
                           int64_t T, int64_t H, int dtype_mode, vts_stream_t stream);
/* Status of the last vts_lstm_recurrent_fwd / _fwd_planes / vts_lstm_layer_fwd launch that used `workspace`, copied
 * (device to device, on `stream`, behind that launch) into the device word *status: 0 = completed; 1 = the
 * bench-sized forward kernel, whose workgroups hand h_t to each other, gave up waiting for a peer: its h sequence
 * and final states then hold NaN.  (No counterpart in the reference: torch raises on the host.) */
int vts_lstm_recurrent_fwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H,
                                  int32_t* status, vts_stream_t stream);
/* reverse recurrence: gates (activations in, pre-activation gradients out), bias gradients */
int vts_lstm_recurrent_bwd(float* gates, const float* cells, const float* dh_seq_t16,
                           const float* dh_last, const float* dc_last, const float* w_hh,
                           float* db_ih, float* db_hh, int accumulate, void* workspace,
                           size_t workspace_bytes, int64_t B, int64_t T, int64_t H, int dtype_mode,
                           vts_stream_t stream);
/* Status of the last vts_lstm_recurrent_bwd / _bwd_planes / vts_lstm_layer_bwd launch that used `workspace`, as
 * vts_lstm_recurrent_fwd_status: 0 = completed; 2 = a workgroup of the bench-sized backward kernel gave up waiting
 * for its peer's partial sums: the gate gradients and bias gradients then hold NaN.  (No counterpart in the
 * reference.) */
int vts_lstm_recurrent_bwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t H,
                                  int32_t* status, vts_stream_t stream);
/* dx_t16 <- dgates . w_ih (gradient w.r.t. the layer input sequence; Cin % 16 == 0) */
int vts_lstm_dgrad(const float* dgates, const float* w_ih, float* dx_t16, int64_t B, int64_t T,
                   int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream);
/* dw[4H,F] (+)= sum over (b,t) of dgates[b,t,:]^T x[b,t+tshift,:F]; tshift 0 for w_ih (x = layer
 * input), -1 for w_hh (x = hseq; step -1 is the zero initial state) */
int vts_lstm_wgrad(const float* dgates, const float* x, int tshift, float* dw, int accumulate,
                   void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t F, int64_t H,
                   int dtype_mode, vts_stream_t stream);

/* Both weight gradients of a layer in one launch, for the shapes it is planned for (vts_lstm_wgrad_pair_supported
 * != 0: bf16x3 mode, H = 256, F >= 256, a contraction long enough for the 256 x 128 split-K kernel; columns of x
 * past the first 256 take a launch of their own):
 * dw_ih[4H,F] (+)= dgates^T x (tshift 0) and dw_hh[4H,H] (+)= dgates^T hseq (tshift -1), dgates read by one
 * launch.  Same results as two vts_lstm_wgrad calls up to the order of the fp32 partial sums; same workspace
 * (vts_lstm_wgrad_workspace_bytes(B, T, F, H)).  Other shapes: VTS_ERR_UNSUPPORTED.
 * (autograd of nn.LSTM in the reference: vae-models/mixture-vae-DE.py:45,60,161) */
int vts_lstm_wgrad_pair_supported(int64_t B, int64_t T, int64_t F, int64_t H, int dtype_mode);
int vts_lstm_wgrad_pair(const float* dgates, const float* x, const float* hseq, float* dw_ih, float* dw_hh,
                        int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                        int64_t F, int64_t H, int dtype_mode, vts_stream_t stream);

/* ---- The layer's three big products over bf16 plane operands (bench-sized shapes, dtype_mode 1) ----
 * In bf16x3 mode every operand of a product is an fp32 value split into a bf16 head and a bf16 tail.  For the
 * shapes vts_lstm_planes_supported() accepts (hidden 256, B a multiple of 128, T even) the sequences can be kept
 * in that split form, as "planes": opaque buffers of 1 KiB tiles in matrix-core operand order, sized by
 * vts_lstm_planes_bytes(B, T, F, is_hseq).  The recurrent kernels write the h sequence and the gate gradients as
 * planes directly, the layer input is converted once (vts_lstm_planes_from_rows), and the input projection, data
 * gradient and both weight gradients then run without converting or transposing anything.  An h-sequence plane
 * buffer (is_hseq = 1) carries one more step slot, the zero initial state, so "h one step back" is an offset.
 * Same arithmetic as the dtype_mode 1 entry points above (three bf16 MFMAs per fp32 product); results differ from
 * them only in the order of the fp32 partial sums.  Workspace of every call: vts_lstm_planes_workspace_bytes().
 * (autograd of nn.LSTM in the reference: vae-models/mixture-vae-DE.py:45,60,161 ; gpd-lstm-DE.py:18,27,102) */
int vts_lstm_planes_supported(int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode);
size_t vts_lstm_planes_bytes(int64_t B, int64_t T, int64_t F, int is_hseq);
size_t vts_lstm_planes_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H);
/* x [B,T,Cin] row-major -> planes (features zero-padded to a multiple of 64) */
int vts_lstm_planes_from_rows(const float* x, void* x_planes, int64_t B, int64_t T, int64_t Cin, int64_t H,
                              int dtype_mode, vts_stream_t stream);
/* vts_lstm_inproj_fwd with the layer input as planes (x_is_hseq: it is the h-sequence buffer of the layer below) */
int vts_lstm_inproj_fwd_planes(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih,
                               const float* b_hh, float* gates, void* workspace, size_t workspace_bytes, int64_t B,
                               int64_t T, int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream);
/* vts_lstm_recurrent_fwd that also writes the h sequence as planes; hseq (fp32 [B,T,H]) may then be NULL */
int vts_lstm_recurrent_fwd_planes(float* gates, float* cells, float* hseq, void* h_planes, float* h_last, float* c_last,
                                  const float* w_hh, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                                  int64_t H, int dtype_mode, vts_stream_t stream);
/* vts_lstm_inproj_fwd_planes + vts_lstm_recurrent_fwd_planes as ONE kernel: the layer's forward pass with both weight
 * matrices stationary on chip and the gate pre-activations never written to memory (per step
 * gates_t = [x_t | h_{t-1}] . [W_ih | W_hh]^T + b_ih + b_hh; nn.LSTM's forward at vae-models/mixture-vae-DE.py:45,60 ;
 * gpd-lstm-DE.py:18,27 ; gaussian-lstm-DE.py:16,28).  `gates` and `cells` receive what the backward entry points
 * read (gate activations, cell states), h_planes the h sequence; hseq (fp32 [B,T,H]), h_last, c_last may be NULL.
 * Taken for the shapes vts_lstm_fused_fwd_supported() accepts: those of vts_lstm_planes_supported() whose input is
 * at most 64, 193..256 or 321..384 features wide and whose grid of B / 16 workgroups is resident at once; the
 * workgroups wait for each other, so the launch has a status word like vts_lstm_recurrent_fwd
 * (vts_lstm_fused_fwd_status: non-zero = a peer never showed up and the outputs are NaN).  Returns
 * VTS_ERR_UNSUPPORTED when the runtime cannot guarantee a resident grid (or the stream is recording a graph):
 * callers then run the two entry points above.  Same arithmetic as they (three bf16 MFMAs per fp32 product, fp32
 * accumulation, the bias added first instead of last); results agree to the order of the fp32 sums.
 * vts_lstm_fused_fwd_preferred() says whether callers that have the choice should take this entry point: 0 unless
 * VTS_LSTM_FUSED_FWD=1 was set at start-up or vts_debug_set_fused_fwd(1) was called - measured at 4096 x 96 windows
 * the one kernel is level with the two (1.55-1.7 ms against 1.53 ms for a 256-wide input, 2.0-2.3 ms against
 * 1.75 ms for a 370-wide one: the saved tensors' stores, 80 KiB per compute unit and step, hold up its only wave
 * per SIMD), so the Python face keeps the two-kernel path by default. */
int vts_lstm_fused_fwd_supported(int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode);
int vts_lstm_fused_fwd_preferred(void);
size_t vts_lstm_fused_fwd_workspace_bytes(int64_t B, int64_t T, int64_t Cin, int64_t H);
int vts_lstm_fused_fwd_planes(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih,
                              const float* b_hh, const float* w_hh, float* gates, float* cells, float* hseq,
                              void* h_planes, float* h_last, float* c_last, void* workspace, size_t workspace_bytes,
                              int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream);
int vts_lstm_fused_fwd_status(const void* workspace, size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin,
                              int64_t H, int32_t* status, vts_stream_t stream);
/* vts_lstm_recurrent_bwd that writes the gate pre-activation gradients as planes (vts_lstm_planes_bytes(B, T, 4H, 0))
 * and leaves `gates` untouched */
int vts_lstm_recurrent_bwd_planes(const float* gates, const float* cells, const float* dh_seq_t16, const float* dh_last,
                                  const float* dc_last, const float* w_hh, void* dg_planes, float* db_ih, float* db_hh,
                                  int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                                  int64_t H, int dtype_mode, vts_stream_t stream);
/* vts_lstm_dgrad from gate-gradient planes (Cin a multiple of 256) */
int vts_lstm_dgrad_planes(const void* dg_planes, const float* w_ih, float* dx_t16, void* workspace,
                          size_t workspace_bytes, int64_t B, int64_t T, int64_t Cin, int64_t H, int dtype_mode,
                          vts_stream_t stream);
/* vts_lstm_wgrad_pair from planes: dw_ih (+)= dgates^T x, dw_hh (+)= dgates^T h(one step back), one launch */
int vts_lstm_wgrad_planes(const void* dg_planes, const void* x_planes, int x_is_hseq, const void* h_planes, float* dw_ih,
                          float* dw_hh, int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t T,
                          int64_t Cin, int64_t H, int dtype_mode, vts_stream_t stream);

/* ---- Test hooks and development switches (vts_debug_*) -------------------------------------------------------
 * NOT part of the product interface: the only process-wide mutable state of the library, exported so that the
 * test suite can reach the time-out paths of the kernels whose workgroups wait for each other and can run the
 * alternative kernels of one shape against each other.  Every one of them has a default that a production caller
 * never needs to touch (the defaults are what the measurements in the documentation were taken with); none changes
 * results beyond the order of fp32 sums, except the *_fault hooks, which make a launch fail on purpose.  Changing
 * one while launches of other host threads are being enqueued affects those launches too. */
/* Test hook, process-wide: member `member` (0..7) of every cluster of the bench-sized forward recurrence withholds
 * its hand-off flags and the waits are cut short, so the timeout path above can be exercised; -1 = off (default). */
void vts_debug_set_cluster_fault(int member);
/* Development switch, process-wide: which forward cluster kernel runs the bench-sized forward recurrence.  on != 0:
 * one ring of eight WAVES per window tile (wave w of each of the cluster's eight workgroups; two waves per SIMD, no
 * workgroup barrier inside the time loop, every wave publishes and polls for its own 16 windows); on = 0: the
 * workgroup-wide hand-off (one wave per SIMD, one flag per workgroup and step).  Start-up value from
 * VTS_LSTM_CLUSTER_RINGS.  Same arithmetic in the same order: identical results. */
void vts_debug_set_cluster_rings(int on);
/* Development switch, process-wide: on != 0 lets vts_lstm_recurrent_bwd_planes run the bench-sized backward
 * recurrence as clusters of eight workgroups with W_hh stationary (the forward pass always does); off (default,
 * unless VTS_LSTM_CLUSTER_BWD=1 is set at start-up) it runs one workgroup per 16 windows, which measures 3 %
 * faster.  Same results either way up to the order of fp32 sums. */
void vts_debug_set_cluster_bwd(int on);
/* Development switch, process-wide: on = 0 runs the bf16x3 backward recurrence at hidden 256 on the kernel that
 * loads a step's saved gates, cells and dh straight into registers at the top of the step's product (the kernel the
 * other hidden sizes use); on != 0 (default, unless VTS_LSTM_STAGED_BWD=0 is set at start-up) on the kernel that
 * stages them through LDS a step ahead and keeps an eighth of the packed W_hh image in registers (5 % faster at
 * the bench shape).  Same arithmetic in the same order: identical results. */
void vts_debug_set_staged_bwd(int on);
/* Development switch, process-wide: on = 0 keeps vts_lstm_recurrent_bwd_planes off the bench-sized backward kernel in
 * which two workgroups share 32 windows, each holding the rows of W_hh of half the hidden units and handing the
 * other the partial sums for its half of dh every step (default on, unless VTS_LSTM_PAIR_BWD=0 is set at
 * start-up; taken for hidden 256, B a multiple of 32 and a grid that is resident at once).  Same results as the
 * one-workgroup kernels up to the order of the fp32 sums. */
void vts_debug_set_pair_bwd(int on);
/* Test hook, process-wide: member `member` (0..7) of every cluster of vts_lstm_fused_fwd_planes withholds its hand-off
 * flags and the waits are cut short (vts_lstm_fused_fwd_status then reads non-zero); -1 = off (default). */
void vts_debug_set_fused_fault(int member);
/* Development switch, process-wide: what vts_lstm_fused_fwd_preferred() answers (default 0, or 1 when
 * VTS_LSTM_FUSED_FWD=1 is set at start-up). */
void vts_debug_set_fused_fwd(int on);
/* Test hook, process-wide: member `member` (0 or 1) of every such pair withholds what it owes its peer and the
 * waits are cut short, so the timeout path (vts_lstm_recurrent_bwd_status) can be exercised; -1 = off (default). */
void vts_debug_set_pair_fault(int member);
/* ---- end of the test hooks ---- */

/* row-major [B,T,N] <-> tile16 [Bp*T,N]; to_rows = 0 packs (padding rows zeroed), 1 unpacks. */
int vts_tile16_convert(float* rows, float* tiled, int64_t B, int64_t T, int64_t N, int to_rows,
                       vts_stream_t stream);

/* ---- Linear (+ fused activation): nn.Linear / F.relu / nn.Dropout / exp / clamp call sites ----
 * gaussian-lstm-DE.py:30-31,43-45 ; gpd-lstm-DE.py:29-33,45-47 ; mixture-vae-DE.py:62-68,81-83.
 * y[M,N] = act(x[M,K] . w[N,K]^T + bias).  x may be a strided view (row stride ldx floats), which
 * is how h_n[-1] is read straight out of hseq.  keep: [M,N] 0/1 mask (VTS_ACT_RELU_DROPOUT,
 * NULL = eval mode), scaled by keep_scale = 1/(1-p).  aux: [M,N] written by VTS_ACT_CLAMP_MIN
 * (1 where the gradient passes). */
int vts_linear_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                   int64_t M, int64_t N, int64_t K, int act, const float* keep, float keep_scale,
                   float* aux, int dtype_mode, vts_stream_t stream);
size_t vts_linear_bwd_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* dy, y, aux: [M,N].  dx (row stride lddx, NULL = not needed; dx_accumulate adds), dw [N,K],
 * db [N] (param_accumulate adds). */
int vts_linear_bwd(const float* dy, const float* y, const float* aux, int act, float keep_scale,
                   const float* x, int64_t ldx, const float* w, float* dx, int64_t lddx,
                   int dx_accumulate, float* dw, float* db, int param_accumulate, void* workspace,
                   size_t workspace_bytes, int64_t M, int64_t N, int64_t K, int dtype_mode,
                   vts_stream_t stream);

/* act = VTS_ACT_RELU_DROPOUT with the mask drawn by the library's counter hash instead of fed in:
 * element (row, col) is kept iff its 16 bits of the hash (index row*N + col, below) are >= drop_p * 2^16, and scaled by 1/(1-drop_p);
 * seed 0 or drop_p 0 = eval mode (plain relu).  Backward is vts_linear_bwd with VTS_ACT_RELU_DROPOUT and
 * keep_scale 1/(1-drop_p): it needs y only, not the mask.  Replaces dropout(relu(linear1(x))) inside
 * nn.TransformerEncoderLayer (mixture-vae-transformer.py:39) without a [tokens, 2048] mask in memory. */
int vts_linear_relu_dropout_fwd(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                                int64_t M, int64_t N, int64_t K, float drop_p, uint64_t seed, int dtype_mode,
                                vts_stream_t stream);

/* ---- LayerNorm over the last dim (eps 1e-5): gpd-lstm-DE.py:19,28 ; mixture-vae-DE.py:46,61 ---- */
int vts_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                      float* mean, float* rstd, int64_t M, int64_t H, float eps, vts_stream_t stream);
size_t vts_layernorm_bwd_workspace_bytes(int64_t M, int64_t H);
int vts_layernorm_bwd(const float* dy, const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int64_t lddx, float* dgamma,
                      float* dbeta, int param_accumulate, void* workspace, size_t workspace_bytes,
                      int64_t M, int64_t H, vts_stream_t stream);

/* ---- Reparameterisation, noise fed in ----
 * VTS_GAUSSIAN: z = mean + eps*exp(0.5*logvar)          gaussian-lstm-DE.py:55-58, mixture-vae-DE.py:10-13
 * VTS_GPD     : z = scale/shape*((1-u)**(-shape)-1)     gpd-lstm-DE.py:10-12,     mixture-vae-DE.py:16-18
 * VTS_MIXTURE : row i takes the Gaussian sample iff choice[i] < softmax(pi)[0], else the GPD one
 *               (the per-row Python loop at mixture-vae-DE.py:94-107 as one kernel).
 * All latent tensors [B,L]; choice [B]; pi [2].  Unused inputs may be NULL. */
int vts_reparam_fwd(int family, const float* mean, const float* logvar, const float* scale,
                    const float* shape, const float* eps, const float* u, const float* choice,
                    const float* pi, float* z, int64_t B, int64_t L, vts_stream_t stream);
int vts_reparam_bwd(int family, const float* dz, const float* logvar, const float* scale,
                    const float* shape, const float* eps, const float* u, const float* choice,
                    const float* pi, float* dmean, float* dlogvar, float* dscale, float* dshape,
                    int64_t B, int64_t L, vts_stream_t stream);

/* ---- ELBO losses ----
 * VTS_GAUSSIAN: mse(recon,y) + beta*mean_B(-0.5*sum_L(1+lv-mu^2-e^lv))        gaussian-lstm-DE.py:66-70
 * VTS_GPD     : mse(recon,y) + beta*mean(log scale + log(1+shape*thr/scale))   gpd-lstm-DE.py:67-71
 * VTS_MIXTURE : mse + pi0*(mse over |y|<=thr + beta*KLg) + pi1*(mse + beta*KLp) mixture-vae-DE.py:115-130
 *               (NaN when no |y| <= thr, as the reference)
 * VTS_MIXTURE_SPLIT: mse + pi0*(mse over |y|<=thr + beta*KLg) + pi1*(mse over |y|>thr + beta*KLq),
 *               KLq = mean_B(-0.5*sum_L(1+scale-shape^2-e^scale))          mixture-vae-transformer.py:100-113
 *               (NaN when either selection is empty, as the reference)
 * fwd writes state[16]: state[0] = loss, the rest are coefficients for bwd.  bwd scales every
 * gradient by *grad_out (device scalar, NULL = 1).  recon,y [B,fs]; latents [B,L]; pi, dpi [2]. */
size_t vts_elbo_workspace_bytes(int64_t B, int64_t fs, int64_t L);
int vts_elbo_fwd(int family, const float* recon, const float* y, const float* mean,
                 const float* logvar, const float* scale, const float* shape, const float* pi,
                 float threshold, float beta, float* state, void* workspace, size_t workspace_bytes,
                 int64_t B, int64_t fs, int64_t L, vts_stream_t stream);
int vts_elbo_bwd(int family, const float* recon, const float* y, const float* mean,
                 const float* logvar, const float* scale, const float* shape, const float* state,
                 const float* grad_out, float threshold, float* drecon, float* dmean, float* dlogvar,
                 float* dscale, float* dshape, float* dpi, int64_t B, int64_t fs, int64_t L,
                 vts_stream_t stream);

/* vts_elbo_fwd for callers that cannot allocate device memory or pass a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, loss
 * and state[16] as vts_elbo_fwd (bit for bit), without the workspace: the per-block partial sums go into a
 * 24 KiB device array that is part of the library's code object (one copy per GPU, held until the process ends).
 * Limit: B * max(fs, L) <= vts_elbo_fwd_static_max_elements() (2^20); larger inputs return VTS_ERR_ARG.  The call
 * allocates and frees nothing.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (different streams) race; calls on one stream, or on different GPUs, do not. */
int64_t vts_elbo_fwd_static_max_elements(void);
int vts_elbo_fwd_static(int family, const float* recon, const float* y, const float* mean,
                        const float* logvar, const float* scale, const float* shape, const float* pi,
                        float threshold, float beta, float* state, int64_t B, int64_t fs, int64_t L,
                        vts_stream_t stream);

/* ---- Sampling fused with the KL sums, and the loss built from those sums ----
 * One training step reads mean/logvar/scale/shape twice in the reference: once to sample z
 * (mixture-vae-DE.py:94-107) and once for the KL terms of the loss (:123-127).  These entry points
 * do both in one pass over HBM and one pass back:
 *   vts_reparam_kl_fwd : z as vts_reparam_fwd, plus klpart[nblk][2] = per-block partial sums of
 *                        {1+lv-mu^2-e^lv,  log scale + log(1+shape*threshold/scale)},
 *                        nblk = vts_reparam_kl_blocks(B, L).  A mixture row reads only the noise of
 *                        the family it selected.
 *   vts_elbo_recon_fwd : the loss of vts_elbo_fwd from recon, y and klpart (same state[16] layout).
 *   vts_elbo_recon_bwd : drecon, dpi and dklpart[nblk][2] = d loss / d klpart.
 *   vts_reparam_kl_bwd : sampling gradient (from dz, NULL = 0) + KL gradient (from dklpart,
 *                        NULL = 0) written once to dmean/dlogvar/dscale/dshape.
 * Results equal vts_reparam_* + vts_elbo_* up to the order of the fp32 partial sums. */
int64_t vts_reparam_kl_blocks(int64_t B, int64_t L);
int vts_reparam_kl_fwd(int family, const float* mean, const float* logvar, const float* scale,
                       const float* shape, const float* eps, const float* u, const float* choice,
                       const float* pi, float threshold, float* z, float* klpart, int64_t B, int64_t L,
                       vts_stream_t stream);
int vts_reparam_kl_bwd(int family, const float* dz, const float* dklpart, const float* mean,
                       const float* logvar, const float* scale, const float* shape, const float* eps,
                       const float* u, const float* choice, const float* pi, float threshold,
                       float* dmean, float* dlogvar, float* dscale, float* dshape, int64_t B, int64_t L,
                       vts_stream_t stream);
int vts_elbo_recon_fwd(int family, const float* recon, const float* y, const float* klpart,
                       int64_t nklpart, const float* pi, float threshold, float beta, float* state,
                       void* workspace, size_t workspace_bytes, int64_t B, int64_t fs, int64_t L,
                       vts_stream_t stream);
int vts_elbo_recon_bwd(int family, const float* recon, const float* y, const float* state,
                       const float* grad_out, float threshold, float* drecon, float* dklpart,
                       int64_t nklpart, float* dpi, int64_t B, int64_t fs, vts_stream_t stream);

/* Data-parallel exact loss.  Every loss above is a mean over rows, and the mixtures' masked terms are means over
 * the rows of a class (mixture-vae-DE.py:119), so the mean of per-rank losses differs from the loss of the
 * concatenated batch whenever ranks hold different numbers of such rows.  vts_elbo_sums writes this rank's additive
 * statistics (sums[8]: six partial sums, the row count, one spare; KL slots from the latent tensors or, when klpart is
 * given, from vts_reparam_kl_fwd's partial sums); the caller adds them up over ranks (one all-reduce of 8 floats);
 * vts_elbo_from_sums forms the loss of the whole batch and this rank's backward state from the totals, for use
 * with vts_elbo_bwd / vts_elbo_recon_bwd.  n_ranks = the number of ranks, whose gradients the caller AVERAGES
 * afterwards: per-row gradient coefficients are multiplied by it, d loss / d pi_params (a whole-batch quantity every
 * rank holds in full) is not.  No counterpart in the reference (single process); with one rank the result is
 * identical to vts_elbo_fwd. */
int vts_elbo_sums(int family, const float* recon, const float* y, const float* mean, const float* logvar,
                  const float* scale, const float* shape, const float* klpart, int64_t nklpart, float threshold,
                  float* sums, void* workspace, size_t workspace_bytes, int64_t B, int64_t fs, int64_t L,
                  vts_stream_t stream);
int vts_elbo_from_sums(int family, const float* sums, const float* pi, float beta, float n_ranks, float* state,
                       int64_t fs, int64_t L, vts_stream_t stream);

/* ---- Fused head / decoder chains -------------------------------------------------------------------------
 * gpd-lstm-DE.py:29-33,45-47 ; mixture-vae-DE.py:62-69,81-83 ; gaussian-lstm-DE.py:30-31,43-45.
 * n_trunk (0..3) Linear + activation layers, then n_heads (1..8) Linear heads of equal width that all read the last
 * trunk output (the chain input when there is no trunk): the encoder's relu(fc1) -> dropout(relu(fc2)) ->
 * {mean, logvar, exp(scale), clamp(shape)} and the decoders' relu(fc1) -> dropout(relu(fc2)) -> out are each one such
 * chain.  One launch forward, two backward, fp32 arithmetic in every mode; same results as the layers run one by
 * one with vts_linear_fwd / vts_linear_bwd up to the order of fp32 sums.
 * Widths: multiples of 4, at most 512 (the heads together too); several heads need a width that is a multiple of
 * 16.  trunk_act: VTS_ACT_NONE / RELU / RELU_DROPOUT, head_act: NONE / RELU / EXP / CLAMP_MIN.
 * w, b: n_trunk + n_heads pointers (trunk first), w[i] = [N_i, K_i] row-major as nn.Linear keeps it; keep: n_trunk
 * pointers to [B, trunk_dim] 0/1 masks for the RELU_DROPOUT layers (NULL entries or NULL = eval mode), kept values
 * scaled by keep_scale; trunk_out[i] [B, trunk_dim[i]] and head_out[h] [B, head_dim] receive the post-activation
 * outputs (the backward pass reads them); head_aux[h] [B, head_dim] is needed for CLAMP_MIN heads only. */
typedef struct vts_mlp_chain {
  int32_t n_trunk, n_heads, in_dim, head_dim;
  int32_t trunk_dim[3], trunk_act[3];
  int32_t head_act[8];
  float keep_scale;
} vts_mlp_chain;
int vts_mlp_chain_supported(const vts_mlp_chain* chain, int64_t B);
int vts_mlp_chain_fwd(const vts_mlp_chain* chain, const float* x, int64_t ldx, int64_t B, const float* const* w,
                      const float* const* b, const float* const* keep, float* const* trunk_out, float* const* head_out,
                      float* const* head_aux, vts_stream_t stream);
size_t vts_mlp_chain_bwd_workspace_bytes(const vts_mlp_chain* chain, int64_t B);
/* d_head[h]: gradient w.r.t. head output h (NULL entry = zero); dx [B, in_dim] with row stride lddx (NULL = not
 * needed); dw[i], db[i] as w, b (db or its entries may be NULL).  Everything is overwritten, nothing accumulated. */
int vts_mlp_chain_bwd(const vts_mlp_chain* chain, const float* x, int64_t ldx, int64_t B, const float* const* w,
                      const float* const* keep, float* const* trunk_out, float* const* head_out,
                      float* const* head_aux, const float* const* d_head, float* dx, int64_t lddx, float* const* dw,
                      float* const* db, void* workspace, size_t workspace_bytes, vts_stream_t stream);

/* ---- Transformer encoder layer pieces: nn.TransformerEncoderLayer(d_model, nhead, dropout=0.2) ----
 * mixture-vae-transformer.py:39-40,49.  A layer is: packed in-projection (vts_linear_fwd) -> vts_mha_fwd ->
 * out-projection -> vts_dropout_add_fwd -> vts_layernorm_fwd -> vts_linear_relu_dropout_fwd -> vts_linear_fwd
 * -> vts_dropout_add_fwd -> vts_layernorm_fwd.
 *
 * Dropout sites take (keep, drop_p, seed): keep != NULL = explicit 0/1 byte mask; else seed != 0 = the
 * library's counter hash (element idx takes 16 bits of hash(seed, idx / 2) - the low ones if idx is even - and is kept iff they are >= drop_p * 2^16; vts_dropout_mask writes
 * that mask out); else no dropout (eval mode).  Kept elements are scaled by 1/(1-drop_p).
 *
 * vts_mha_*: softmax(q k^T / sqrt(dh)) v per (window, head), dropout on the weights.  q: Tq query rows per
 * window (row b*Tq + i, row stride ldq floats), k and v: T rows per window (row stride ldkv); head h reads
 * columns [h*dh, (h+1)*dh).  With the packed projection qkv[B*T, 3d]: q = qkv, k = qkv + d, v = qkv + 2d,
 * ldq = ldkv = 3d, Tq = T.  Tq = 1 with q from a [B, d] buffer is the last-token-only form (only
 * transformer_out[-1] is used, :50).  out [B*Tq, ldo]; lse [B, nhead, Tq] is saved for backward.  The mask
 * (or hash index) layout is [B, nhead, Tq, T].  dh in {8, 16, 32}; strides multiples of 4 floats; the score
 * matrix is never written to memory.  No counterpart of the attention weights is returned (need_weights=False). */
int vts_mha_fwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out,
                int64_t ldo, float* lse, const uint8_t* keep, float drop_p, uint64_t seed, int64_t B, int64_t T,
                int64_t Tq, int64_t nhead, int64_t dh, vts_stream_t stream);
int vts_mha_bwd(const float* dout, int64_t lddo, const float* q, int64_t ldq, const float* k, const float* v,
                int64_t ldkv, const float* out, int64_t ldo, const float* lse, const uint8_t* keep,
                float drop_p, uint64_t seed, float* dq, int64_t lddq, float* dk, float* dv, int64_t lddkv,
                int64_t B, int64_t T, int64_t Tq, int64_t nhead, int64_t dh, vts_stream_t stream);
/* Feed-forward block of the layer, y = linear2(dropout(relu(linear1(x)))), x [M, d] (row stride ldx),
 * w1 [F, d], w2 [d, F] (F = dim_feedforward, 2048 in the reference's layers).  The dropout mask: keep (float
 * 0/1 [M, F]) if given, else the counter hash when seed != 0 (element index row*F + col), else none.
 * fwd writes hidden [M, F] (needed by backward; the buffer must hold vts_ffn_hidden_bytes(M, F), rows padded to
 * a multiple of 128 that the kernels may scribble on) and, when mask != NULL, one bit per hidden element that is
 * non-zero (vts_ffn_mask_bytes(M, F) bytes; needs F % 64 == 0): backward then applies the relu/dropout
 * derivative from 1/32 of the bytes.  bwd: drop_scale = 1/(1-drop_p) if a mask was applied, else 1;
 * dx may be NULL.  linear2's data-gradient GEMM applies the activation derivative and accumulates db1 in
 * its epilogue, so the [M, F] gradient is written once and read twice (dx, dw1). */
size_t vts_ffn_hidden_bytes(int64_t M, int64_t F);
size_t vts_ffn_mask_bytes(int64_t M, int64_t F);
size_t vts_ffn_bwd_workspace_bytes(int64_t M, int64_t d, int64_t F);
int vts_ffn_fwd(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                const float* keep, float drop_p, uint64_t seed, float* hidden, void* mask, float* y, int64_t M,
                int64_t d, int64_t F, int dtype_mode, vts_stream_t stream);
int vts_ffn_bwd(const float* dy, const float* x, int64_t ldx, const float* w1, const float* w2,
                const float* hidden, const void* mask, float drop_scale, float* dx, int64_t lddx, float* dw1,
                float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, int64_t M, int64_t d,
                int64_t F, int dtype_mode, vts_stream_t stream);
/* The block's backward pass without any [M, F] tensor in memory: where vts_ffn_recompute_supported() answers 1
 * (bf16x3 mode, d = 64, F a multiple of 256, the shapes the fused kernels take) vts_ffn_fwd accepts hidden = NULL
 * and writes only the packed mask, and vts_ffn_bwd_recompute works hidden and the pre-activation gradient out
 * again from x, dy, the weights and that mask, tile by tile, inside the kernel that accumulates dw1 and dw2
 * (six products instead of four; 12 bytes per hidden element less HBM traffic, forward and backward together).
 * Same arguments as vts_ffn_bwd with b1 in place of hidden; dx is required; mask must be the buffer that very
 * forward call wrote (16-byte aligned, vts_ffn_mask_bytes(M, F) bytes: behind the mask words it carries the weight
 * image of the forward kernel, which the weight-gradient kernel reads).  nn.TransformerEncoderLayer's
 * linear1 / relu / dropout / linear2 at vae-models/mixture-vae-transformer.py:39. */
int vts_ffn_recompute_supported(const float* x, int64_t ldx, int64_t M, int64_t d, int64_t F, int dtype_mode);
size_t vts_ffn_bwd_recompute_workspace_bytes(int64_t M, int64_t d, int64_t F);
int vts_ffn_bwd_recompute(const float* dy, const float* x, int64_t ldx, const float* w1, const float* b1,
                          const float* w2, const void* mask, float drop_scale, float* dx, int64_t lddx, float* dw1,
                          float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, int64_t M,
                          int64_t d, int64_t F, int dtype_mode, vts_stream_t stream);
/* out = residual + dropout(x) (residual NULL = 0): x + dropout1(sa) and x + dropout2(ff) of the layer */
int vts_dropout_add_fwd(const float* residual, const float* x, const uint8_t* keep, float drop_p, uint64_t seed,
                        float* out, int64_t n, vts_stream_t stream);
/* dx = dout * mask / (1 - drop_p); the residual branch's gradient is dout itself */
int vts_dropout_bwd(const float* dout, const uint8_t* keep, float drop_p, uint64_t seed, float* dx, int64_t n,
                    vts_stream_t stream);
/* keep[i] = 1 iff the counter hash keeps element i for (seed, drop_p) */
int vts_dropout_mask(uint64_t seed, float drop_p, uint8_t* keep, int64_t n, vts_stream_t stream);

/* ---- Adam (torch.optim.Adam defaults, no weight decay) over one flat buffer ----
 * gaussian-lstm-DE.py:142,154 ; gpd-lstm-DE.py:88,104 ; mixture-vae-DE.py:147,163.
 * step is 1-based; grad_scale multiplies g first (1/world_size after a summed all-reduce). */
int vts_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                  vts_stream_t stream);
/* vts_adam_step behind a guard: n_guard (1..64) int32 words of device memory, e.g. the status words of the step's
 * recurrent launches (vts_lstm_recurrent_{fwd,bwd}_status, vts_lstm_fused_fwd_status).  If any of them is non-zero
 * when the kernel runs, no buffer is touched: gradients of a launch that timed out never reach the parameters, and
 * the host can enqueue the update without first waiting for the step to finish.  No counterpart in the reference
 * (torch raises on the host). */
int vts_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                          float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* guard,
                          int64_t n_guard, vts_stream_t stream);

/* ---- three-component latent mixture (Gaussian / GPD / Bernoulli) ----
 * graph/vae_mixture_graph_model.py with use_gcn = use_d_knn = False, use_gpd = use_bernoulli = True, the
 * configuration of graph/train_ld_2011.py:91.  Latent tensors [B][L], recon / y / choice [B] (future_steps 1),
 * pi [3] = pi_params.
 * reparameterize (:182-209): row b takes the Gaussian sample if choice[b] < p0, the GPD sample if
 * choice[b] < p0 + p1, else the Bernoulli sample, (p0, p1, p2) = softmax(pi).  `bern` holds either the 0/1
 * sample itself (bern_is_uniform = 0: replay of a recorded torch.bernoulli draw) or a uniform draw that is
 * compared with sigmoid(logits) here (bern_is_uniform = 1, :21-23).  The backward gives the sampling
 * gradients of mean / logvar / scale / shape; the pick and the Bernoulli sample carry none. */
int vts_reparam3_fwd(const float* mean, const float* logvar, const float* scale, const float* shape,
                     const float* logits, const float* bern, int bern_is_uniform, const float* eps, const float* u,
                     const float* choice, const float* pi, float* z, int64_t B, int64_t L, vts_stream_t stream);
int vts_reparam3_bwd(const float* dz, const float* logvar, const float* scale, const float* shape, const float* eps,
                     const float* u, const float* choice, const float* pi, float* dmean, float* dlogvar, float* dscale,
                     float* dshape, int64_t B, int64_t L, vts_stream_t stream);
/* loss_function (:222-285) with rows classed by y (classify_observations :26-50: zero if y == 0, else normal
 * if y <= threshold, else extreme):
 *   loss = p0*(mse(recon, y | normal) + beta*KL_gauss) + p1*mean_{extreme rows, L}[log s + (1 + 1/xi)*log(1 + xi*(y-u)/s)]
 *        + p2*(-mean_{zero rows, L}[y*log sig(l) + (1-y)*log(1 - sig(l))]);  a class without rows contributes 0,
 *   except that an empty normal class makes the loss NaN (mse of an empty selection), as in the reference.
 * state: 16 floats written by the forward and read by the backward; gout: [1] upstream gradient. */
size_t vts_elbo3_workspace_bytes(int64_t B, int64_t L);
int vts_elbo3_fwd(const float* recon, const float* y, const float* mean, const float* logvar, const float* scale,
                  const float* shape, const float* logits, const float* pi, float threshold, float beta, float* loss,
                  float* state, void* workspace, size_t workspace_bytes, int64_t B, int64_t L, vts_stream_t stream);
int vts_elbo3_bwd(const float* recon, const float* y, const float* mean, const float* logvar, const float* scale,
                  const float* shape, const float* logits, float threshold, float beta, const float* state,
                  const float* gout, float* drecon, float* dmean, float* dlogvar, float* dscale, float* dshape,
                  float* dlogits, float* dpi, int64_t B, int64_t L, vts_stream_t stream);

/* ---- training windows cut on the device from a series resident in HBM ----
 * The step in front of the train step: the reference builds windows on the host, standardises the copies
 * and uploads them -- graph/train_ld_2011.py:11-17 (create_sliding_windows), :47-59 (StandardScaler
 * fit / transform); block-maximum targets graph/training_block_max_hurdat.py:81-86,100-106.
 *   series [Nt][C] fp32 row-major; starts [B] int64 (device): first series row of each window
 *   X [B][T][C] (16-byte aligned) = series rows starts[b] .. starts[b]+T-1, or null
 *   y: VTS_TARGET_NEXT      [B][P]  column 0 of the P rows after the window (train_ld_2011.py:15)
 *      VTS_TARGET_BLOCK_MAX [B]     their maximum, NaN-propagating like numpy.max (training_block_max_hurdat.py:103)
 *      or null
 *   x_mean, x_scale [C] fp64 / y_stats [2] fp64 (mean, scale), device, or null = raw values.
 *   Standardised values equal sklearn's StandardScaler.transform on float32 input bit for bit:
 *   f32(f64(f32(f64(x) - mean)) / scale).  A window (or target) that leaves the series is written as NaN. */
#define VTS_TARGET_NEXT 0
#define VTS_TARGET_BLOCK_MAX 1
int vts_windows_build(const float* series, int64_t Nt, int64_t C, const int64_t* starts, int64_t B, int64_t T,
                      int64_t P, int target, const double* x_mean, const double* x_scale, const double* y_stats,
                      float* X, float* y, vts_stream_t stream);
/* StandardScaler.fit over the rows of those windows (X.reshape(-1, C): a series row counts once per window
 * that contains it) and over their VTS_TARGET_NEXT targets: population variance, scale = sqrt(var), 1 for a
 * constant column.  x_stats [2][C] fp64 = means then scales; y_stats [2] fp64; either may be null.
 * fp64 sums in a fixed order (deterministic); agrees with sklearn to ~1e-13 relative, not bit for bit. */
size_t vts_windows_stats_workspace_bytes(int64_t Nt);
int vts_windows_stats(const float* series, int64_t Nt, int64_t C, const int64_t* starts, int64_t B, int64_t T,
                      int64_t P, int target, double* x_stats, double* y_stats, void* workspace,
                      size_t workspace_bytes, vts_stream_t stream);

/* The same update with the step counter kept on the device, for callers that record the step's launches once
 * and replay them (HIP graphs): step_state is 16 bytes of device memory, [int64 step][2 floats of scratch],
 * zero before the first step; each call increments the counter and uses the new value. */
int vts_adam_step_counted(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          int64_t* step_state, float lr, float beta1, float beta2, float eps, float grad_scale,
                          vts_stream_t stream);

/* dst[dst_offset[c] .. + numel[c]) = src[c] for c < count (src[c] NULL: zeros): the gradients autograd produced, one
 * tensor a parameter, into the flat gradient buffer the optimizer and the all-reduce work on (the flat-buffer
 * counterpart of optimizer.step() reading p.grad: gaussian-lstm-DE.py:155, gpd-lstm-DE.py:104,
 * mixture-vae-DE.py:163).  src, numel, dst_offset are host arrays; one launch per 40 chunks. */
int vts_gather_chunks(const float* const* src, const int64_t* numel, const int64_t* dst_offset, int64_t count,
                      float* dst, vts_stream_t stream);

/* out[i] = in[i] * (*scalar) */
int vts_scale(const float* in, const float* scalar, float* out, int64_t n, vts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VTS_H */
