/* libha2g_hip.so -- C ABI of the hand-written gfx950 (MI355X / CDNA4) kernels behind the HA2G hierarchy
 * train step.
 *
 * The reference (alvinliu0/HA2G) has no FFI: its hot path is Python/torch modules.  Each entry point below
 * replaces the torch operator(s) the reference reaches through the cited lines (paths relative to the
 * reference's scripts/ directory); the Python mirror of the reference interface (the ha2g_amd package) binds them
 * with ctypes, and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only (no torch types); all pointers are DEVICE pointers to fp32
 * (or int64/int32 where typed) unless noted; outputs and workspaces are caller-allocated; no allocation,
 * no implicit synchronisation; every call enqueues on `stream` (a hipStream_t passed as void*) and is safe
 * to capture into a hipGraph; re-entrant across streams.  Return 0 on success, negative on error with the
 * message available from ha2g_last_error() (thread-local).
 * The ha2g_*_set_mode / ha2g_*_debug_* switches are PROCESS-GLOBAL configuration (which arithmetic variant the GEMM / convolution entry
 * points dispatch to): set them before enqueueing work, not concurrently with launches from another thread.  Tile and split-K choices
 * (hence the fp32 summation order of a shape) depend on the device's compute-unit count, queried once per device.
 */
#ifndef HA2G_HIP_H
#define HA2G_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature changes or entry points are added that a host must not mix with an older library:
 * ha2g_amd/_lib.py refuses to bind a library whose ha2g_abi_version() differs from this macro. */
#define HA2G_ABI_VERSION 6
int ha2g_abi_version(void);
const char* ha2g_last_error(void);

/* ---- dense fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32) -------------------------------------------------
 * C[M,N] = act(alpha*op(A)*op(B) + beta*C + bias[n]);  row-major; transa=1: A stored [K,M]; transb=1: B stored [N,K].
 * act: 0 none, 1 relu, 2 leaky-relu(0.01), 3 sigmoid.  ws/ws_bytes: optional split-K scratch.
 * Replaces nn.Linear / addmm everywhere on the path: GRU input projections (model/hierarchy_net.py:87-88,144),
 * generator head :89-93,146, TCN convs as im2col GEMMs (model/tcn.py:19-31), tap FCs
 * (model/ResNetSE34V2.py:36,40,44,163,174,185), speaker MLPs (:194-202), SE FCs (model/ResNetBlocks.py:84-89). */
int ha2g_gemm_f32(int transa, int transb, int M, int N, int K, float alpha, const float* A, long lda, const float* B,
                  long ldb, float beta, float* C, long ldc, const float* bias, int act, float* ws, long ws_bytes,
                  void* stream);
/* dW[M,N] = beta*dW + dY^T X  and  db[M] = bias_beta*db + column sums of dY  in one launch (+ the split-K reduce launch it needs anyway);
 * dY stored [K][M] (K = samples), X [K][N].  The weight and bias gradients of nn.Linear / nn.Conv1d (autograd's addmm backward + sum(0):
 * model/hierarchy_net.py:87-93, model/tcn.py:19-31, model/ResNetSE34V2.py:163-202) without a separate pass over dY. */
int ha2g_gemm_wgrad_bias_f32(int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float beta, float* dW, long ldw,
                             float bias_beta, float* db, float* ws, long ws_bytes, void* stream);
/* `groups` (1..8) independent GEMMs of one shape in ONE launch: per group g the semantics of ha2g_gemm_f32 (and, with csumg on the
 * weight-gradient shape transa=1/transb=0, of ha2g_gemm_wgrad_bias_f32) on Ag[g], Bg[g], Cg[g], biasg[g], csumg[g] (host arrays of device
 * pointers; biasg / csumg or their entries may be null).  The same layer of the three (TED-Gesture) or six (TED-Expressive) generators'
 * text encoders -- separate nn.Module instances in the reference (model/hierarchy_net.py:66-70) -- fills the chip as one grid instead of
 * three under-filled ones. */
int ha2g_gemm_grouped_f32(int groups, int transa, int transb, int M, int N, int K, float alpha, const float* const* Ag, long lda,
                          const float* const* Bg, long ldb, float beta, float* const* Cg, long ldc, const float* const* biasg, int act,
                          float* const* csumg, float csum_beta, float* ws, long ws_bytes, void* stream);
/* out[c] = beta*out[c] + sum_r X[r*ld + c]  (bias gradients); ws >= ha2g_colsum_workspace_bytes(cols) */
long ha2g_colsum_workspace_bytes(int cols);
int ha2g_colsum_f32(const float* X, long ld, long rows, int cols, float* out, float beta, float* ws, void* stream);

/* ---- NHWC convolution as implicit GEMM (audio encoder: model/ResNetSE34V2.py:27-42,96-116,
 *      model/ResNetBlocks.py:12-15,24-29).  x [N,H,W,Cin], w [Cout][KH][KW][Cin], y [N,OH,OW,Cout]. ---- */
int ha2g_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, int act, void* stream);
/* wt = weight permuted to [Cin][KH][KW][Cout] by ha2g_conv2d_weight_ohwi_to_ihwo_f32; dx = beta*dx + ... */
int ha2g_conv2d_dgrad_f32(const float* dy, const float* wt, float* dx, int N, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, float beta, void* stream);
long ha2g_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream);
int ha2g_conv2d_weight_ohwi_to_ihwo_f32(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream);
/* ---- plane-based split-bf16 products (round 3; csrc/conv_planes.hip): the operands arrive PRE-SPLIT as two bf16 planes (hi = bf16(x),
 *      lo = bf16(x - hi)) written once by their producer (ha2g_bn_bwd_planes_f32, the weight re-layout below) and go global -> LDS by DMA; the
 *      same three-MFMA product, k order and accumulation as the kernels behind ha2g_conv2d_dgrad_f32 in the default mode: bit-identical results.
 *      Autograd's conv2d backward w.r.t. the input (model/ResNetBlocks.py:24-29 under loss.backward(), train_eval/train_hierarchy.py:264). ---- */
void ha2g_conv_planes_enable(int on);
void ha2g_conv_planes_debug(int bits);     /* timing ablations only: 1 = no DMA after the first k tile, 2 = no MFMA (results are then meaningless) */
int ha2g_f32_to_planes(const float* x, void* hi, void* lo, long n, void* stream);
int ha2g_conv2d_weight_ihwo_planes(const float* w, void* wt_hi, void* wt_lo, int Cout, int KH, int KW, int Cin, void* stream);
/* ha2g_conv2d_weight_ihwo_planes for n <= 48 weights in one launch; w / wt_hi / wt_lo / cout / kk / cin are HOST arrays of n entries */
int ha2g_conv2d_weight_ihwo_planes_multi(const void* const* w, void* const* wt_hi, void* const* wt_lo, const int* cout, const int* kk, const int* cin,
                                         int n, void* stream);
/* ABI 2: the same operations on np = 2 or 3 PIECE planes, piece q of a tensor at base + q * ps elements (equally spaced planes of one
 * allocation).  np = 3 (x = p0 + p1 + p2, all 24 mantissa bits; six MFMAs per product, smallest first) is the fp32-class form that the default
 * backward runs (ha2g_gemm_set_mode bit 6, ha2g_gemm_bwd_pieces): the arithmetic class of the reference's loss.backward()
 * (train_eval/train_hierarchy.py:264); the two-plane entry points above are np = 2, ps = lo - hi. */
int ha2g_f32_to_planes_np(const float* x, void* planes, long ps, int np, long n, void* stream);
int ha2g_conv2d_weight_ihwo_planes_multi_np(const void* const* w, void* const* wt, const long* ps, const int* cout, const int* kk, const int* cin,
                                            int n, int np, void* stream);
int ha2g_conv2d_dgrad_planes_np_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W,
                                    int Cin, int Cout, int KH, int KW, int stride, int pad, float beta, void* stream);
int ha2g_conv2d_wgrad_planes_np_f32(const void* x, long x_ps, const void* dy, long dy_ps, int np, float* dw, int N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream);
/* forward convolution of trunk layers 2-4 from three-piece planes (fp32 output, optional ReLU): nn.Conv2d forward (model/ResNetBlocks.py:24-29,
 * model/ResNetSE34V2.py:96-116) at fp32-class accuracy; x planes from the producers (ha2g_bn_apply_planes_np_f32, ha2g_se_scale_add_relu_planes_np_f32),
 * weight planes [Cout][KH][KW][Cin] from ha2g_f32_to_planes_multi_np (n <= 48 tensors in one launch, HOST arrays) */
int ha2g_f32_to_planes_multi_np(const void* const* x, void* const* planes, const long* ps, const long* numel, int n, int np, void* stream);
int ha2g_conv2d_fwd_planes_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_fwd_planes_np_f32(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride, int pad, int relu, void* stream);
/* ... with the statistics of the BatchNorm that follows (conv -> [ReLU] -> BatchNorm, model/ResNetBlocks.py:24-29) left behind by the same launch:
 * stat_part [2][Cout][stat_nblk] doubles = per row tile the sum and the sum of squares of the stored output; stat_nblk =
 * ha2g_conv2d_fwd_planes_stat_blocks(...) (0: this geometry's kernel has no statistics epilogue -- run ha2g_bn_stats_f32 on y instead);
 * ha2g_bn_stats_finalize_f32 turns the partial sums into mean / invstd / running statistics (ABI 4) */
int ha2g_conv2d_fwd_planes_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_fwd_planes_stat_tiles_per_image(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);   /* > 0: blocks = tiles inside one image, that many per image */
int ha2g_conv2d_fwd_planes_np_stats_f32(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout,
                                        int KH, int KW, int stride, int pad, int relu, void* stat_part, int stat_nblk, void* stream);
/* dense products on three-piece planes (nn.Linear / GRU input projections and their backward, model/hierarchy_net.py:87-93,144-147):
 * ha2g_f32_to_planes_2d_np splits a 2-D fp32 operand into K-contiguous, zero-padded piece planes (transpose = 1: the planes hold x^T);
 * ha2g_gemm_planes_np_f32: C [M][N] = act(A B^T + bias) + beta C from the planes of A [M][lda] and B [N][ldb], lda = ldb = K rounded up to 32 */
int ha2g_f32_to_planes_2d_np(const float* x, long ldx, long rows, int cols, void* planes, long ps, long ldp, int np, int transpose, void* stream);
/* n <= 16 of those splits in one launch, each into its own window of ONE plane set (ABI 5): job i splits x[i] [rows[i]][ldx[i]] (cols[i] valid columns) into the
 * window at planes[i] (piece q at planes[i] + q * ps elements, row stride ldp): transpose = 0 -> rows[i] x wcols[i] with zeros in [cols[i], wcols[i]);
 * transpose = 1 -> cols[i] x wcols[i] holding x^T with zeros in [rows[i], wcols[i]).  The arrays are HOST arrays.  Builds the merged operands
 * [W_ih; W_ih_reverse] of every layer of an nn.GRU stack (model/hierarchy_net.py:87) without a torch.cat + split per layer. */
int ha2g_f32_to_planes_2d_multi_np(const void* const* x, const long* ldx, const int* rows, const int* cols, void* const* planes, const int* wcols, long ps, long ldp,
                                   int n, int np, int transpose, void* stream);
int ha2g_gemm_planes_np_f32(const void* a, long a_ps, long lda, const void* b, long b_ps, long ldb, int np, int M, int N, int K, float beta,
                            float* C, long ldc, const float* bias, int act, float* ws, long ws_bytes, void* stream);
void ha2g_conv_c32_wgrad_prefetch(int on);   /* A/B (ABI 5): the 32-channel three-piece weight gradient with the next tile's operands prefetched into registers (1, default; bit-identical) */
void ha2g_conv_c32_prefetch(int on);             /* A/B: the 32-channel three-piece direct convolution with the next tile's loads prefetched into registers (1, default) or the single-buffered first form (0) */
void ha2g_conv_planes_bufaddr(int on);          /* A/B: the plane kernel's DMA through buffer resources (1, default: padding = out-of-range offsets) or flat addresses + zero page (0) */
void ha2g_conv_planes_korder(int channel_major);  /* A/B: k order of the plane kernel's split products: 1 (default) = the nine taps of a 32-channel slice back to back (the re-read pixels hit L2), 0 = tap-major */
void ha2g_conv_planes_tile3(int t);       /* A/B: tile of the three-piece plane kernel (0 = default, 1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = the eight-wave ping-pong kernel 256x128 / 256x64) */
int ha2g_gemm_bwd_pieces(void);           /* 0 = backward products on the fp32 MFMA / plain bf16, 2 / 3 = bf16 pieces per operand of the split products */
int ha2g_conv2d_dgrad_planes_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_dgrad_planes_f32(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, float beta, void* stream);
/* dw [Cout][KH][KW][Cin] = beta*dw + dy^T im2col(x) from the planes of x [N,H,W,Cin] and dy [N,H,W,Cout] (3x3 / stride 1 / pad 1, channels
 * multiples of 64): every pixel of x and dy goes global -> LDS once per 64 x 64 block of dW (the nine taps share the staged patch); fp32
 * accumulation per workgroup, the workgroups' partial slabs (ws >= ..._workspace_bytes) are added in double.  Autograd's conv2d backward w.r.t.
 * the weight (model/ResNetBlocks.py:24-29). */
/* Data-parallel gradient exchange without torch.distributed (csrc/comm.hip): RCCL behind plain pointers, one process per GPU.  Rank 0 fills
 * a 128-byte id (host memory) and hands it to the other ranks; every rank calls ha2g_comm_init on its own device (collective); the flat fp32
 * gradient buffers then go through ha2g_allreduce_bucket in place on the caller's stream (average = 1: mean over ranks).  RCCL is resolved
 * with dlopen("librccl.so.1") at the first call -- ha2g_comm_available() = 0 when it is absent.  Replaces nn.DataParallel, scripts/train.py:133-143. */
int ha2g_comm_available(void);
int ha2g_comm_unique_id(void* id128);
int ha2g_comm_init(const void* id128, int rank, int world, void** comm);
int ha2g_comm_world(void* comm);
int ha2g_allreduce_bucket(void* comm, float* buf, long n, int average, void* stream);
int ha2g_comm_destroy(void* comm);
/* bf16-storage mode (BASELINE config 5, bench.py --bf16): x / y / dy / dx are bf16 tensors in HBM, w bf16 [Cout][KH][KW][Cin], wt bf16
 * [Cin][KH][KW][Cout] (= the hi plane of ha2g_conv2d_weight_ihwo_planes); fp32 accumulation; 3x3 / pad 1 or 1x1 / pad 0, stride 1 or 2,
 * channel counts multiples of 32 (nn.Conv2d and its backward w.r.t. the input, model/ResNetBlocks.py:24-29, model/ResNetSE34V2.py:96-116) */
void ha2g_side_cus(int n);
void ha2g_conv_planes_waves(int n);       /* A/B: 4 (default) or 8 waves per workgroup of the plane convolution kernel */
void ha2g_conv_planes_ring(int depth);   /* A/B: LDS ring depth of the plane convolution kernel (0 = default, 2..4) */
int ha2g_conv2d_b16_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_fwd_b16(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                        void* stream);
int ha2g_conv2d_dgrad_b16(const void* dy, const void* wt, void* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          float beta, void* stream);
int ha2g_conv2d_wgrad_b16_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
long ha2g_conv2d_wgrad_b16_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int ha2g_conv2d_wgrad_b16(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          float beta, float* ws, long ws_bytes, void* stream);
/* bf16-storage mode, the HBM-bound passes of the audio tower over bf16 tensors (same arithmetic as their _f32 twins: fp32 / double statistics and
 * accumulators, round-to-nearest-even on every activation store).  ha2g_add_f32_to_b16: out = bf16(a + b), a bf16 or NULL, b fp32. */
int ha2g_bn_stats_b16(const void* x, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var, float momentum,
                      float eps, float* ws, void* stream);
int ha2g_bn_apply_b16(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y, long rows, int C,
                      int act, void* stream);
int ha2g_bn_apply_pool_b16(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y, int N, int HW,
                           int C, float* pooled, float* ws, void* stream);
int ha2g_bn_bwd_b16(const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma, void* dx, float* dgamma,
                    float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta, float* ws, void* stream);
int ha2g_se_scale_add_relu_b16(const void* x, const float* s, const void* res, void* out, int N, int HW, int C, void* stream);
int ha2g_se_bwd_scale_b16(const void* dout, const void* out, const void* x, float* ds, int N, int HW, int C, const float* gate, float* ws,
                          void* stream);
int ha2g_se_bwd_apply_b16(const void* dout, const void* out, const float* s, const float* dpool, void* dres, void* dx, int N, int HW, int C,
                          void* stream);
int ha2g_f32_to_b16(const float* x, void* y, long n, void* stream);
int ha2g_b16_to_f32(const void* x, float* y, long n, void* stream);
int ha2g_add_f32_to_b16(const void* a, const float* b, void* out, long n, void* stream);
int ha2g_stem_conv_fwd_b16(const float* x, const float* w, const float* bias, void* y, int N, int H, int W, void* stream);
int ha2g_stem_conv_wgrad_b16(const float* x, const void* dy, float* dw, float* db, int N, int H, int W, float beta, float* ws, void* stream);
int ha2g_conv2d_wgrad_planes_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
long ha2g_conv2d_wgrad_planes_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int ha2g_conv2d_wgrad_planes_f32(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N, int H, int W, int Cin,
                                 int Cout, int KH, int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream);
/* Matrix-core mode bits; default 6.  The split-bf16 inner product writes each fp32 operand as hi + lo bf16 halves and
 * runs a_lo*b_hi + a_hi*b_lo + a_hi*b_hi as three bf16 MFMAs with fp32 accumulation (~4e-6 rms-rel per GEMM vs 4e-7).
 *   bit 1 (on):  WEIGHT gradients (dW = dY^T X, conv wgrad) -- the error goes straight to the optimizer;
 *   bit 2 (on):  DATA gradients (dX = dY W, conv dgrad) -- measured: the worst error/tolerance ratio of the whole-step
 *                parity checks does not move (0.698 -> 0.699, tools/margins.py);
 *   bit 0 (off): forward GEMMs / convolutions too (another 1.5-2.5x on those, but the error compounds through the
 *                34-layer audio tower and the step no longer meets the 1e-4 parity bar).
 *   bit 3 (off): forward GEMMs / convolutions on the 3-piece split (6 bf16 MFMAs): as accurate as the fp32 MFMA chain
 *                (3.7e-7 vs 4.4e-7 rms-rel), 1.1-1.45x faster per GEMM, ~1.4 % of the step.
 *   bit 4 (off): every vectorisable GEMM / convolution with PLAIN bf16 operands (one bf16 MFMA per product, fp32 accumulate,
 *                fp32 storage) -- reduced precision (2^-9 per operand), for BASELINE config 5 / `bench.py --bf16` only.
 *   bit 5 (off): the 3-piece split of bit 3 for forward DENSE GEMMs only (GRU input projections, TCN / discriminator im2col GEMMs,
 *                generator head): fp32-accurate; the convolutions and the audio tower's own (narrow) GEMMs stay on the fp32 MFMA, so the tower's
 *                forward arithmetic is bit-identical with and without it (tests/test_gpu_kernels.py).  Measured slower on the whole step (60.1 vs 58.0 ms:
 *                the skinny-K projections are LDS-bound on the split kernel), kept opt-in.
 *   bit 6 (ON by default since round 4 -- ha2g_amd/_lib.py sets mode 70 = 64 + 6): the split backward products of bits 1 / 2 use THREE bf16
 *                pieces per operand and six MFMAs (x = p0 + p1 + p2 holds all 24 mantissa bits; products down to 2^-24 kept): fp32-class,
 *                i.e. the arithmetic class of the reference's fp32 loss.backward() (train_hierarchy.py:264).  Kernel families without a
 *                three-piece form (the BPTT chain, the direct 32-channel kernels, shapes the planes do not serve) run the EXACT fp32 MFMA in
 *                this mode -- never two pieces.  Mode 6 (two pieces, 16-bit operand mantissa) survives as a labelled secondary mode.
 * 0 = exact fp32 MFMA everywhere. */
void ha2g_gemm_set_mode(int mode);
/* ---- in-kernel split-K reduction (ABI 5) -------------------------------------------------------------------------------------------------
 * A split-K launch (small tile grids: weight gradients dW = dY^T X of nn.Linear / nn.GRU / the TCN convolutions, the GRU's dX product; the
 * reference's addmm backward, model/hierarchy_net.py:87-93,144-147, model/tcn.py:19-31) leaves `splits` raw slabs in `ws`.  With a ticket buffer
 * registered for the launch stream the slices of an output tile take a ticket and the LAST arriver adds the slabs in slice order (double
 * accumulation: the arithmetic of the reduce launch it replaces; no float atomics, bitwise reproducible) inside the same launch.
 * ha2g_splitk_set_tickets: `tickets` = ha2g_splitk_ticket_words() ints of device memory, ZEROED once by the caller and never written by it again
 * (the kernels leave them zero); applies to every split-K launch on `stream` of the current device from then on; tickets = NULL unregisters
 * (a second reduce launch per split-K launch, as before ABI 5).  One buffer per stream: kernels of one stream run in order.
 * ha2g_splitk_in_kernel(0): A/B switch -- ignore the registered buffers. */
int ha2g_splitk_set_tickets(void* tickets, long n_ints, void* stream);
int ha2g_splitk_ticket_words(void);
void ha2g_splitk_in_kernel(int on);
/* tuning aids: the largest number of slab bytes one workgroup adds in the kernel (larger tiles x splits keep the reduce launch; default 1.5 MB);
 * the plane GEMM's split-K rule (model = 1: time model + in-kernel reduction, 0: the round-5 rule; force > 0: that many k slices) */
void ha2g_gemm_debug_inkernel_bytes(long n);
void ha2g_gemm_debug_plane_ksplit(int model, int force);
/* tuning aid: forward GEMMs narrower than n columns stay on the fp32 MFMA */
void ha2g_gemm_debug_x6_min_n(int n);
/* tuning aid (tools/gemm_tile_sweep.py): force the dense tile shape (0-8, see kTileBM/kTileBN in csrc/gemm.hip; -1 = default rule,
 * -2 = round 1's tile rule; results are bit-identical under every tile shape) and the split-K
 * count (0 = heuristic) of every following ha2g_gemm_f32 call */
void ha2g_gemm_debug_tile(int cfg, int splits);
/* bit 0 (default 1): forward 32->32 channel 3x3 convolutions on the direct LDS-patch kernel conv_c32.hip (else implicit GEMM; the two are
 * bit-identical: same MFMA chain per accumulator);
 * bit 1 (default 0): their data gradients on the fp32 direct kernel; bit 2 (default 0): 1 = take their data gradients OFF the
 * split-bf16 direct kernel (default path, 121 vs 227 us) back to the implicit GEMM; bits 4-5: timing ablations */
void ha2g_conv_debug_direct_c32(int on);
void ha2g_conv_debug_cfg(int cfg);   /* tile-shape override for tools/conv_bench.py (-1 = heuristic) */
/* stem Conv2d(1->32, 3x3, pad 1) + bias + ReLU (model/ResNetSE34V2.py:27,127-128); x [N,H,W], y [N,H,W,32] */
int ha2g_stem_conv_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, void* stream);
/* dw [32][3][3], db [32] (= beta * old + gradient); dy = gradient w.r.t. the pre-ReLU output; ws >= 4096 * 320 floats of scratch */
int ha2g_stem_conv_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, float beta,
                             float* ws, void* stream);
/* NHWC PixelShuffle(r) (model/ResNetSE34V2.py:166-167,177-178); inverse=1 routes the gradient back */
int ha2g_pixel_shuffle_f32(const float* in, float* out, int N, int H, int W, int Cout, int r, int inverse, void* stream);

/* [N][H][W][C] -> [N][W][C][H]: tap-FC input rows with K index c*H+h (model/ResNetSE34V2.py:160-162); inverse=1 maps back */
int ha2g_nhwc_to_nwch_f32(const float* in, float* out, int N, int H, int W, int C, int inverse, void* stream);

/* ---- bidirectional GRU layer (nn.GRU: model/hierarchy_net.py:87-88,144 generator; :213,232 discriminator) ---- */
long ha2g_gru_packed_floats(int H);           /* floats of one packed W_hh image (per direction, per form) */
int ha2g_gru_supported_hidden(int H);         /* 300, 64, 32 are instantiated */
int ha2g_gru_pack_whh(const float* whh, float* packed_fwd, float* packed_bwd, int H, void* stream);
/* the same for n <= 16 matrices in one launch; whh / pf / pb are HOST arrays of n device pointers */
int ha2g_gru_pack_whh_multi(const void* const* whh, void* const* pf, void* const* pb, int n, int H, void* stream);
/* the layer's four bias gradients from ONE column sum (ha2g_colsum_f32) of dg [rows][(r z n_i) fwd | (r z n_i) rev | n_h fwd | n_h rev]: d b_ih = (r,z,n_i),
 * d b_hh = (r,z,n_h); beta = 1 accumulates into the destinations */
int ha2g_gru_bias_grads_f32(const float* colsums, float* dbih_fwd, float* dbhh_fwd, float* dbih_rev, float* dbhh_rev, int H,
                            float beta, void* stream);
/* gi [B][T][2][3H] = x W_ih^T + b_ih (both directions); wp = packed_fwd images (dir 0, dir 1) back to back;
 * y [B][T][2H]; rs (nullable reserve) [B][T][2][4][H] */
int ha2g_gru_layer_fwd(const float* gi, const float* wp, const float* bhh_fwd, const float* bhh_rev, float* y,
                       float* rs, int B, int T, int H, void* stream);
/* Workgroup-cluster form of the recurrence (H = 300, T <= ha2g_gru_cluster_max_steps()): 5 workgroups per (16-row tile,
 * direction) keep their slice of W_hh in registers and all-gather h every step through 8-byte {value, tag} granules; the k-blocks
 * of the MFMA chain are consumed member by member while the next member's granules are still in flight.
 * xch: scratch of ha2g_gru_cluster_workspace_bytes(), ZEROED ONCE at allocation (its tail holds the device-side launch epoch that
 * stamps the tags, so no per-launch clearing and hipGraph replays stay correct); err: device int32, set to 1 if a hand-off timed
 * out (the step's results are then invalid).  ha2g_gru_cluster_supported(H): H == 300 AND the current device has >= 10 compute
 * units (every workgroup of a launch must be co-resident; a launch takes CUs/10 batch tiles, at most 24) -- otherwise use
 * ha2g_gru_layer_fwd / _bwd. */
long ha2g_gru_cluster_workspace_bytes(void);
int ha2g_gru_cluster_max_steps(void);
int ha2g_gru_cluster_supported(int H);
int ha2g_gru_layer_fwd_cluster(const float* gi, const float* wp, const float* bhh_fwd, const float* bhh_rev, float* y,
                               float* rs, void* xch, int* err, int B, int T, int H, void* stream);
/* ABI 2: the THREE-PIECE form of the cluster forward (round 4, the default in the fp32-class mode): the recurrent product W_hh h runs as six
 * v_mfma_f32_16x16x32_bf16 per 32 k on three bf16 pieces of both operands (all 24 mantissa bits; fp32 accumulate) instead of eight fp32 MFMAs --
 * 0.39 of the fp32 chain's matrix-pipe cycles at fp32-class accuracy (NOT bit-identical to ha2g_gru_layer_fwd).  wp3 = the two directions'
 * three-piece W_hh images (ha2g_gru_pack_whh3 into direction d at byte offset d * ha2g_gru_packed3_bytes()); the rest as above. */
long ha2g_gru_packed3_bytes(void);
int ha2g_gru_pack_whh3(const float* w_hh, void* out, int H, void* stream);
/* ha2g_gru_pack_whh3 (transposed = 0) / ha2g_gru_pack_whh3t (1) for n <= 16 matrices in one launch; w / out: HOST arrays of n device pointers */
int ha2g_gru_pack_whh3_multi(const void* const* w, void* const* out, int n, int H, int transposed, void* stream);
int ha2g_gru_layer_fwd_cluster3(const float* gi, const void* wp3, const float* bhh_fwd, const float* bhh_rev, float* y, float* rs,
                                void* xch, int* err, int B, int T, int H, void* stream);
/* the BPTT twin on three pieces: wp3t = the transposed images (ha2g_gru_pack_whh3t); contract of ha2g_gru_layer_bwd_cluster otherwise */
int ha2g_gru_pack_whh3t(const float* w_hh, void* out, int H, void* stream);
int ha2g_gru_layer_bwd_cluster3(const float* dy, const float* y, const float* rs, const void* wp3t, float* dg, float* hp, void* xch, int* err,
                                int B, int T, int H, void* stream);
int ha2g_gru_layer_bwd_cluster(const float* dy, const float* y, const float* rs, const float* wpt, float* dg, float* hp, void* xch,
                               int* err, int B, int T, int H, void* stream);
/* ablation bits for tools/dbg_cluster.py: 1 no wait, 2 no exchange, 4 force the write-through publish */
void ha2g_gru_cluster_debug(int mode);
/* timing probe of the BPTT step loop (debug bit 7 of ha2g_gru_cluster_debug, tools/gru_fwd3_bench.py): out[5 members][16] = shader-clock cycles per
 * phase of wave 0 of cluster 0, summed over the steps of the last probed launch; [m][8] = the number of steps.  HOST pointer. */
int ha2g_gru_cluster_prof(unsigned long long* out);
/* ABI 3 (round 5): a process that shares its device caps the 16-row tiles of one cluster launch at its share of compute units / 10 (0 = whole device) */
void ha2g_gru_cluster_tile_cap(int tiles);
/* test aid: `blocks` (1..256) workgroups that each hold one compute unit's whole LDS for `microseconds` on `stream` -- a stand-in for a foreign
 * kernel (an RCCL collective of the data-parallel step, scripts/train.py:133-143 replaced by ha2g_amd/ddp.py) that is resident while a cluster
 * GRU launch needs its workgroups co-resident; the launch must then complete or set its error word, never hang.  sink: any device int. */
int ha2g_debug_occupy(int blocks, long microseconds, int* sink, void* stream);
/* dg [B][T][8H], a row = [d gi (r z n) of the forward direction | d gi (r z n) of the reverse direction | d gh_n forward | d gh_n reverse]: both
 * directions' input-gate gradients are contiguous, so dX = dg[:, :6H] [W_ih; W_ih_reverse] is ONE product (d gh = (d gi_r, d gi_z, d gh_n)); wpt = packed_bwd images (dir 0, dir 1); hp (nullable) [B][T][2H] receives
 * the h_prev each step used (y shifted by one step per direction, zero at the sequence ends) = the dW_hh GEMM's operand */
int ha2g_gru_layer_bwd(const float* dy, const float* y, const float* rs, const float* wpt, float* dg, float* hp, int B, int T,
                       int H, void* stream);

/* ---- BatchNorm (train mode) over [rows][C] channels-last data (nn.BatchNorm2d: model/ResNetBlocks.py:13,15,
 *      ResNetSE34V2.py:29,35,39,43,103; nn.BatchNorm1d: model/hierarchy_net.py:205,208) ---- */
long ha2g_bn_workspace_floats(int C);
int ha2g_bn_stats_f32(const float* x, long rows, int C, float* mean, float* invstd, float* running_mean,
                      float* running_var, float momentum, float eps, float* ws, void* stream);
/* the second half of ha2g_bn_stats_f32 on partial sums [2][C][nblk] (doubles: sum, sum of squares per block) a producer's epilogue wrote */
int ha2g_bn_stats_finalize_f32(const void* part, int nblk, long rows, int C, float* mean, float* invstd, float* running_mean,
                               float* running_var, float momentum, float eps, void* stream);
int ha2g_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      float* y, long rows, int C, int act, void* stream);
/* BatchNorm apply fused with the SE squeeze that follows bn2 (model/ResNetBlocks.py:29-36,81-83): y = bn(x) for x [N][HW][C]
 * and pooled[n][c] = mean_hw y in ONE pass over the tensor.  ws >= ha2g_bn_apply_pool_workspace_floats(N, HW, C). */
long ha2g_bn_apply_pool_workspace_floats(int N, int HW, int C);
int ha2g_bn_apply_pool_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                           int N, int HW, int C, float* pooled, float* ws, void* stream);
/* relu_mask = 1: x is a ReLU output feeding the BatchNorm (conv -> ReLU -> BN, ResNetBlocks.py:24-26); dx is then masked by x > 0 */
int ha2g_bn_bwd_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
                    float* dx, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma /*nullable: += */,
                    float* acc_dbeta /*nullable: += */, float* ws, void* stream);
/* the same with dx written as two bf16 planes dx_hi / dx_lo [rows][C] (hi = bf16(dx), lo = bf16(dx - hi)) for the plane-based split-bf16
 * convolution gradients below; dx (fp32) may be NULL when every consumer reads the planes */
int ha2g_bn_bwd_planes_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* dx_hi,
                           void* dx_lo, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                           float* ws, void* stream);
/* forward producers of the x operand of the plane-based weight gradient: ha2g_bn_apply_f32 / ha2g_se_scale_add_relu_f32 that write their output
 * a second time as bf16 hi / lo planes (model/ResNetBlocks.py:24-36: bn1's output feeds conv2, the block's output feeds the next block's conv1) */
int ha2g_bn_apply_planes_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y, void* y_hi,
                             void* y_lo, long rows, int C, int act, void* stream);
int ha2g_se_scale_add_relu_planes_f32(const float* x, const float* s, const float* res, float* out, void* o_hi, void* o_lo, int N, int HW, int C,
                                      void* stream);
/* ABI 2: the three producers above writing np = 2 or 3 equally spaced piece planes (piece q at planes + q * ps elements); bn_apply: y may be NULL
 * (planes only: three pieces reproduce the fp32 value exactly, and the next convolution reads nothing else) */
int ha2g_bn_bwd_planes_np_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* planes,
                              long ps, int np, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                              float* ws, void* stream);
/* ABI 5: BatchNorm-backward statistics out of the epilogue of the data gradient that produces the BatchNorm's dy (conv2's data gradient -> bn1,
 * model/ResNetBlocks.py:24-29 under autograd's backward, train_hierarchy.py:264).  ha2g_conv2d_dgrad_planes_stat_blocks: tiles per channel the launch
 * writes (0 = not served: keep ha2g_bn_bwd_planes_np_f32); ha2g_conv2d_dgrad_planes_np_bnstats_f32: the data gradient (beta = 0) + stat_part
 * [2][Cin][stat_nblk] doubles = tile sums of dx and dx * xhat (x_bn = the BatchNorm's input [N,H,W,Cin], its mean / invstd);
 * ha2g_bn_bwd_planes_np_partials_f32: dgamma / dbeta from those tiles + the apply pass -- no column pass over dx and x_bn. */
int ha2g_conv2d_dgrad_planes_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_dgrad_planes_np_bnstats_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W, int Cin, int Cout,
                                            int KH, int KW, int stride, int pad, const float* x_bn, const float* mean, const float* invstd, void* stat_part,
                                            int stat_nblk, void* stream);
int ha2g_bn_bwd_planes_np_partials_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* planes,
                                       long ps, int np, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                                       const void* stat_part, int stat_nblk, void* stream);
int ha2g_bn_apply_planes_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y, void* planes,
                                long ps, int np, long rows, int C, int act, void* stream);
int ha2g_se_scale_add_relu_planes_np_f32(const float* x, const float* s, const float* res, float* out, void* planes, long ps, int np, int N, int HW,
                                         int C, void* stream);
/* ---- squeeze-excite pointwise pieces (model/ResNetBlocks.py:81-95 and the residual tail :33-36) ---- */
int ha2g_hw_mean_f32(const float* x, float* out, int N, int HW, int C, void* stream);
int ha2g_se_scale_add_relu_f32(const float* x, const float* s, const float* res, float* out, int N, int HW, int C, void* stream);
/* SE tail of a block whose bn2 statistics came out of conv2's epilogue (model/ResNetBlocks.py:29-36,81-95): bn2's output is never materialised.
 * pooled = the squeeze from the per-tile column sums of c2 (tiles inside one image); out = relu(bn2(c2) * s + res) (+ np piece planes, np = 0: none);
 * ds = sum_hw dout (out > 0) bn2(c2) -- bn2(c2) recomputed per element with ha2g_bn_apply_pool_f32's expression (the same bits) */
int ha2g_bn_pool_from_partials_f32(const void* part, int nblk, int N, int HW, int C, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* pooled, void* stream);
int ha2g_se_bn_scale_add_relu_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* s,
                                     const float* res, float* out, void* planes, long ps, int np, int N, int HW, int C, void* stream);
int ha2g_se_bwd_scale_bn_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, void* stream);
int ha2g_se_bwd_scale_f32(const float* dout, const float* out, const float* x, float* ds, int N, int HW, int C, const float* gate, float* ws,
                          void* stream);   /* gate (nullable) [N][C]: ds is multiplied by gate * (1 - gate), the sigmoid derivative of the SE gate; */
     /* ws: ha2g_bn_apply_pool_workspace_floats(N, HW, C) floats (row-chunked partials) or null (one block per image) */
/* data path of the SE excitation MLP's backward (ResNetBlocks.py:84-89 under autograd) in one launch: dh1 = relu'(h1) * (dsc w2), dpool = inv_hw * dh1 w0;
 * w2 = se.fc.2.weight [C][R], w0 = se.fc.0.weight [R][C]; C <= 256, R <= 32 */
/* forward of the SE excitation MLP in one launch (model/ResNetBlocks.py:84-89): h1 = relu(fc.0(pooled)), sc = sigmoid(fc.2(h1)); the squeeze
 * `pooled` is given (pooled_in) or taken from the per-tile column sums of bn2's input (stat_part: see ha2g_bn_pool_from_partials_f32; pooled_out
 * receives it).  Widths as ha2g_se_mlp_bwd_supported. */
int ha2g_se_mlp_fwd_f32(const float* pooled_in, const void* stat_part, int nblk, int HW, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* w0, const float* b0, const float* w2, const float* b2, float* pooled_out, float* h1,
                        float* sc, int N, int C, int R, void* stream);
void ha2g_bn_debug_rows_per_trip(int n);     /* A/B (ABI 5): rows per trip (2, 4 = default, 8) of the BatchNorm-backward statistics pass; bit-identical sums */
int ha2g_se_mlp_bwd_supported(int C, int R);
int ha2g_se_mlp_bwd_f32(const float* dsc, const float* h1, const float* w2, const float* w0, float* dh1, float* dpool, int N, int C, int R,
                        float inv_hw, void* stream);
/* weight and bias gradients of the SE excitation MLP, ACCUMULATED into their gradient buffers, in one launch (ABI 5; autograd's addmm backward + sum(0)
 * of the two nn.Linear of model/ResNetBlocks.py:84-89): dw2 [C][R] += dsc^T h1, db2 [C] += column sums of dsc, dw0 [R][C] += dh1^T pooled, db0 [R] += column
 * sums of dh1; dsc [N][C], h1 / dh1 [N][R], pooled [N][C].  Replaces two ha2g_gemm_wgrad_bias_f32 launches per block. */
int ha2g_se_mlp_wgrad_f32(const float* dsc, const float* h1, const float* dh1, const float* pooled, float* dw2, float* db2, float* dw0, float* db0,
                          int N, int C, int R, void* stream);
/* ABI 6: ha2g_se_mlp_wgrad_f32 of n <= 16 blocks in ONE launch (HOST arrays of n device pointers / widths; every block's batch is N images): the tower's
 * sixteen SE layers after its backward is enqueued instead of sixteen small launches between the convolutions' weight gradients. */
int ha2g_se_mlp_wgrad_multi_f32(int n, const void* const* dsc, const void* const* h1, const void* const* dh1, const void* const* pooled, void* const* dw2,
                                void* const* db2, void* const* dw0, void* const* db0, const int* C, const int* R, int N, void* stream);
/* ABI 5: ha2g_se_bwd_scale[_bn]_f32 + ha2g_se_mlp_bwd_f32 as two launches instead of three (the reduction's final pass runs inside the MLP launch; ds is
 * still written).  mean == NULL: x = bn2's output; else x = bn2's input and bn2 is applied on the fly (mean / invstd / gamma / beta). */
int ha2g_se_bwd_scale_mlp_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, const float* h1, const float* w2,
                              const float* w0, float* dh1, float* dpool, int R, void* stream);
int ha2g_se_bwd_apply_f32(const float* dout, const float* out, const float* s, const float* dpool, float* dres,
                          float* dx, int N, int HW, int C, void* stream);
/* ABI 6: the SE backward and bn2's backward of a block (model/ResNetBlocks.py:30-37 under autograd) without the gradient tensor between them.  With
 * dpre = dout * (out > 0), bn2's dy is dz = dpre * s[n,c] + dpool[n,c]; its column sums follow from per-image sums the SE reduction pass takes anyway
 * (sum dpre, sum dpre * xhat, sum xhat), so ha2g_se_bwd_apply_f32's write of dz and ha2g_bn_bwd*'s statistics pass over (dz, x) are not run.
 * reduce_mlp: x = bn2's INPUT [N][HW][C], mean / invstd / gamma / beta = bn2's, gate = the SE gate [N][C], ws >= ha2g_se_bn_bwd_workspace_floats floats;
 *   writes ds [N][C], dh1 [N][R], dpool [N][C] (already / HW) as ha2g_se_bwd_scale_mlp_f32 does, and stat [2][C][N] doubles (bn2's per-image sums).
 * apply: dres = dpre; bn2's data gradient as fp32 (dx, nullable when planes != NULL) and / or np (2 | 3) bf16 piece planes (piece stride ps elements);
 *   dgamma / dbeta [C] = bn2's parameter gradients, also ADDED to acc_dgamma / acc_dbeta when those are not NULL.
 * mask_bits (both; nullable): the ReLU decisions as ha2g_se_bn_scale_add_relu_mask_np_f32 wrote them. */
/* ABI 6: per-IMAGE statistics partials of x [N][HW][C] (sum x, sum x^2; doubles [2][C][nblk], nblk = N * ha2g_bn_image_partial_chunks(N, HW), the blocks of
 * an image consecutive): what a plane convolution's statistics epilogue leaves behind, for a tensor whose producer has none (layer 1's 32-channel
 * convolutions) -- ha2g_bn_stats_finalize_f32 and ha2g_se_mlp_fwd_f32 (the SE squeeze) read them, bn2's output is then never materialised. */
int ha2g_bn_image_partial_chunks(int N, int HW);
int ha2g_bn_image_partials_f32(const float* x, int N, int HW, int C, double* part, void* stream);
/* ABI 6: the data gradient of a block's conv1 (3x3, stride 1; model/ResNetBlocks.py:21-37 under autograd) with the identity shortcut's gradient added in the
 * epilogue: dx = conv_transpose(dy, w) + (decision bit ? resid : 0), resid = the gradient of the block's output, resid_bits = the ReLU decisions
 * ha2g_se_bn_scale_add_relu_mask_np_f32 left -- what beta = 1 onto a materialised dres = resid * (out > 0) computed, bit for bit, without writing dres
 * (ha2g_se_bn_bwd_apply_np_f32 then takes dres = NULL).  *_planes_*: layers 2-4 on the patch-resident plane kernel; the other: layer 1 (32 channels). */
int ha2g_conv2d_dgrad_planes_resid_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_dgrad_planes_np_resid_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad, const float* resid, const void* resid_bits, void* stream);
int ha2g_conv2d_dgrad_resid_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ha2g_conv2d_dgrad_resid_f32(const float* dy, const float* wt, float* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                const float* resid, const void* resid_bits, void* stream);
long ha2g_se_bn_bwd_workspace_floats(int N, int HW, int C);
void ha2g_bn_debug_small_lds(int on);   /* the backward statistics passes' reduction: 0 = through 16-24 KB of LDS; 1 (default) = C = 32: shuffles + 4-6 KB, wider: one partial per wave, no LDS; 2 = the small form for C = 32 only.  A/B */
void ha2g_se_bn_debug(int rows_per_trip, int chunk_shift);   /* A/B of the reduction pass (4 | 2 rows of loads in flight; chunks per image >> chunk_shift); default (4, 0) */
int ha2g_se_bn_bwd_reduce_mlp_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, const float* h1, const float* w2,
                                  const float* w0, float* dh1, float* dpool, int R, double* stat, const void* mask_bits, void* stream);
int ha2g_se_bn_bwd_apply_np_f32(const float* dout, const float* out, const float* x, const float* s, const float* dpool, const float* mean,
                                const float* invstd, const float* gamma, float* dres, float* dx, void* planes, long ps, int np, float* dgamma, float* dbeta,
                                float* acc_dgamma, float* acc_dbeta, const double* stat, int N, int HW, int C, const void* mask_bits, void* stream);
/* ha2g_se_bn_scale_add_relu_np_f32 that also leaves the block's ReLU decisions (out > 0) behind as bits: mask_bits = N * HW * C / 32 32-bit words, four bits
 * per 16-byte vector, eight vectors per word (C % 32 == 0).  ha2g_se_bn_bwd_reduce_mlp_f32 / _apply_np_f32 with mask_bits != NULL read them instead of
 * `out` (which may then be NULL): a 32nd of the bytes, twice per block. */
int ha2g_se_bn_scale_add_relu_mask_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* s,
                                          const float* res, float* out, void* planes, long ps, int np, int N, int HW, int C, void* mask_bits, void* stream);
/* speaker-softmax blending of the three audio taps (model/ResNetSE34V2.py:202-212) */
int ha2g_blend_fwd_f32(const float* logits, const float* f0, const float* f1, const float* f2, float* w, float* blend,
                       int B, int L, int TF, void* stream);
int ha2g_blend_bwd_f32(const float* dblend, const float* dw_ext, const float* w, const float* f0, const float* f1,
                       const float* f2, float* df0, float* df1, float* df2, float* dlogits, int B, int L, int TF, void* stream);

/* ---- text encoder pieces (model/hierarchy_net.py:48-52, model/tcn.py:16-46, torch weight_norm) ---- */
int ha2g_embedding_fwd_f32(const long* tok, const float* W, float* out, long n, int C, void* stream);
/* dW += ; heavy = most frequent id (the padding id 0), summed by a 2-level masked column sum; ws >= 64*C doubles */
int ha2g_embedding_bwd_f32(const long* tok, const float* dY, float* dW, int n, int C, long heavy, float* ws, void* stream);
int ha2g_im2col1d_f32(const float* x, float* col, int B, int T, int C, int k, int dil, int pad_left, int To, void* stream);
int ha2g_col2im1d_f32(const float* dcol, float* dx, int B, int T, int C, int k, int dil, int pad_left, int To, void* stream);
int ha2g_weight_norm_fwd_f32(const float* g, const float* v, float* w, float* norm, int Cout, int n, void* stream);
/* dg / dv = beta * (old) + gradient: beta = 1 accumulates straight into the parameters' gradient buffers */
int ha2g_weight_norm_bwd_f32(const float* dw, const float* g, const float* v, const float* norm, float* dg, float* dv,
                             int Cout, int n, float beta, void* stream);
/* up to 32 weight-norm layers of one [Cout][n] shape in one launch (host arrays of device pointers): the convolutions of the generators'
 * text encoders (model/tcn.py:19-31, one torch.nn.utils.weight_norm per conv) are parameters-only work at the head of every step */
int ha2g_weight_norm_multi_fwd_f32(int count, const float* const* g, const float* const* v, float* const* w, float* const* norm, int Cout, int n,
                                   void* stream);
int ha2g_weight_norm_multi_bwd_f32(int count, const float* const* dw, const float* const* g, const float* const* v, const float* const* norm,
                                   float* const* dg, float* const* dv, int Cout, int n, float beta, void* stream);

/* ---- pointwise / RNG ---- */
/* op: 0 a+b, 1 a*b, 2 relu(a+b), 3 relu', 4 leaky', 5 sigmoid', 6 elu, 7 elu', 8 reparam (model/embedding_net.py:10-13),
 *     9 reparam d/dlogvar, 10 alpha*a+beta*b, 11 leaky, 12 relu, 13 alpha*a, 14 a*b[0]*alpha (b = device scalar),
 *     15 sigmoid(a), 16 a*sigmoid(b)*sigmoid(-b) (sigmoid' from the pre-activation b),
 *     17 1/sqrt(a+alpha), 18 leaky-relu with slope alpha */
int ha2g_eltwise_f32(int op, const float* a, const float* b, const float* c, float* out, long n, float alpha, float beta, void* stream);
/* out[r][h] = y[r][h] + y[r][H+h] (sum of the two GRU directions, model/hierarchy_net.py:145); inverse=1: gradient fan-out */
int ha2g_dirsum_f32(const float* y, float* out, long rows, int H, int inverse, void* stream);
/* Philox4x32-10 dropout; state = device uint64[2] {seed, step}; out and/or mask (pre-scaled keep mask) may be null */
int ha2g_dropout_f32(const float* x, float* out, float* mask, long n, float p, const void* state, unsigned stream_id, void* stream);
/* the same dropout fused with its neighbour in a TCN block (ABI 5; model/tcn.py:21-31,44-46): mode 1: out = relu(x * mask + b) (dropout -> + residual -> ReLU);
 * mode 2: out = b > 0 ? x * mask : 0 (backward: dropout' then the ReLU' of the convolution in front of it, b = that convolution's output).  The mask is the
 * one ha2g_dropout_f32 draws for the same (state, stream_id, element). */
int ha2g_dropout_fused_f32(const float* x, const float* b, float* out, long n, float p, const void* state, unsigned stream_id, int mode, void* stream);
/* out = x * mask over elements [elem_offset, elem_offset + n) of the tensor the mask of (state, stream_id) is defined on (ABI 5): the backward of the GRU's
 * inter-layer dropout (nn.GRU(dropout=...), model/hierarchy_net.py:87) when only a row slice carries gradient; elem_offset % 4 == 0 */
int ha2g_dropout_slice_f32(const float* x, float* out, long n, long elem_offset, float p, const void* state, unsigned stream_id, void* stream);
/* ABI 6: ha2g_im2col1d_f32 of dropout(x) without the dropped tensor (model/tcn.py:21-31: conv1 -> ReLU -> dropout -> conv2's columns): the mask
 * ha2g_dropout_f32 draws for (state, stream_id) over x [B][T][C] is re-drawn for both taps.  k = 2 and C % 4 == 0 only (ha2g_im2col1d_drop_supported). */
int ha2g_im2col1d_drop_supported(int C, int k);
int ha2g_im2col1d_drop_f32(const float* x, float* col, int B, int T, int C, int k, int dil, int pad_left, int To, float p, const void* state,
                           unsigned stream_id, void* stream);
int ha2g_rng_advance(void* state, void* stream);

/* ---- generator input pack + hierarchy scatter (train_eval/train_hierarchy.py:153-169, expressive :163-212;
 *      model/hierarchy_net.py:121-141) ----
 * pre_seq [rows][T][P+1]: frames < n_pre = (target frame, constraint bit 1); frames >= n_pre = columns of the coarser level's
 * output `prev` [rows][T][Pprev] through map[P+1] (int32, -1 = stays 0; built from the reference's slice assignments in order,
 * so the expressive step's one-column shift of the head values is reproduced).  prev/map may be null (first level).
 * Backward: inv[Pprev][2] = the (at most two) pre_seq columns each output column feeds (-1 = none). */
int ha2g_pre_seq_fwd_f32(const float* target, const float* prev, const int* map, float* out, long rows, int T, int P, int Pprev,
                         int n_pre, void* stream);
int ha2g_pre_seq_bwd_f32(const float* dpre, const int* inv, float* dprev, long rows, int T, int P, int Pprev, int n_pre, void* stream);
/* in_data [rows][T][Wa+Wb+Wc+Wz] = [a | b | c | z[row] expanded over T]; backward splits the gradient (da/db/dc/dz nullable) */
int ha2g_gen_concat_fwd_f32(const float* a, const float* b, const float* c, const float* z, float* out, long rows, int T, int Wa, int Wb,
                            int Wc, int Wz, void* stream);
int ha2g_gen_concat_bwd_f32(const float* d, float* da, float* db, float* dc, float* dz, long rows, int T, int Wa, int Wb, int Wc, int Wz,
                            void* stream);
/* sliding-window synthesis, smoothing of the motion transition (scripts/synthesize_hierarchy.py:150-161): window `index` ([T][P]) is
 * written into out_all [frames][P] at frame index*(T-n_pre); its first n_pre frames are cross-faded with what the previous window left
 * there:  out = prev*(n-j)/(n+1) + next*(j+1)/(n+1), j = 0..n-1 */
int ha2g_window_blend_f32(const float* win, float* out_all, int index, int T, int n_pre, int P, void* stream);
/* total loss = sum_i w_i * term_i (train_hierarchy.py:226-262) in one launch, left-to-right fp32 like the Python expression.
 * terms_host / weights_host are HOST arrays (n <= 24) of device scalar pointers / weights, read at call time. */
int ha2g_weighted_sum_f32(const void* const* terms_host, const float* weights_host, int n, float* out, void* stream);
int ha2g_weighted_sum_bwd_f32(const float* weights_host, int n, const float* g, float* out, void* stream);

/* ---- loss terms (train_eval/train_hierarchy.py): value + unit gradient in one pass ---- */
int ha2g_sum_f32(const float* x, long n, float* out, float scale, int accumulate, void* stream);
int ha2g_huber_f32(const float* x, const float* y, long n, float beta, float* loss, float* dx, float* ws, void* stream);      /* :173-176 */
int ha2g_kld_f32(const float* mu, const float* logvar, int n, float* loss, float* dmu, float* dlogvar, void* stream);          /* :226 */
int ha2g_divreg_f32(const float* out, const float* rnd, const float* z, const float* zr, int B, int TP, int Z, float beta,
                    float* loss, float* dout, float* ws, void* stream);                                                          /* :213-222 */
/* palm_host: HOST int32 [npalm][2] bone pairs whose raw cross products are appended as bones nb, nb+1, ... (expressive :430-432) */
int ha2g_phys_angle_f32(const float* out, const float* mean_dir, int rows, int nb, const int* pairs, int npairs,
                        const float* avg, const float* var, const int* palm_host, int npalm, float* loss, float* dout,
                        float* ws, void* stream);                                                                                /* :242-262 */
int ha2g_gan_loss_f32(int mode, const float* a, const float* b, int n, float* loss, float* da, float* db, void* stream);       /* :128,180 */
long ha2g_contrastive_workspace_floats(int N);
int ha2g_contrastive_f32(const float* a, const float* b, int N, int expressive, float* loss, float* da, float* db, float* ws,
                         void* stream);                                                                                          /* :54-68 */

/* ---- FGD evaluator statistics (model/embedding_space_evaluator.py:57-154), accumulated on the device in float64 ----
 * feat_stats: sum[D] += sum_n f[n][:], outer[D][D] += sum_n f[n] f[n]^T (D <= 128): mean / covariance of the latent features (np.mean, np.cov);
 * l1_rows: out[0] += sum_n sum_d |a[n][d] - b[n][d]| (feat_dist :139-144, diversity :112-123);
 * recon_metrics: out[0] += sum_b ( mean_{t,d}|r-p| + mean_{t<T-1,d}|(r[t+1]-r[t]) - (p[t+1]-p[t])| ), out[1] += sum_{b,t,bone} (1 - cos(r_bone, p_bone))
 *                (:78-103; poses [B][T][P], P = 3 * bones, cosine_similarity eps 1e-8) */
int ha2g_feat_stats_f64(const float* f, int N, int D, double* sum, double* outer, void* stream);
int ha2g_l1_rows_f64(const float* a, const float* b, long n, double* out, void* stream);
int ha2g_recon_metrics_f64(const float* recon, const float* poses, int B, int T, int P, double* out /* 2 + 2*B doubles: [0..1] accumulators, rest scratch */, void* stream);

/* ---- optimizer (torch.optim.Adam as set up in train.py:155-170) ---- */
int ha2g_adam_step_inc(int* step, void* stream);
/* lr, betas, eps are doubles like torch's Python scalars: 1 - beta, the bias corrections 1 - beta^step and lr / bc1 are formed
 * in double (in fp32, 1 - 0.999 alone is off by 1.3e-5 relative), the per-element update runs in fp32 */
int ha2g_adam_f32(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                  const int* step, void* stream);
/* ABI 2: the same with a GUARD word (device int32, may be NULL): while *guard != 0 the step counter, the parameters and the moments are left
 * untouched.  The train step passes the cluster-GRU error word (ha2g_gru_layer_*_cluster's `err`), so a step whose recurrences timed out is a
 * no-op on the optimizer state instead of an update from garbage gradients (optimizer.step() of train_hierarchy.py:131,270-274). */
int ha2g_adam_step_inc_guarded(int* step, const int* guard, void* stream);
int ha2g_adam_guarded_f32(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                          const int* step, const int* guard, void* stream);

/* ---- sparse embedding gradients + lazy row-wise Adam (tables: model/hierarchy_net.py:31-34, optimizer train.py:155-170) ----
 * unique_tokens: tok [n] int64 -> uniq (slot 0 = the padding id 0, then distinct ids in order of first occurrence), remap [n] (slot of each
 *   position), count (device int32); map = int32 [n_rows] scratch that is INT_MAX on entry and exit, cpos = int32 [n] scratch.  The compact row
 *   gradient is then ha2g_embedding_bwd_f32(remap, dY, vals [count][C], heavy = 0).
 * adam_scalars: table[*step] = {lr/(1-b1^t), 1/sqrt(1-b2^t)} exactly as ha2g_adam_f32's prologue computes them (float2 table of `cap` entries).
 * sparse_adam: for the `*count` distinct rows `ids` (grid = max_rows): replay the zero-gradient Adam updates the row missed since last[row]
 *   (dense Adam moves EVERY row every step), then, when vals != NULL, apply step *step with gradient row vals[r]; vals == NULL = catch-up only
 *   (before an embedding read).  Bit-identical to ha2g_adam_f32 on a dense gradient with zeros elsewhere.  Steps >= table_steps (a captured
 *   graph replays past any host-side count) take their scalars from the same double-precision formula with `lr` instead of the table; the
 *   catch-up of a row is one serial chain over every step it missed (dense-Adam semantics), i.e. O(steps since the row was last touched). */
int ha2g_unique_tokens(const long* tok, int n, int* map, int* cpos, long* uniq, long* remap, int* count, void* stream);
int ha2g_adam_scalars(const int* step, double lr, double b1, double b2, void* table, int cap, void* stream);
int ha2g_sparse_adam_f32(float* W, float* M, float* V, int* last, const long* ids, const int* count, int max_rows, const float* vals,
                         const void* table, const int* step, int C, double b1, double b2, double eps, int table_steps, double lr, void* stream);
/* ABI 2: ha2g_sparse_adam_f32 under a new name (its signature had grown in round 3 under the old one) + the guard word of ha2g_adam_guarded_f32:
 * a flagged step applies no real update; catch-ups (vals == NULL) replay valid earlier steps and run regardless.  Past `table_steps` replayed
 * steps take `lr` as it is NOW: exact while lr is constant (train.py:155-170 never changes it). */
int ha2g_sparse_adam2_f32(float* W, float* M, float* V, int* last, const long* ids, const int* count, int max_rows, const float* vals,
                          const void* table, const int* step, int C, double b1, double b2, double eps, int table_steps, double lr, const int* guard,
                          void* stream);
int ha2g_iota_ids(long* ids, int* count, int n, void* stream);

/* ---- log-mel front-end on the GPU: replaces the offline librosa step
 *      scripts/utils/data_utils.py:34-38 extract_melspectrogram / dataset_script/script/make_ted_dataset.py:121-123:
 *      melspectrogram(y, sr=16000, n_fft=1024, hop_length=512, power=2) -> power_to_db(ref=max over the clip) -> float16.
 *      STFT = fp32 MFMA GEMM of the centre-padded clip (lda = 512: overlapping frames, nothing materialised) against a
 *      Hann-windowed DFT basis; 128 Slaney mel filters = second GEMM.  PARITY UNPINNED (librosa is not vendored by the
 *      reference and absent here): checked against oracle/logmel_oracle.py, a restatement of librosa's published defaults. ---- */
long ha2g_logmel_tables_floats(void);                      /* DFT basis [1026][1024] + mel filters [128][516] */
int ha2g_logmel_frames(long n_samples);                    /* 1 + n / 512 (center=True) */
int ha2g_logmel_init_f32(float* tables, int sr, void* stream);
long ha2g_logmel_workspace_floats(int B, long n_samples);
/* y [B][n] -> out [B][128][frames] dB in [-80, 0]; pad_reflect: 1 = librosa <= 0.9 default ('reflect'), 0 = zeros (>= 0.10);
 * round_f16: 1 = round through float16 like the reference's stored arrays */
int ha2g_logmel_f32(const float* y, int B, long n_samples, int pad_reflect, const float* tables, int round_f16, float* out,
                    float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif
