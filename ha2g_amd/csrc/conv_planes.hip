// Split-bf16 matrix products on PRE-SPLIT operands ("planes"): round 3.
//
// The split-bf16 inner product of gemm.hip (x = hi + lo, hi = bf16(x), lo = bf16(x - hi); a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate) split every operand element in EVERY consumer tile at LDS-staging time.  rocprofv3 counters of
// those kernels (profiles/r03_pmc_bwd_gemm.txt): 15-18 VALU + 6-13 SALU instructions per MFMA, matrix pipe 19-23 % busy -- instruction-issue
// bound, not MFMA / LDS / HBM bound.  Here the PRODUCER of a tensor (the BatchNorm-backward apply pass, the weight re-layout kernel) writes
// it once as two bf16 planes, and the consumer moves 16-byte pieces of those planes straight from global memory into LDS with
// global_load_lds_dwordx4 (no staging registers, no conversion, ~1.5 VALU per MFMA left for addresses).  Same hi / lo values, same MFMA
// order per accumulator and the same k order as gemm_x3_kernel<.., NP = 2> ==> results are BIT-IDENTICAL to the kernels they replace
// (tests/test_gpu_planes.py asserts torch.equal), so no parity fixture moves.
//
// LDS image: per operand and plane [rows][32 bf16] = 64-byte rows, the four 16-byte pieces of a row XOR-swizzled by (row >> 2) & 3 so that
// the 16-lane groups of a ds_read_b128 (guide, LDS section) touch 16 distinct 16-byte slots: conflict-free fragment reads.  The DMA writes
// LDS lane-linearly (wave-uniform base + lane * 16), so the swizzle is applied to the per-lane SOURCE address (guide 5.4 rule 21).
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// zeros: the source of every masked-out 16-byte piece (zero padding of the convolution, rows past M)
__device__ __attribute__((aligned(64))) unsigned short g_zero_page[64];

// piece q of element i -> pl[q * ps + i]   (NP = 2: the hi / lo planes of round 3; NP = 3: all 24 mantissa bits, split3_bf16)
template <int NP>
__global__ void f32_to_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ pl, long ps, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        unsigned a[NP], b[NP];
        splitn_bf16<NP>(v.x, v.y, a); splitn_bf16<NP>(v.z, v.w, b);
#pragma unroll
        for (int q = 0; q < NP; ++q) reinterpret_cast<uint2*>(pl + q * ps)[i] = make_uint2(a[q], b[q]);
    }
}

// the same for up to 48 tensors in ONE launch (blockIdx.y = tensor): the forward weights of every plane-served convolution of the tower
struct SplitBatch { const float* x[48]; unsigned short* pl[48]; long ps[48]; long n4[48]; };
template <int NP>
__global__ void f32_to_planes_multi_kernel(SplitBatch b) {
    const int t = blockIdx.y;
    const float* __restrict__ x = b.x[t];
    unsigned short* __restrict__ pl = b.pl[t];
    const long ps = b.ps[t], n4 = b.n4[t];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        unsigned a[NP], c[NP];
        splitn_bf16<NP>(v.x, v.y, a); splitn_bf16<NP>(v.z, v.w, c);
#pragma unroll
        for (int q = 0; q < NP; ++q) reinterpret_cast<uint2*>(pl + q * ps)[i] = make_uint2(a[q], c[q]);
    }
}

// Dense operands of the plane GEMM (plane_gemm_launch): x [rows][ldx] fp32, `cols` valid columns -> NP piece planes [rows][ldp] with zeros in
// [cols, ldp) (ldp = cols rounded up to 32: whole k tiles) ...
template <int NP>
__global__ void f32_to_planes_pad_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ pl, long ps, long ldp, long rows, int cols) {
    const long q4 = ldp >> 2, total = rows * q4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / q4; const int c = (int)(i - r * q4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = x + r * ldx + c;
        if (c + 3 < cols) v = *reinterpret_cast<const float4*>(src);
        else { if (c < cols) v.x = src[0]; if (c + 1 < cols) v.y = src[1]; if (c + 2 < cols) v.z = src[2]; }
        unsigned a[NP], b[NP];
        splitn_bf16<NP>(v.x, v.y, a); splitn_bf16<NP>(v.z, v.w, b);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(pl + q * ps + r * ldp + c) = make_uint2(a[q], b[q]);
    }
}
// ... and the TRANSPOSED form: x [rows][ldx] -> planes [cols][ldp] with element (c, r) = x[r][c], zeros in [rows, ldp) (the operands of dW = dY^T X
// and the weight of dX = dY W arrive with k as their slow dimension).  32 x 32 tiles through LDS; grid = (ceil(ldp / 32), ceil(cols / 32)).
template <int NP>
__global__ __launch_bounds__(256) void f32_to_planes_t_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ pl, long ps, long ldp,
                                                              long rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows of 32 threads
    const long r0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long r = r0 + ty + 8 * j; const int c = c0 + tx;
        tile[ty + 8 * j][tx] = (r < rows && c < cols) ? x[r * ldx + c] : 0.f;
    }
    __syncthreads();
    // output row = c0 + (thread / 8), four output columns (= input rows) r0 + 4 (thread % 8) ..: 8-byte plane stores
    const int oc = threadIdx.x >> 3, or4 = (threadIdx.x & 7) << 2;
    if (c0 + oc < cols && r0 + or4 < ldp) {
        unsigned a[NP], b[NP];
        splitn_bf16<NP>(tile[or4][oc], tile[or4 + 1][oc], a); splitn_bf16<NP>(tile[or4 + 2][oc], tile[or4 + 3][oc], b);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(pl + q * ps + (long)(c0 + oc) * ldp + r0 + or4) = make_uint2(a[q], b[q]);
    }
}

// Round 6: up to 16 of those 2-D splits in ONE launch (blockIdx.z = job), each into its own window of a shared plane buffer -- the B operands of a GRU
// stack's merged input projections [W_ih; W_ih_reverse] (rows 0..3H-1 / 3H..6H-1 of a [6H][Kp] plane set) and of its dX products (the transposed form:
// columns 0..3H-1 / 3H..6H-1 of a [K][round_up(6H, 32)] plane set) were a torch.cat + a split launch per layer.
// Job: x [rows][ldx] -> planes window at pl (piece q at pl + q * ps), row stride ldp.  tr = 0: `cols` valid columns, zeros in [cols, wcols);
// tr = 1: element (c, r) = x[r][c] for c < cols, r < rows, zeros in [rows, wcols) (wcols = the window's width in elements, a multiple of 4).
struct Split2dJobs { const float* x[16]; unsigned short* pl[16]; long ldx[16]; int rows[16], cols[16], wcols[16]; long ps, ldp; int tr; };
template <int NP>
__global__ __launch_bounds__(256) void f32_to_planes_2d_multi_kernel(Split2dJobs jb) {
    const int j = blockIdx.z;
    const float* __restrict__ x = jb.x[j];
    unsigned short* __restrict__ pl = jb.pl[j];
    const long ldx = jb.ldx[j], ps = jb.ps, ldp = jb.ldp;
    const int rows = jb.rows[j], cols = jb.cols[j], wcols = jb.wcols[j];
    if (!jb.tr) {
        const int q4 = wcols >> 2;
        const long total = (long)rows * q4;
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < total; i += (long)gridDim.x * gridDim.y * 256) {
            const long r = i / q4; const int c = (int)(i - r * q4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* src = x + r * ldx + c;
            if (c + 3 < cols) v = *reinterpret_cast<const float4*>(src);
            else { if (c < cols) v.x = src[0]; if (c + 1 < cols) v.y = src[1]; if (c + 2 < cols) v.z = src[2]; }
            unsigned a[NP], b[NP];
            splitn_bf16<NP>(v.x, v.y, a); splitn_bf16<NP>(v.z, v.w, b);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(pl + q * ps + r * ldp + c) = make_uint2(a[q], b[q]);
        }
        return;
    }
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long r0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
    if (r0 >= wcols || c0 >= cols) return;                       // (block-uniform)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long r = r0 + ty + 8 * k; const int c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? x[r * ldx + c] : 0.f;
    }
    __syncthreads();
    const int oc = threadIdx.x >> 3, or4 = (threadIdx.x & 7) << 2;
    if (c0 + oc < cols && r0 + or4 < wcols) {
        unsigned a[NP], b[NP];
        splitn_bf16<NP>(tile[or4][oc], tile[or4 + 1][oc], a); splitn_bf16<NP>(tile[or4 + 2][oc], tile[or4 + 3][oc], b);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(pl + q * ps + (long)(c0 + oc) * ldp + r0 + or4) = make_uint2(a[q], b[q]);
    }
}

// w [Cout][KK][Cin] fp32 (OHWI) -> planes of wt [Cin][KK][Cout]: the B operand of the data gradient, rows = input channels, k = (tap, cout)
__global__ void weight_ihwo_planes_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int Cout,
                                          int KK, int Cin) {
    const long total = (long)Cout * KK * Cin / 2;                       // two consecutive cout per thread
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long e = 2 * i;
        const int co = (int)(e % Cout); const long t = e / Cout; const int kk = (int)(t % KK); const int ci = (int)(t / KK);
        const float a = w[((long)co * KK + kk) * Cin + ci], b = w[((long)(co + 1) * KK + kk) * Cin + ci];
        unsigned h, l;
        split2_bf16(a, b, h, l);
        reinterpret_cast<unsigned*>(hi)[i] = h;
        reinterpret_cast<unsigned*>(lo)[i] = l;
    }
}

// The same for up to 48 weights in ONE launch (blockIdx.y = tensor): the tower's backward used to run one re-layout launch per convolution on
// its critical path (29 per step), each a few microseconds of kernel and a level of the replayed graph.
constexpr int WPB_MAX = 48;
struct WPlanesBatch {
    const float* w[WPB_MAX]; unsigned short* pl[WPB_MAX]; long ps[WPB_MAX];      // piece q of tensor t at pl[t] + q * ps[t]
    int cout[WPB_MAX], kk[WPB_MAX], cin[WPB_MAX];
};
template <int NP>
__global__ void weight_ihwo_planes_multi_kernel(WPlanesBatch b) {
    const int t = blockIdx.y;
    const float* __restrict__ w = b.w[t];
    const int Cout = b.cout[t], KK = b.kk[t], Cin = b.cin[t];
    const long total = (long)Cout * KK * Cin / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long e = 2 * i;
        const int co = (int)(e % Cout); const long q = e / Cout; const int kk = (int)(q % KK); const int ci = (int)(q / KK);
        const float x = w[((long)co * KK + kk) * Cin + ci], y = w[((long)(co + 1) * KK + kk) * Cin + ci];
        unsigned pc[NP];
        splitn_bf16<NP>(x, y, pc);
#pragma unroll
        for (int k = 0; k < NP; ++k) reinterpret_cast<unsigned*>(b.pl[t] + k * b.ps[t])[i] = pc[k];
    }
}

// One PARITY CLASS of output pixels.  Stride 1: a single class, all taps.  Stride 2: the output pixels (oy, ox) with (oy % 2, ox % 2) = (py, px)
// receive contributions only from the taps with (oy + pad - kh) and (ox + pad - kw) even -- 1, 2, 2 and 4 of the nine 3x3 taps for the four
// classes (ONE of the four classes for a 1x1 kernel) -- so each class is its own implicit GEMM over ITS taps only.  The round-2 kernels ran all
// nine taps for every pixel with a validity mask: 4x the matrix work of a stride-2 data gradient.  The taps of a class are listed in
// ascending order and their source offsets are affine (first tap's pixel minus ((kh - kh0) / s) rows and ((kw - kw0) / s) columns), so the k loop
// is the stride-1 loop over a shorter tap list; the non-zero products of every output element arrive in the same order as before.
struct PClass {
    int py, px, OHc, OWc, M;                                     // parity, class grid, pixels of the class (over all images)
    int ntaps, kh0, kw0;                                         // taps of the class; first valid kh / kw
    int tap[9];                                                  // filter tap index kh * KW + kw
    int doff[9];                                                 // source offset of the tap relative to the first one, in pixels of the dy grid (<= 0)
    unsigned char khs[12], kws[12];                              // pconv_q_kernel: (kh, kw) of each tap (host-filled: no division by KW per tap and lane)
    unsigned mg_ow, mg_oh;                                       // pconv_q_kernel: ceil(2^32 / OWc), ceil(2^32 / OHc) (0 when the divisor is 1): fast_div
};
struct PConvP {
    PlaneSet a;                                                  // dy planes [img][GH][GW][GC] (piece q at a.p + q * a.ps)
    PlaneSet b;                                                  // weight planes [N][K], K = KH*KW*GC
    float* C; long ldc; float beta;
    const float* rsd; const unsigned* rsd_bits;      // masked residual of the patch-resident kernel's epilogue (ha2g_conv2d_dgrad_planes_np_resid_f32), or null
    int N, K;
    int GH, GW, GC, OH, OW, KH, KW, pad, stride;                 // gathered tensor (dy; forward: x) / output pixel grid (dx; forward: y)
    int fwd, relu;                                               // forward convolution instead of the data gradient; ReLU on the bf16 output
    int ncls;
    PClass cls[4];
    int dbg;                                                     // timing ablations (ha2g_conv_planes_debug): 1 = no DMA after tile 0, 2 = no MFMA
    // dense use (plane_gemm_launch: C = A B^T as a 1x1 "convolution" over M pixels; pconv_q_kernel only)
    const float* bias; int act;                                  // epilogue: + bias[col], act 0 none / 1 relu (= relu) / 2 leaky-relu(0.01)
    int vec;                                                     // pconv_q_kernel: 16-byte output stores allowed (N % 4 == 0, ldc % 4 == 0, aligned pointers)
    int buf, a_bytes, b_bytes;                                   // pconv_q_kernel: buffer-addressed DMA (ha2g_conv_planes_bufaddr) and the byte size of one piece plane of A / B
    int kmaj;                                                    // pconv_q_kernel: 1 = channel-major k order (default), 0 = tap-major (ha2g_conv_planes_korder)
    int ksplit, kt_per;                                          // split-K over blockIdx.z (ncls == 1): k tiles [z * kt_per, ..) -> raw partial slab z of ws
    float* ws;                                                   // [ksplit][M][N]
    int* tickets;                                                // dense use with ksplit > 1: in-kernel reduction (common.h: splitk_last_arriver), one ticket per output tile; null = the caller reduces
    // BatchNorm statistics of the stored output from the epilogue (pconv_r_kernel, forward): per row tile and channel the sum and the sum of
    // squares (double) -> stat[(which * N + channel) * stat_nblk + tile]; null = none.  norm.hip's bn_stats_final_kernel adds the tiles in order.
    double* stat; int stat_nblk;
    // round 6, data gradient on the patch-resident kernel: the tile sums are those of the BatchNorm BACKWARD this gradient feeds -- sum of dy and of
    // dy * xhat, xhat = (bsx - bsmean) * bsinv, bsx = the BatchNorm's input [pixels][N] (the layout of the output) -- instead of the forward sums
    const float* bsx; const float* bsmean; const float* bsinv;
};

// XCD-aware workgroup -> tile mapping (same rule as gemm.hip's tile_of_block: XCD x owns a contiguous eighth of the tile sequence, n fastest)
// m / d for 0 <= m < 2^31 with mg = ceil(2^32 / d) (0 for d = 1): the estimate umulhi(m, mg) is the quotient or one more; one correction makes it exact
__device__ __forceinline__ int fast_div(int m, int d, unsigned mg) {
    if (mg == 0u) return m;
    int q = (int)__umulhi((unsigned)m, mg);
    if (q * d > m) --q;
    return q;
}
__device__ __forceinline__ void ptile_of_block(int& bx, int& by) {
    const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
    const int lin = blockIdx.y * nbx + blockIdx.x;
    const int xcd = lin & 7, per = total >> 3, rem = total & 7;
    const int seq = xcd * per + (xcd < rem ? xcd : rem) + (lin >> 3);
    bx = seq / nby; by = seq - bx * nby;
}

// Implicit GEMM over planes, blockIdx.z = parity class (PClass):
//   data gradient (p.fwd = 0):  dx[m][n] (+)= sum_{tap, co} dy[src(m, tap)][co] * wt[n][tap][co]      (stride 1 or 2)
//   forward       (p.fwd = 1):  y[m][n]   =   sum_{tap, ci} x[pix(m) * stride - pad + tap][ci] * w[n][tap][ci]   (one class: every tap)
// BM x BN output tile, 4 waves as WM x WN, wave tile (32 MI) x (32 NI), k tile = 32 channels of one filter tap, a RING of S LDS buffers and
// one barrier per k tile: tile kt + S - 1 is requested right after the barrier that retires tile kt - 1's buffer, so S - 1 tiles of DMA are in
// flight under a tile's MFMAs and the wait in front of a tile is a COUNTED `s_waitcnt vmcnt((S - 2) * loads per stage)` (every wave issues the
// same number of DMA instructions per stage), not a drain.  S = 2 is the round-3 first version (one tile ahead; two 64 KB workgroups per
// CU); the L2 -> LDS round trip under load (>= 1 us) is 2-3x a tile's 0.3 us of MFMAs, which is what S = 3 / 4 (one workgroup per CU) cover.
// NP = 3 (round 4, the default backward): three pieces per operand, SIX MFMAs per product, smallest products first
//   a2 b0 + a0 b2 + a1 b1 + a1 b0 + a0 b1 + a0 b0  -- every term down to 2^-24 of the product: the arithmetic class of the fp32 MFMA chain
//   (the reference's, train_hierarchy.py:264 loss.backward() in fp32), at 6 x 8 instead of 8 x 16 matrix-pipe passes per 32x32x16 block.
// NP = 2: hi + lo planes, three MFMAs per product (16-bit operand mantissa: the round-3 default, now the labelled secondary mode).  NP = 1: the
// operands ARE bf16 tensors (bf16-storage mode of BASELINE config 5, `--bf16`): one plane, one MFMA.
// OUT = 0: fp32 output (beta-accumulate); OUT = 1: bf16 output (optional ReLU, beta-accumulate).
template <int N_> __device__ __forceinline__ void wait_vm_lds_barrier() {
    // this wave's DMA older than the newest N_ instructions has landed, its LDS reads are complete; then the workgroup barrier.  The "memory"
    // clobber keeps the compiler's LDS accesses on their side of it (a bare __builtin_amdgcn_s_barrier() does not).
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N_) : "memory");
}

template <int BM, int BN, int WM, int WN, int NP, int OUT, int S>
__global__ __launch_bounds__(64 * WM * WN) void pconv_kernel(PConvP p) {
    constexpr int NWV = WM * WN;                                 // 4 waves, or 8 (two per SIMD, twice the tile: ha2g_conv_planes_waves)
    static_assert(NWV == 4 || NWV == 8, "four or eight waves");
    static_assert(S >= 2 && S <= 4, "ring depth");
    constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
    constexpr int RA = BM / 16, RB = BN / 16;                    // 16-row DMA pieces (1 KiB per wave instruction) per plane
    static_assert(RA % NWV == 0 && (RB % NWV == 0 || NWV % RB == 0), "every wave stages whole row blocks");
    constexpr int NA = RA / NWV, NB = (RB + NWV - 1) / NWV;      // row blocks per wave, per plane (RB < NWV: waves rb and rb + RB stage the same
                                                                 // block -- identical bytes to the same LDS address, every wave issues NB requests)
    constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;          // bytes
    constexpr int BUF = NP * (PLANE_A + PLANE_B);
    constexpr int LPS = NP * (NA + NB);                          // DMA instructions per wave and stage
    static_assert((S - 2) * LPS <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];      // S * BUF bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    ptile_of_block(bx, by);
    const PClass& pc = p.cls[blockIdx.z];
    const int m0 = bx * BM, n0 = by * BN;
    if (m0 >= pc.M) return;                                      // classes differ in size when H or W is odd
    const int nkc = p.GC >> 5;                                   // k tiles per filter tap
    const int nk = pc.ntaps * nkc;

    // ---- staging state: this lane's rows (fixed for the whole k loop) ----
    const int srow = lane >> 2;                                  // row inside a 16-row piece
    long a_base[NA]; unsigned a_mask[NA]; int a_lc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + NWV * i) * 16 + srow;
        a_lc[i] = (lane & 3) ^ ((row >> 2) & 3);                  // logical 16-byte piece this lane fetches (its LDS slot is lane & 3)
        const int m = m0 + row;
        a_mask[i] = 0u; a_base[i] = 0;
        if (m < pc.M) {
            const int oxc = m % pc.OWc; const int t = m / pc.OWc; const int oyc = t % pc.OHc; const int img = t / pc.OHc;
            if (p.fwd) {
                const int sy0 = oyc * p.stride - p.pad, sx0 = oxc * p.stride - p.pad;          // source pixel of tap (0, 0)
                a_base[i] = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC + a_lc[i] * 8;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int kh = pc.tap[ti] / p.KW, kw = pc.tap[ti] - kh * p.KW;
                    if (sy0 + kh >= 0 && sy0 + kh < p.GH && sx0 + kw >= 0 && sx0 + kw < p.GW) a_mask[i] |= 1u << ti;
                }
            } else {
                const int u = oyc * p.stride + pc.py + p.pad, v = oxc * p.stride + pc.px + p.pad;
                const int sy0 = (u - pc.kh0) / p.stride, sx0 = (v - pc.kw0) / p.stride;      // source pixel of the class's first tap (exact divisions)
                a_base[i] = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC + a_lc[i] * 8;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int kh = pc.tap[ti] / p.KW, kw = pc.tap[ti] - kh * p.KW;
                    const int sy = (u - kh) / p.stride, sx = (v - kw) / p.stride;
                    if (u - kh >= 0 && v - kw >= 0 && sy < p.GH && sx < p.GW) a_mask[i] |= 1u << ti;
                }
            }
        }
    }
    long b_off[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = ((wave + NWV * i) % RB) * 16 + srow;
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        b_off[i] = n < p.N ? (long)n * p.K + lc * 8 : -1;
    }
    const unsigned short* zero = g_zero_page;

    auto stage = [&](int kt, int buf) {
        const int ti = kt / nkc, c0 = (kt - ti * nkc) << 5;
        const long koff = c0 + (long)pc.doff[ti] * p.GC;
        const unsigned bit = 1u << ti;
        unsigned char* dst = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool on = (a_mask[i] & bit) != 0u;
            const long o = a_base[i] + koff;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.a.p + q * p.a.ps + o : zero), (lds_ptr_t)(dst + q * PLANE_A + (wave + NWV * i) * 1024), 16, 0, 0);
        }
        const long kb = (long)pc.tap[ti] * p.GC + c0;             // k index of the weight planes: (tap, channel)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool on = b_off[i] >= 0;
            const int rb = (wave + NWV * i) % RB;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.b.p + q * p.b.ps + b_off[i] + kb : zero),
                                                 (lds_ptr_t)(dst + NP * PLANE_A + q * PLANE_B + rb * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int sw = (l31 >> 2) & 3;                               // swizzle of this lane's fragment rows (row = 32 * tile + l31)
    const int a_row_off = (wm * 32 * MI + l31) * 64, b_row_off = (wn * 32 * NI + l31) * 64;

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) stage(s, s);
    int cur = 0, nxt = S - 1;                                    // ring slots of tile kt and of tile kt + S - 1
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed in every wave (S - 2 newer stages may still be in flight; near the tail fewer were issued: drain) and every
        // wave has finished reading tile kt - 1, whose slot the request below overwrites
        if (kt + S - 2 < nk) wait_vm_lds_barrier<(S - 2) * LPS>();
        else wait_vm_lds_barrier<0>();
        if (kt + S - 1 < nk && !(p.dbg & 1)) stage(kt + S - 1, nxt);
        const unsigned char* ab = smem + cur * BUF + a_row_off;
        const unsigned char* bb = smem + cur * BUF + NP * PLANE_A + b_row_off;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int po = ((2 * kc + lhi) ^ sw) * 16;
            bf16x8_t af[NP][MI], bf[NP][NI];                     // piece q of the A / B fragments
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[q][i] = *reinterpret_cast<const bf16x8_t*>(ab + q * PLANE_A + i * 2048 + po);
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[q][j] = *reinterpret_cast<const bf16x8_t*>(bb + q * PLANE_B + j * 2048 + po);
            }
            if (p.dbg & 2) {                                     // ablation: keep the fragment reads alive, skip the matrix pipe
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(af[q][i]));
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(bf[q][j]));
                }
                continue;
            }
            if constexpr (NP == 3) {
                // six products, smallest first; each pass walks ALL accumulators so that consecutive MFMAs are independent
                constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA[t]][i], bf[QB[t]][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if constexpr (NP == 2) {                 // lo * hi, hi * lo, hi * hi: the order (per accumulator) of gemm_x3_kernel<.., 2>
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP - 1][i], bf[0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[NP - 1][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        cur = cur + 1 == S ? 0 : cur + 1;
        nxt = nxt + 1 == S ? 0 : nxt + 1;
    }

    // ---- epilogue: C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (32 * NI) + j * 32 + l31;
            if (OUT == 0 && col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const bool rok = row < pc.M;
                if (OUT == 0 && !rok) continue;
                long orow = row;                                             // stride 1 / forward: class-linear = output-linear
                if (!p.fwd && p.stride != 1 && rok) {
                    const int oxc = row % pc.OWc; const int t = row / pc.OWc; const int oyc = t % pc.OHc; const int img = t / pc.OHc;
                    orow = ((long)img * p.OH + oyc * p.stride + pc.py) * p.OW + oxc * p.stride + pc.px;
                }
                if (OUT == 0) {
                    float* dst = p.C + orow * p.ldc + col;
                    float v = 1.0f * acc[i][j][r] + 0.f;              // alpha = 1, no bias: the epilogue arithmetic of gemm_x3_kernel
                    if (p.relu) v = fmaxf(v, 0.f);                   // forward convolutions (conv -> ReLU -> BN ordering of the reference)
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                } else {
                    // bf16 output: two adjacent columns (adjacent lanes) per 4-byte store
                    unsigned short* dstb = reinterpret_cast<unsigned short*>(p.C) + orow * p.ldc + (col & ~1);
                    float v = acc[i][j][r];
                    const float vn = __shfl_down(v, 1, 64);
                    if (rok && !(l31 & 1) && col < p.N) {                   // N is even: col and col + 1 are both valid
                        float v0 = v, v1 = vn;
                        if (p.beta != 0.f) {
                            const unsigned old = *reinterpret_cast<const unsigned*>(dstb);
                            v0 += p.beta * __uint_as_float(old << 16); v1 += p.beta * __uint_as_float(old & 0xffff0000u);
                        }
                        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        unsigned h, l;
                        split2_bf16(v0, v1, h, l);
                        *reinterpret_cast<unsigned*>(dstb) = h;
                    }
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// PING-PONG form of the plane convolution (round 4).  Counter evidence first (profiles/r04_planes_ablate_np3_v1.txt, B = 512, C = 256, three
// pieces, 128 x 128 tile, one workgroup per CU): full 581 us, without DMA 419, without MFMA 388, neither 157 -- the three phases of a k tile
// (fragment reads + barrier 0.48 us, DMA 0.71 us, 48 MFMAs 0.81 us) add up: with ONE wave per SIMD and a barrier per k tile nothing overlaps.
// Here a workgroup is TWO groups of four waves (two waves per SIMD), each group owning a 128 x BN half of a 256 x BN output tile, and the
// groups run in ANTI-PHASE: while group g issues the 48 (24) MFMAs of its k tile out of REGISTERS, group 1 - g reads ALL fragments of its next
// k tile from LDS into registers (96 / 72 VGPRs) and then requests the tile after that by DMA.  One workgroup barrier per phase.  The matrix
// pipe of a SIMD is therefore always fed by one of its two waves while the other one waits for LDS / issues DMA (guide: "two waves per SIMD:
// what they share" -- the merged interval pairs matrix work with memory work).
//   phase 2k     group 0: LOAD(k)    = fragments of tile k -> registers; request A0(k+1) and B(k+1)        group 1: COMPUTE(k-1)
//   phase 2k+1   group 0: COMPUTE(k) = the MFMAs; then wait for its own DMA (it had the whole phase)       group 1: LOAD(k); request A1(k+1)
// Hazards: a DMA into stage (k+1) % 2 is issued one barrier after the last fragment read of that stage (tile k-1); a tile is read one barrier
// after its issuers' `s_waitcnt vmcnt(0)`.  B (the weights) is shared by both groups and requested by group 0 only -- group 1 reads B(k) one
// phase later than group 0, B(k+2) replaces it one phase after that.  LDS: A 2 groups x 2 stages x NP x 8 KB + B 2 stages x NP x BN x 64 B
// = 144 KB (BN = 128) / 120 KB (BN = 64) for three pieces.
template <int BN, int NP, int OUT>
__global__ __launch_bounds__(512) void pconv_pp_kernel(PConvP p) {
    constexpr int GM = 128;                                      // rows of one group's half tile
    constexpr int MI = 2, NI = BN / 64;                          // group = 2 x 2 waves, wave tile 64 x (BN / 2)
    constexpr int PLANE_A = GM * 64, PLANE_B = BN * 64;          // bytes
    constexpr int A_STAGE = NP * PLANE_A, B_STAGE = NP * PLANE_B;
    constexpr int B_BASE = 4 * A_STAGE;                          // A: [group][stage], then B: [stage]
    constexpr int NBW = (BN / 16) / 4;                           // B row blocks per wave of group 0 (2 / 1)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    int bx, by;
    ptile_of_block(bx, by);
    const PClass& pc = p.cls[blockIdx.z];
    const int m0 = bx * 256 + grp * GM, n0 = by * BN;
    if (bx * 256 >= pc.M) return;                                // whole workgroup out of range (classes differ in size)
    const int nkc = p.GC >> 5;
    const int nk = pc.ntaps * nkc;

    // ---- staging state: this lane's two A rows (of its group's half tile) and, in group 0, its B rows ----
    const int srow = lane >> 2;
    long a_base[2]; unsigned a_mask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (w4 + 4 * i) * 16 + srow;
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        a_mask[i] = 0u; a_base[i] = 0;
        if (m < pc.M) {
            const int oxc = m % pc.OWc; const int t = m / pc.OWc; const int oyc = t % pc.OHc; const int img = t / pc.OHc;
            if (p.fwd) {
                const int sy0 = oyc * p.stride - p.pad, sx0 = oxc * p.stride - p.pad;
                a_base[i] = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC + lc * 8;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int kh = pc.tap[ti] / p.KW, kw = pc.tap[ti] - kh * p.KW;
                    if (sy0 + kh >= 0 && sy0 + kh < p.GH && sx0 + kw >= 0 && sx0 + kw < p.GW) a_mask[i] |= 1u << ti;
                }
            } else {
                const int u = oyc * p.stride + pc.py + p.pad, v = oxc * p.stride + pc.px + p.pad;
                const int sy0 = (u - pc.kh0) / p.stride, sx0 = (v - pc.kw0) / p.stride;
                a_base[i] = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC + lc * 8;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int kh = pc.tap[ti] / p.KW, kw = pc.tap[ti] - kh * p.KW;
                    const int sy = (u - kh) / p.stride, sx = (v - kw) / p.stride;
                    if (u - kh >= 0 && v - kw >= 0 && sy < p.GH && sx < p.GW) a_mask[i] |= 1u << ti;
                }
            }
        }
    }
    long b_off[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int row = (w4 + 4 * i) * 16 + srow;
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        b_off[i] = n < p.N ? (long)n * p.K + lc * 8 : -1;
    }
    const unsigned short* zero = g_zero_page;
    unsigned char* const a_lds = smem + grp * 2 * A_STAGE;      // this group's two A stages

    auto stage_a = [&](int kt) {
        const int ti = kt / nkc, c0 = (kt - ti * nkc) << 5;
        const long koff = c0 + (long)pc.doff[ti] * p.GC;
        const unsigned bit = 1u << ti;
        unsigned char* dst = a_lds + (kt & 1) * A_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool on = (a_mask[i] & bit) != 0u;
            const long o = a_base[i] + koff;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.a.p + q * p.a.ps + o : zero), (lds_ptr_t)(dst + q * PLANE_A + (w4 + 4 * i) * 1024), 16, 0, 0);
        }
    };
    auto stage_b = [&](int kt) {
        const int ti = kt / nkc, c0 = (kt - ti * nkc) << 5;
        const long kb = (long)pc.tap[ti] * p.GC + c0;
        unsigned char* dst = smem + B_BASE + (kt & 1) * B_STAGE;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const bool on = b_off[i] >= 0;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.b.p + q * p.b.ps + b_off[i] + kb : zero), (lds_ptr_t)(dst + q * PLANE_B + (w4 + 4 * i) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    const int a_row_off = (wm * 64 + l31) * 64, b_row_off = (wn * 32 * NI + l31) * 64;
    bf16x8_t af[2][NP][MI], bf[2][NP][NI];                       // [k chunk][piece][tile]: the fragments of ONE whole k tile live in registers

    auto load_frags = [&](int kt) {
        const unsigned char* ab = a_lds + (kt & 1) * A_STAGE + a_row_off;
        const unsigned char* bb = smem + B_BASE + (kt & 1) * B_STAGE + b_row_off;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int po = ((2 * kc + lhi) ^ sw) * 16;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[kc][q][i] = *reinterpret_cast<const bf16x8_t*>(ab + q * PLANE_A + i * 2048 + po);
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[kc][q][j] = *reinterpret_cast<const bf16x8_t*>(bb + q * PLANE_B + j * 2048 + po);
            }
        }
    };
    auto compute = [&]() {
        if (p.dbg & 2) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(af[kc][q][i]));
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(bf[kc][q][j]));
                }
            return;
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            if constexpr (NP == 3) {
                constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first (pconv_kernel)
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc][QA[t]][i], bf[kc][QB[t]][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if constexpr (NP == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc][NP - 1][i], bf[kc][0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc][0][i], bf[kc][NP - 1][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kc][0][i], bf[kc][0][j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    };
    // end of a LOAD phase: this wave's fragment reads are complete (its DMA stays in flight); end of a COMPUTE phase: its DMA has landed.
    // (A two-tiles-ahead variant -- request tile k + 2 into the stage tile k was just read from, counted `vmcnt` -- was measured no faster and
    // is WRONG as written: the stage is read by the group's OTHER waves too, and only a workgroup barrier orders their reads before the DMA.)
    auto end_load = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    auto end_compute = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };

    // ---- prologue: tile 0 of everything ----
    stage_a(0);
    if (grp == 0) stage_b(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool dma = !(p.dbg & 1);
    if (grp == 0) {
        for (int k = 0; k < nk; ++k) {
            load_frags(k);                                       // phase 2k
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nk && dma) { stage_a(k + 1); stage_b(k + 1); }
            end_load();
            compute();                                           // phase 2k + 1
            __builtin_amdgcn_sched_barrier(0);
            end_compute();
        }
        asm volatile("s_barrier" ::: "memory");                  // phase 2 nk: group 1's last COMPUTE
    } else {
        asm volatile("s_barrier" ::: "memory");                  // phase 0: group 0's first LOAD
        for (int k = 0; k < nk; ++k) {
            load_frags(k);                                       // phase 2k + 1
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nk && dma) stage_a(k + 1);
            end_load();
            compute();                                           // phase 2k + 2
            __builtin_amdgcn_sched_barrier(0);
            end_compute();
        }
    }

    // ---- epilogue (as pconv_kernel) ----
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (32 * NI) + j * 32 + l31;
            if (OUT == 0 && col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const bool rok = row < pc.M;
                if (OUT == 0 && !rok) continue;
                long orow = row;
                if (!p.fwd && p.stride != 1 && rok) {
                    const int oxc = row % pc.OWc; const int t = row / pc.OWc; const int oyc = t % pc.OHc; const int img = t / pc.OHc;
                    orow = ((long)img * p.OH + oyc * p.stride + pc.py) * p.OW + oxc * p.stride + pc.px;
                }
                if (OUT == 0) {
                    float* dst = p.C + orow * p.ldc + col;
                    float v = acc[i][j][r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                } else {
                    unsigned short* dstb = reinterpret_cast<unsigned short*>(p.C) + orow * p.ldc + (col & ~1);
                    float v = acc[i][j][r];
                    const float vn = __shfl_down(v, 1, 64);
                    if (rok && !(l31 & 1) && col < p.N) {
                        float v0 = v, v1 = vn;
                        if (p.beta != 0.f) {
                            const unsigned old = *reinterpret_cast<const unsigned*>(dstb);
                            v0 += p.beta * __uint_as_float(old << 16); v1 += p.beta * __uint_as_float(old & 0xffff0000u);
                        }
                        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        unsigned h, l;
                        split2_bf16(v0, v1, h, l);
                        *reinterpret_cast<unsigned*>(dstb) = h;
                    }
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// QUANTISATION-FREE form of the ping-pong kernel (round 4): v_mfma_f32_16x16x32_bf16 tiles, so that a group's half tile is ANY multiple of 16
// rows.  At the headline batch the trunk's convolutions are small against the chip: layer 3 is 73 728 rows = 288 workgroup tiles of 256 rows on
// 256 CUs -- two rounds, the second one 12 % full (56 % of the matrix pipe over the launch); layer 4 is 144 tiles (one round on 56 % of the CUs);
// layer 2 is 1 120 tiles (4.4 -> 5 rounds).  With 2 x 16 MT rows per workgroup the tile count can be made a multiple of the CU count:
//   layer 3: MT = 9 (288 rows) x 128 columns -> 256 workgroups, ONE full round;   layer 4: MT = 9 x 64 columns (four column tiles) -> 256;
//   layer 2: MT = 7 (224 rows) x 64 columns -> 1 280 workgroups = five full rounds.
// Same anti-phase schedule as pconv_pp_kernel (LOAD: every fragment of a k tile -> registers, then the next tile's DMA; COMPUTE: MFMAs out of
// registers); the four waves of a group split the COLUMNS (each wave: all 16 MT rows x BN / 4 columns, MT x NI accumulators of four registers),
// so the A fragments are read by all four waves (LDS bandwidth is not the bound here) and one ds_read_b128 per 16-row tile and piece is a whole
// K = 32 fragment (lane l: row l & 15, 16-byte piece l >> 4).  LDS rows are 64 bytes; the piece a lane group touches is XOR-ed with
// F[(row >> 2) & 3], F = {0, 3, 2, 1}: the four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...) then cover all 16 slots of the
// 256-byte bank row (the 32 x 32 kernels' swizzle (row >> 2) & 3 is 2-way conflicted for this access pattern).
template <int MT, int BN, int NP, int OUT>
__global__ __launch_bounds__(512) void pconv_q_kernel(PConvP p) {
    constexpr int GM = 16 * MT;                                  // rows of one group's half tile
    constexpr int NI = BN / 64;                                  // 16-column tiles per wave (wave = BN / 4 columns)
    constexpr int PLANE_A = GM * 64, PLANE_B = BN * 64;          // bytes
    constexpr int A_STAGE = NP * PLANE_A, B_STAGE = NP * PLANE_B;
    constexpr int B_BASE = 4 * A_STAGE;                          // A: [group][stage], then B: [stage]
    constexpr int NTW = (MT + 3) / 4;                            // A row tiles a wave stages (tiles w4, w4 + 4, ...)
    constexpr int NBW = (BN / 16) / 4;                           // B row blocks per wave of group 0
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    if (p.dbg & 32) return;                                      // ablation bit 5: launch cost only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    int bx, by;
    ptile_of_block(bx, by);
    const bool ksp = p.ksplit > 1;
    const PClass& pc = p.cls[ksp ? 0 : blockIdx.z];
    const int m0 = bx * (2 * GM) + grp * GM, n0 = by * BN;
    if (bx * (2 * GM) >= pc.M) return;
    const int nkc = p.GC >> 5;
    const int nk_all = pc.ntaps * nkc;
    const int kt0 = ksp ? (int)blockIdx.z * p.kt_per : 0;        // this workgroup's k tiles: [kt0, kt0 + nk)
    const int nk = ksp ? (kt0 + p.kt_per < nk_all ? p.kt_per : nk_all - kt0) : nk_all;

    // ---- staging state: this lane's row in each of the wave's A row tiles and, in group 0, B row blocks ----
    const int srow = lane >> 2;
    constexpr int FSW[4] = {0, 3, 2, 1};
    const int lc8 = ((lane & 3) ^ FSW[(srow >> 2) & 3]) * 8;      // logical 16-byte piece (in elements) this lane fetches into slot lane & 3
    unsigned a_off[NTW]; unsigned a_mask[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tile = w4 + 4 * i;
        const int m = m0 + tile * 16 + srow;
        a_mask[i] = 0u; a_off[i] = 0u;
        if (tile < MT && m < pc.M) {
            // (the first version divided by OWc / OHc / KW / stride per row and tap here: ~2 000 instructions per lane, 36 of the 162 us of a
            //  64-channel launch went into launch + prologue -- tools/planes_ablate.py, columns "no k loop")
            const int t = fast_div(m, pc.OWc, pc.mg_ow), oxc = m - t * pc.OWc;
            const int img = fast_div(t, pc.OHc, pc.mg_oh), oyc = t - img * pc.OHc;
            long base;
            if (p.fwd) {
                const int sy0 = oyc * p.stride - p.pad, sx0 = oxc * p.stride - p.pad;
                base = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int kh = pc.khs[ti], kw = pc.kws[ti];
                    if (sy0 + kh >= 0 && sy0 + kh < p.GH && sx0 + kw >= 0 && sx0 + kw < p.GW) a_mask[i] |= 1u << ti;
                }
            } else {                                             // data gradient (stride 1, or the single live parity class of a stride-2 1x1 convolution)
                const int sh = p.stride >> 1;                    // stride 1 / 2: u - kh is a multiple of the stride for every tap of the class
                const int u = oyc * p.stride + pc.py + p.pad, v = oxc * p.stride + pc.px + p.pad;
                const int sy0 = (u - pc.kh0) >> sh, sx0 = (v - pc.kw0) >> sh;
                base = (((long)img * p.GH + sy0) * p.GW + sx0) * p.GC;
                for (int ti = 0; ti < pc.ntaps; ++ti) {
                    const int dy_ = u - pc.khs[ti], dx_ = v - pc.kws[ti];
                    if (dy_ >= 0 && dx_ >= 0 && (dy_ >> sh) < p.GH && (dx_ >> sh) < p.GW) a_mask[i] |= 1u << ti;
                }
            }
            // masked-in taps address inside the plane; the offset of tap (0, 0) may be "negative" at the border: kept modulo 2^32 and added to the
            // (non-negative) tap offset in 32-bit arithmetic -- a plane is < 2^31 elements (host check)
            a_off[i] = (unsigned)(base + lc8);
        }
    }
    long b_off[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int n = n0 + (w4 + 4 * i) * 16 + srow;
        b_off[i] = n < p.N ? (long)n * p.K + lc8 : -1;
    }
    const unsigned short* zero = g_zero_page;
    unsigned char* const a_lds = smem + grp * 2 * A_STAGE;

    // k tile -> (tap, first channel).  Split products walk the k axis CHANNEL-major -- the nine taps of one 32-channel slice back to back: a tap
    // re-reads the pixels of the previous one shifted by a pixel or a row, so the re-read now follows within one k tile and hits the XCD's L2.
    // Tap-major (all channels of tap 0, then tap 1, ...) put 4-8 k tiles of every workgroup of the XCD between them: at C = 128 the PMC showed
    // 470 MB fetched past L2 per launch for 57 MB of operands (profiles/r04_pmc_step_bytes_b128_before_korder.txt).  The one-piece kernel of the
    // bf16-storage mode keeps the tap-major order (its fixtures are pinned to that summation order).
    // The k tiles are staged strictly in order (0, 1, 2, ... of this workgroup's range), so (tap, channel slice) is a CURSOR advanced once per tile -- the
    // first version recomputed it by an integer division per staging call: ~35 SALU + 15 VALU each, in the phase that bounds the kernel -- and the
    // tap's table entries (kernel arguments: scalar loads) are fetched one tile ahead.
    const bool cmaj = NP >= 2 && p.kmaj;
    int s_ti, s_cc;
    if (cmaj) { s_cc = kt0 / pc.ntaps; s_ti = kt0 - s_cc * pc.ntaps; } else { s_ti = kt0 / nkc; s_cc = kt0 - s_ti * nkc; }
    int s_doff = pc.doff[s_ti], s_tap = pc.tap[s_ti];
    auto next_k = [&]() {
        if (cmaj) { if (++s_ti == pc.ntaps) { s_ti = 0; ++s_cc; } }
        else { if (++s_cc == nkc) { s_cc = 0; ++s_ti; } }
        s_doff = pc.doff[s_ti]; s_tap = pc.tap[s_ti];           // past the last tile: an unused entry of the same struct
    };
    // Buffer-addressed DMA (p.buf, default): one buffer resource per piece plane (SGPRs), the lane's address is a 32-bit byte offset inside the plane,
    // and padding is an offset past the end -- the load returns zeros (tools/probe/buffer_lds_probe.hip) -- instead of a per-lane 64-bit select
    // between the plane and the zero page for every piece: 3 VALU per row tile and k tile instead of ~8 per DMA instruction.
    // (the resources are wave-uniform SGPR quads built at the use site: loop-invariant, the compiler keeps them in SGPRs)
    constexpr unsigned OOB = 0xfffffff0u;
    auto stage_a = [&](int kl) {                                 // kl = k tile index local to this workgroup's range
        const unsigned koff = (unsigned)((s_cc << 5) + (long)s_doff * p.GC);
        const unsigned bit = 1u << s_ti;
        unsigned char* dst = a_lds + (kl & 1) * A_STAGE;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (w4 + 4 * i < MT) {                               // wave-uniform
                const bool on = (a_mask[i] & bit) != 0u;
                const unsigned o = a_off[i] + koff;
                if (p.buf) {
                    const unsigned vo = on ? o << 1 : OOB;
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(p.a.p + q * p.a.ps), 0, p.a_bytes, 0x00020000),
                                                                 (lds_ptr_t)(dst + q * PLANE_A + (w4 + 4 * i) * 1024), 16, (int)vo, 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.a.p + q * p.a.ps + o : zero), (lds_ptr_t)(dst + q * PLANE_A + (w4 + 4 * i) * 1024), 16, 0, 0);
                }
            }
        }
    };
    auto stage_b = [&](int kl) {
        const long kb = (long)s_tap * p.GC + (s_cc << 5);
        unsigned char* dst = smem + B_BASE + (kl & 1) * B_STAGE;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const bool on = b_off[i] >= 0;
            if (p.buf) {
                const unsigned vo = on ? (unsigned)(b_off[i] + kb) << 1 : OOB;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(p.b.p + q * p.b.ps), 0, p.b_bytes, 0x00020000),
                                                             (lds_ptr_t)(dst + q * PLANE_B + (w4 + 4 * i) * 1024), 16, (int)vo, 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.b.p + q * p.b.ps + b_off[i] + kb : zero), (lds_ptr_t)(dst + q * PLANE_B + (w4 + 4 * i) * 1024), 16, 0, 0);
            }
        }
    };

    f32x4_t acc[MT][NI];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, kp = lane >> 4;
    const int po = (kp ^ FSW[(l15 >> 2) & 3]) * 16;              // this lane's (swizzled) 16-byte piece inside a 64-byte row
    const int a_lane = l15 * 64 + po, b_lane = (w4 * (BN / 4) + l15) * 64 + po;
    bf16x8_t af[NP][MT], bf[NP][NI];                             // every fragment of ONE k tile

    auto load_frags = [&](int kt) {
        const unsigned char* ab = a_lds + (kt & 1) * A_STAGE + a_lane;
        const unsigned char* bb = smem + B_BASE + (kt & 1) * B_STAGE + b_lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[q][j] = *reinterpret_cast<const bf16x8_t*>(bb + q * PLANE_B + j * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[q][i] = *reinterpret_cast<const bf16x8_t*>(ab + q * PLANE_A + i * 1024);
        }
    };
    auto compute = [&]() {
        if (p.dbg & 2) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(af[q][i]));
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(bf[q][j]));
            }
            return;
        }
        if constexpr (NP == 3) {
            constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[QB[t]][j], af[QA[t]][i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (NP == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[NP - 1][i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[NP - 1][j], af[0][i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][j], af[0][i], acc[i][j], 0, 0, 0);
                }
        }
    };
    auto end_load = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    auto end_compute = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };

    stage_a(0);
    if (grp == 0) stage_b(0);
    next_k();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool dma = !(p.dbg & 1);
    const int nkr = (p.dbg & 4) ? 0 : nk;                        // ablation bit 2: skip the k loop (launch + prologue + epilogue remain)
    if (grp == 0) {
        for (int k = 0; k < nkr; ++k) {
            if (p.dbg & 16) {                                    // A/B: the first form (fragment reads, then the next tile's DMA)
                load_frags(k);                                   // phase 2k
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < nk && dma) { stage_a(k + 1); stage_b(k + 1); next_k(); }
            } else {                                             // the next tile's DMA FIRST: its stage was retired a phase ago, and every cycle of flight counts
                if (k + 1 < nk && dma) { stage_a(k + 1); stage_b(k + 1); next_k(); }
                __builtin_amdgcn_sched_barrier(0);
                load_frags(k);
            }
            __builtin_amdgcn_sched_barrier(0);
            end_load();
            compute();                                           // phase 2k + 1
            __builtin_amdgcn_sched_barrier(0);
            end_compute();
        }
        asm volatile("s_barrier" ::: "memory");
    } else {
        asm volatile("s_barrier" ::: "memory");
        for (int k = 0; k < nkr; ++k) {
            if (p.dbg & 16) {
                load_frags(k);                                   // phase 2k + 1
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < nk && dma) { stage_a(k + 1); next_k(); }
            } else {
                if (k + 1 < nk && dma) { stage_a(k + 1); next_k(); }
                __builtin_amdgcn_sched_barrier(0);
                load_frags(k);
            }
            __builtin_amdgcn_sched_barrier(0);
            end_load();
            compute();                                           // phase 2k + 2
            __builtin_amdgcn_sched_barrier(0);
            end_compute();
        }
    }

    // ---- epilogue.  The MFMAs take the WEIGHT fragment as their first operand: D = W X^T, so a lane's four accumulator registers are four consecutive
    //      output CHANNELS (4 (lane >> 4) + r) of one pixel (lane & 15) -- one 16-byte store per 16x16 tile and lane instead of four 4-byte stores
    //      to four different rows (the output stores were 27 of the 162 us of a 64-channel launch: tools/planes_ablate.py, "no stores"). ----
    static_assert(OUT == 0, "fp32 output only");
    if (p.dbg & 8) return;                                       // ablation bit 3: no output stores
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = m0 + i * 16 + l15;
        if (row >= pc.M) continue;
        long orow = row;                                         // forward / stride-1 data gradient: output pixel = GEMM row
        if (!p.fwd && p.stride != 1) {                           // the live parity class of a stride-2 1x1 data gradient: its pixels inside the full grid
            const int t = fast_div(row, pc.OWc, pc.mg_ow), oxc = row - t * pc.OWc;
            const int img = fast_div(t, pc.OHc, pc.mg_oh), oyc = t - img * pc.OHc;
            orow = ((long)img * p.OH + oyc * p.stride + pc.py) * p.OW + oxc * p.stride + pc.px;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col0 = n0 + w4 * (BN / 4) + j * 16 + 4 * kp;
            if (col0 >= p.N) continue;
            f32x4_t v = acc[i][j];
            if (ksp) {                                           // raw partial: the reduce applies the epilogue
                float* d = p.ws + ((long)blockIdx.z * pc.M + orow) * p.N + col0;
                if (p.vec) *reinterpret_cast<f32x4_t*>(d) = v;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (col0 + r < p.N) d[r] = v[r];
                }
                continue;
            }
            float* dst = p.C + orow * p.ldc + col0;
            if (p.vec) {
                if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + col0);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }   // convolution callers: relu(conv)
                if (p.beta != 0.f) v += p.beta * *reinterpret_cast<const f32x4_t*>(dst);
#pragma unroll
                for (int r = 0; r < 4; ++r) {                    // dense callers: act(A B^T + bias + beta C), the convention of ha2g_gemm_f32
                    if (p.act == 1) v[r] = fmaxf(v[r], 0.f);
                    else if (p.act == 2) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
                }
                *reinterpret_cast<f32x4_t*>(dst) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (col0 + r >= p.N) continue;
                    float x = v[r];
                    if (p.bias) x += p.bias[col0 + r];
                    if (p.relu) x = fmaxf(x, 0.f);
                    if (p.beta != 0.f) x += p.beta * dst[r];
                    if (p.act == 1) x = fmaxf(x, 0.f);
                    else if (p.act == 2) x = x > 0.f ? x : 0.01f * x;
                    dst[r] = x;
                }
            }
        }
    }
    // in-kernel split-K reduction (dense use only: output row = GEMM row): the last of this tile's k slices to arrive adds the ksplit raw slabs in slice
    // order, in double (the arithmetic of splitk_reduce_kernel), applies bias / beta / activation and stores the 2 GM x BN tile
    if (ksp && p.tickets != nullptr) {
        int* sh = reinterpret_cast<int*>(smem);
        if (splitk_last_arriver(p.tickets + (long)by * gridDim.x + bx, (int)gridDim.z, sh)) {
            const long MN = (long)pc.M * p.N;
            const int r0 = bx * (2 * GM);
            for (int idx = tid; idx < 2 * GM * (BN / 4); idx += 512) {
                const int row = r0 + idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
                if (row >= pc.M || col >= p.N) continue;
                const int nv = p.N - col < 4 ? p.N - col : 4;
                const float* src = p.ws + (long)row * p.N + col;
                double sd[4] = {0.0, 0.0, 0.0, 0.0};
                if (p.vec) splitk_ordered_sum4(src, MN, (int)gridDim.z, sd);
                else for (int c = 0; c < nv; ++c) sd[c] = splitk_ordered_sum(src + c, MN, (int)gridDim.z);
                float* dst = p.C + (long)row * p.ldc + col;
                f32x4_t v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = 0.f;
                    if (c < nv) {
                        x = (float)sd[c] + (p.bias ? p.bias[col + c] : 0.f);
                        if (p.beta != 0.f) x += p.beta * dst[c];
                        if (p.act == 1) x = fmaxf(x, 0.f);
                        else if (p.act == 2) x = x > 0.f ? x : 0.01f * x;
                    }
                    v[c] = x;
                }
                if (p.vec) *reinterpret_cast<f32x4_t*>(dst) = v;
                else for (int c = 0; c < nv; ++c) dst[c] = v[c];
            }
        }
    }
    // BatchNorm statistics of the stored output (see pconv_r_kernel's epilogue; forward convolutions only: relu is the only epilogue term): block
    // 2 bx + group of stat_nblk = 2 gridDim.x
    if (p.stat != nullptr) {
        constexpr int NV = NI * 8;
        double sv[NV];
        bool live[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) live[i] = m0 + i * 16 + l15 < pc.M;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // SHIFTED lane sums: the fp32 part runs on v - K, K = the lane's first pixel of this channel, so its rounding is relative to
                // the channel's SPREAD, not to its mean -- raw sums of squares lose (1 + mean^2 / var) x 5e-7 of the variance when the finalize pass
                // forms s2 / n - mean^2 (a channel with |mean| = 30 sigma: 5e-4).  Back to raw sums in double: S1 = s1 + n K, S2 = s2 + K (2 s1 + n K).
                float kf = acc[0][j][u];
                if (p.relu) kf = fmaxf(kf, 0.f);
                kf = live[0] ? kf : 0.f;
                float s1 = 0.f, s2 = 0.f;
                int nl = 0;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float v = acc[i][j][u];
                    if (p.relu) v = fmaxf(v, 0.f);
                    v = live[i] ? v - kf : 0.f;
                    nl += live[i] ? 1 : 0;
                    s1 += v; s2 = fmaf(v, v, s2);
                }
                const double kd = (double)kf, nk = (double)nl * kd;
                sv[(j * 4 + u) * 2] = (double)s1 + nk; sv[(j * 4 + u) * 2 + 1] = (double)s2 + kd * (2.0 * (double)s1 + nk);
            }
        row16_reduce_scatter<NV>(sv, lane);
        const int idx = l15 & (NV - 1);
        const int ch = n0 + w4 * (BN / 4) + (idx >> 3) * 16 + 4 * kp + ((idx >> 1) & 3);
        if ((NV == 16 || l15 < 8) && ch < p.N) p.stat[((long)(idx & 1) * p.N + ch) * p.stat_nblk + 2 * bx + grp] = sv[0];
    }
}



// ---------------------------------------------------------------------------------------------------------------------------------------
// PATCH-RESIDENT plane kernel (round 5): pconv_r_kernel -- 3x3 / stride 1 / pad 1, forward or data gradient, three pieces.
// The q kernel re-stages a shifted A tile for every k tile (one tap of one 32-channel slice): every pixel travels L2 -> LDS NINE times per slice, at
// 100-220 cycles of issue per LDS-DMA instruction, in a LOAD phase that two workgroup barriers per k tile keep in lockstep with the other group's
// COMPUTE phase.  Here
//   * a workgroup = FOUR waves = one tile of GM = 16 MT consecutive pixels of ONE image x BN output channels (MT = 7: 112 px = 1/20 of a 64x35 image;
//     MT = 9: 144 px = 1/4 of 32x18, all of 16x9); its A operand is the PATCH of the tile -- the tile's image rows plus a one-pixel halo, one 32-channel
//     slice, three pieces, 52-61 KB -- staged by DMA ONCE per slice; the nine taps read their fragments from it at shifted addresses (as
//     conv3x3_c32pp_kernel does): 4-7x fewer bytes and DMA instructions through the LDS-DMA path;
//   * the weight fragments never touch LDS: a lane's fragment (column n, k chunk kp) is 16 contiguous bytes of row n of the [N][K] weight planes --
//     one buffer_load_dwordx4 per (piece, column tile) with a per-lane constant offset and the k tile's (tap, slice) offset in an SGPR, issued a whole
//     k tile ahead into a second register set;
//   * so a k tile needs NO barrier: reads -> MFMAs per wave, and the two or three workgroups resident per CU (52 KB of LDS, 160-256 VGPRs) fill each
//     other's read phases, prologues and epilogues.  Barriers remain where the single-buffered patch is replaced (once per 32-channel slice): its DMA
//     is issued behind the last tap's fragment reads and flies under that tap's MFMAs;
//   * patch rows are PW = W + 8 pixels wide (1 left pad, 7 right): a 16-pixel fragment that wraps around an image row continues 8 patch pixels
//     further on, which the bank pattern cannot tell from a consecutive run -- with the slot swizzle  s ^ 2 ((pixel >> 2) & 1)  and the fragment
//     columns permuted (columns {0-3, 12-15} = pixels 0-7, columns 4-11 = pixels 8-15; see conv3x3_c32pp_kernel) every ds_read_b128 of every tap is
//     conflict-free (enumerated for the three trunk geometries; W + 2 leaves 1.4-1.9x the conflict-free cycles);
//   * same k order (channel-major, taps ascending), same six products smallest first, same accumulators as the q kernel => BIT-IDENTICAL output
//     (tests/test_gpu_np3.py asserts torch.equal against the q kernel on every trunk geometry, forward and data gradient).
struct RGeo {
    int gpi;                  // tiles per image
    int ntiles;               // tiles in all (images x gpi)
    int PW, patch_px;         // patch row length (W + 8), patch pixels reserved per piece (272 or 320: the kernel's PPX)
    int nchunks;              // 16-pixel DMA chunks of a patch actually used
    unsigned mgW, mgPW, mg_gpi;
    int toff64[9];            // byte offset of each tap's source pixel inside the patch (host-filled ints: scalar loads -- a byte table indexed at run
                              // time became global_load_ubyte + s_waitcnt vmcnt(0) at the head of every k tile, i.e. a wait for the weight loads)
};
template <int MT, int BN, int PPX, int WPS>
__global__ __launch_bounds__(256, WPS) void pconv_r_kernel(PConvP p, RGeo g) {
    constexpr int NP = 3;
    constexpr int GM = 16 * MT;
    constexpr int NI = BN / 64;
    constexpr int PLANE_A = PPX * 64;                            // bytes; a multiple of 512 (the fragment address trick below relies on it)
    static_assert(PLANE_A % 512 == 0 && 2 * PLANE_A < 65536, "piece offsets are ds_read immediates");
    constexpr int NCW = (PPX / 16 + 3) / 4;                      // DMA chunks of the patch a wave issues at most
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    ptile_of_block(bx, by);
    const PClass& pc = p.cls[0];
    const int n0 = by * BN;
    const int W = p.GW, H = p.GH, HW = H * W, PW = g.PW;
    const int gt = bx;                                            // tile index: (image, tile of the image)
    const int img = fast_div(gt, g.gpi, g.mg_gpi);
    const int tp0 = (gt - img * g.gpi) * GM;                      // first pixel of the tile inside its image
    const int tend = tp0 + GM < HW ? tp0 + GM : HW;
    const int r0 = fast_div(tp0, W, g.mgW);                       // first image row of the tile; the patch starts one row above, one column left
    const int nkc = p.GC >> 5;

    // ---- patch DMA: chunk c = 16 patch pixels x 64 bytes of one piece; lane -> (pixel c * 16 + lane / 4, PHYSICAL slot lane & 3), fetched from the
    //      logical slot  phys ^ sw(pixel)  of the source pixel (the swizzle is applied to the source address: the LDS side of a DMA is lane-linear)
    constexpr unsigned OOB = 0xfffffff0u;
    unsigned poff[NCW];
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
        const int c = w4 + 4 * i;
        poff[i] = OOB;
        if (c < g.nchunks) {
            const int q = c * 16 + (lane >> 2);
            const int pr = fast_div(q, PW, g.mgPW), pcx = q - pr * PW;
            const int y = r0 - 1 + pr, x = pcx - 1;
            const int slot = (lane & 3) ^ ((q >> 1) & 2);
            if (y >= 0 && y < H && x >= 0 && x < W) poff[i] = (unsigned)(((((long)img * H + y) * W + x) * p.GC + slot * 8) * 2);
        }
    }
    auto stage_patch = [&](int cc) {
        const unsigned coff = (unsigned)(cc << 6);               // 32 channels = 64 bytes
#pragma unroll
        for (int i = 0; i < NCW; ++i) {
            const int c = w4 + 4 * i;
            if (c < g.nchunks) {                                 // wave-uniform
                const unsigned vo = poff[i] == OOB ? OOB : poff[i] + coff;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(p.a.p + q * p.a.ps), 0, p.a_bytes, 0x00020000),
                                                             (lds_ptr_t)(smem + q * PLANE_A + c * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
    };

    f32x4_t acc[MT][NI];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15, kp = lane >> 4;
    const int pxo = l15 < 4 ? l15 : (l15 < 12 ? l15 + 4 : l15 - 8);          // fragment column -> pixel of its 16-pixel row tile (bank pattern, see above)
    unsigned boff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + w4 * (BN / 4) + j * 16 + l15;
        boff[j] = n < p.N ? (unsigned)(((long)n * p.K + kp * 8) * 2) : OOB;
    }
    // byte offset (from smem) of this lane's 16 bytes of its pixel of row tile i at tap offset 0, BEFORE the slot swizzle:  v = pp * 64 + kp * 16;
    // the swizzled offset is  v ^ ((v >> 3) & 32)  -- bit 8 of v is bit 2 of the patch pixel pp (the piece planes are multiples of 512 bytes), bit 5
    // is the high bit of the slot: slot ^ 2 ((pp >> 2) & 1) without recomputing pp
    int vb[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int pix = tp0 + i * 16 + pxo;
        if (pix >= tend) pix = tend - 1;                          // clamp: inside the patch, result discarded
        const int oy = fast_div(pix, W, g.mgW), ox = pix - oy * W;
        vb[i] = ((oy - r0) * PW + ox) * 64 + kp * 16;
    }
    bf16x8_t af[NP][MT], bfb[2][NP][NI];                        // the weight fragments of TWO k tiles: the next one's are in flight a whole k tile ahead

    auto load_b = [&](auto SET, int ti, int cc) {
        constexpr int sb = decltype(SET)::value;
        const int kb2 = (pc.tap[ti] * p.GC + (cc << 5)) * 2;     // bytes
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(p.b.p + q * p.b.ps), 0, p.b_bytes, 0x00020000),
                                                                        (int)boff[j], kb2, 0);
                bfb[sb][q][j] = __builtin_bit_cast(bf16x8_t, v);
            }
    };
    auto load_frags = [&](int ti) {
        const int toff64 = g.toff64[ti];
        constexpr int QO[3] = {2, 0, 1};                         // in the order the products need them: piece 2 (first product), then 0, then 1
#pragma unroll
        for (int qi = 0; qi < NP; ++qi)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int v = vb[i] + toff64;
                af[QO[qi]][i] = *reinterpret_cast<const bf16x8_t*>(smem + (v ^ ((v >> 3) & 32)) + QO[qi] * PLANE_A);
            }
    };
    auto compute = [&](auto SET) {
        constexpr int sb = decltype(SET)::value;
        if (p.dbg & 2) {                                         // ablation bit 1 (ha2g_conv_planes_debug): no MFMA -- the fragments stay live
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(af[q][i]));
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(bfb[sb][q][j]));
            }
            return;
        }
        constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};          // smallest products first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfb[sb][QB[t]][j], af[QA[t]][i], acc[i][j], 0, 0, 0);
    };

    if (p.dbg & 32) return;                                      // ablation bit 5: launch cost only
    const int nk = (p.dbg & 4) ? 0 : pc.ntaps * nkc;             // ablation bit 2: no k loop (launch + prologue + epilogue remain)
    const bool dma = !(p.dbg & 1);                               // ablation bit 0: no DMA / weight loads after the prologue's
    // The waves that share a SIMD belong to different workgroups running the same code: left alone they fall into step (two waves that both want the
    // matrix pipe interleave their MFMAs, finish their k tile together and then read fragments together: the pipe idles while both read).  A STATIC
    // priority by hardware wave slot breaks the tie: the odd slot's MFMAs go first, it reaches its read phase while the even slot computes.
    if (!(p.dbg & 64) && (__builtin_amdgcn_s_getreg(0x1804) & 1)) __builtin_amdgcn_s_setprio(2);          // HW_REG_HW_ID[3:0] = wave slot of the SIMD
    stage_patch(0);
    load_b(std::integral_constant<int, 0>{}, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int ti = 0, cc = 0;
    auto step = [&](auto CUR, int k) {
        constexpr int cur = decltype(CUR)::value;
        int tn = ti + 1, cn = cc;
        if (tn == pc.ntaps) { tn = 0; ++cn; }
        if (k + 1 < nk && dma) load_b(std::integral_constant<int, 1 - cur>{}, tn, cn);   // the NEXT k tile's weight fragments: a whole k tile ahead of their use
        __builtin_amdgcn_sched_barrier(0);
        load_frags(ti);
        const bool swap = tn == 0 && cn < nkc && k + 1 < nk;      // last tap of a slice: the patch is replaced behind its reads
        if (swap) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // every wave's reads of this slice are complete
            __builtin_amdgcn_sched_barrier(0);
            if (dma) stage_patch(cn);                                                   // flies under this tap's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        compute(CUR);
        if (swap) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");              // the new slice has landed (all waves' chunks)
            __builtin_amdgcn_sched_barrier(0);
        }
        ti = tn; cc = cn;
    };
    for (int k = 0; k < nk; k += 2) {
        step(std::integral_constant<int, 0>{}, k);
        if (k + 1 < nk) step(std::integral_constant<int, 1>{}, k + 1);
    }

    // ---- epilogue: D = W X^T, a lane holds four consecutive output channels of one pixel -> one 16-byte store per 16x16 tile (as the q kernel) ----
    if (p.dbg & 8) return;                                       // ablation bit 3: no output stores
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = tp0 + i * 16 + pxo;
        if (pix >= tend) continue;
        const long orow = (long)img * HW + pix;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col0 = n0 + w4 * (BN / 4) + j * 16 + 4 * kp;
            if (col0 >= p.N) continue;
            f32x4_t v = acc[i][j];
            float* dst = p.C + orow * p.ldc + col0;
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (p.beta != 0.f) v += p.beta * *reinterpret_cast<const f32x4_t*>(dst);
            if (p.rsd != nullptr) {
                // masked residual (round 6; the identity shortcut under autograd, ResNetBlocks.py:34-36): + (out > 0 ? dout : 0), decisions from the forward
                // tail's bit words -- what "beta = 1 onto dres" added, without dres ever being written
                const long e = orow * p.ldc + col0, vi = e >> 2;
                const f32x4_t dd = *reinterpret_cast<const f32x4_t*>(p.rsd + e);
                const unsigned b = (p.rsd_bits[vi >> 3] >> (4 * (int)(vi & 7))) & 15u;
                v[0] += (b & 1u) ? dd[0] : 0.f; v[1] += (b & 2u) ? dd[1] : 0.f; v[2] += (b & 4u) ? dd[2] : 0.f; v[3] += (b & 8u) ? dd[3] : 0.f;
            }
            *reinterpret_cast<f32x4_t*>(dst) = v;
        }
    }
    // ---- BatchNorm statistics of the tile (forward, the BatchNorm that follows the convolution: ResNetBlocks.py:24-29): the separate column pass
    //      over the output (norm.hip col_partial_kernel<0>, 0.9 ms of the step's main queue) re-read what this epilogue holds in registers.
    //      Per lane the column sums of its MT <= 9 pixels (fp32: nine terms), then in double: a reduce-scatter over the 16 pixel lanes, one double per
    //      (channel, which) and tile; the tiles are added in double by bn_stats_final_kernel.  (All-double lane sums cost 5.5 us per launch; the lane
    //      sums are shifted instead, see below.)
    if (p.stat != nullptr && p.bsx != nullptr) {
        // BatchNorm-BACKWARD statistics of the gradient this launch produces (bn1 behind conv2's data gradient, ResNetBlocks.py:24-29 under autograd): the
        // column pass col_partial_kernel<1> re-read dy (what this epilogue holds in registers) and the BatchNorm's input; here only the input tile is
        // read.  Lane sums in double, the arithmetic of that pass ((double) x - mean) * invstd -- sum(dy * xhat) cancels by 1e3..1e4 on nearly dead
        // channels --, then the same reduce-scatter and tile layout as the forward sums: norm.hip's pair_final_kernel adds the tiles in order.
        constexpr int NV = NI * 8;
        double sv[NV];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col0 = n0 + w4 * (BN / 4) + j * 16 + 4 * kp;
            f32x4_t m4 = {0.f, 0.f, 0.f, 0.f}, i4 = m4;
            if (col0 < p.N) { m4 = *reinterpret_cast<const f32x4_t*>(p.bsmean + col0); i4 = *reinterpret_cast<const f32x4_t*>(p.bsinv + col0); }
            double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int pix = tp0 + i * 16 + pxo;
                if (pix < tend && col0 < p.N) {
                    const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(p.bsx + ((long)img * HW + pix) * p.ldc + col0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double d = (double)acc[i][j][u];
                        s1[u] += d;
                        s2[u] += d * (((double)xv[u] - (double)m4[u]) * (double)i4[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { sv[(j * 4 + u) * 2] = s1[u]; sv[(j * 4 + u) * 2 + 1] = s2[u]; }
        }
        row16_reduce_scatter<NV>(sv, lane);
        const int idx = l15 & (NV - 1);
        const int ch = n0 + w4 * (BN / 4) + (idx >> 3) * 16 + 4 * kp + ((idx >> 1) & 3);
        if ((NV == 16 || l15 < 8) && ch < p.N) p.stat[((long)(idx & 1) * p.N + ch) * p.stat_nblk + gt] = sv[0];
    } else
    if (p.stat != nullptr) {
        constexpr int NV = NI * 8;
        double sv[NV];
        bool live[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) live[i] = tp0 + i * 16 + pxo < tend;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // SHIFTED lane sums: the fp32 part runs on v - K, K = the lane's first pixel of this channel, so its rounding is relative to
                // the channel's SPREAD, not to its mean -- raw sums of squares lose (1 + mean^2 / var) x 5e-7 of the variance when the finalize pass
                // forms s2 / n - mean^2 (a channel with |mean| = 30 sigma: 5e-4).  Back to raw sums in double: S1 = s1 + n K, S2 = s2 + K (2 s1 + n K).
                float kf = acc[0][j][u];
                if (p.relu) kf = fmaxf(kf, 0.f);
                kf = live[0] ? kf : 0.f;
                float s1 = 0.f, s2 = 0.f;
                int nl = 0;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float v = acc[i][j][u];
                    if (p.relu) v = fmaxf(v, 0.f);
                    v = live[i] ? v - kf : 0.f;
                    nl += live[i] ? 1 : 0;
                    s1 += v; s2 = fmaf(v, v, s2);
                }
                const double kd = (double)kf, nk = (double)nl * kd;
                sv[(j * 4 + u) * 2] = (double)s1 + nk; sv[(j * 4 + u) * 2 + 1] = (double)s2 + kd * (2.0 * (double)s1 + nk);
            }
        row16_reduce_scatter<NV>(sv, lane);
        const int idx = l15 & (NV - 1);
        const int ch = n0 + w4 * (BN / 4) + (idx >> 3) * 16 + 4 * kp + ((idx >> 1) & 3);
        if ((NV == 16 || l15 < 8) && ch < p.N) p.stat[((long)(idx & 1) * p.N + ch) * p.stat_nblk + gt] = sv[0];
    }
}

// Weight gradient of a 3x3 / stride-1 / pad-1 convolution from planes:  dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci].
// The implicit GEMM (gemm.hip, A_MC x B_IM) stages the im2col gather of x -- the same pixels nine times -- and both operands once per
// 128-wide output tile; rocprofv3 puts it at 15 VALU per MFMA (split + gather arithmetic) and, in the step, bound by L2 -> LDS traffic beside
// the data-gradient stream.  Here a workgroup owns a 64 (co) x 64 (ci) block of dW for ALL nine taps (each wave a 32 x 32 corner = nine
// accumulators, 144 AGPRs, as in conv3x3_c32_wgrad_kernel) and walks a range of 64-pixel tiles: per tile the dy strip [64 px][64 co] and the
// x patch [(rows + halo) x (W + 2) px][64 ci] go global -> LDS ONCE, by DMA, as bf16 hi / lo planes (zero padding = the zero page), and every
// MFMA operand is a pair of transpose reads (ds_read_b64_tr_b16: k = pixel is the slow index of both operands); the x fragment of tap
// (kh, kw) is the same read kh (W + 2) + kw patch rows further down.  128-byte LDS rows; the two 64-byte halves of a row are swapped on
// rows with bit 1 set so that the four pixel rows of a transpose read cover all 64 banks (the swap is applied to the DMA's source address).
// Tiles are double-buffered: tile i + 1 streams in while tile i's 108 MFMAs per wave run.  Each workgroup writes ONE partial block into its
// chunk's dW-shaped slab; gemm.hip's wide reduce adds the chunks in double.
struct PWgradP {
    const unsigned short* x_hi; const unsigned short* x_lo;
    const unsigned short* dy_hi; const unsigned short* dy_lo;
    float* part;                        // [nchunks][Cout][9][Cin]
    int N, H, W, Cin, Cout;
    int tpi, tiles_per_chunk, patch_rows;      // tiles per image; consecutive tiles per workgroup; LDS rows reserved for one patch plane (multiple of 8)
    int dbg;                            // timing ablation (ha2g_conv_planes_debug): 1 = no DMA after the first tile (a branch around the MFMAs would
                                        // split the basic block the read / MFMA interleave is scheduled in)
};

// WT = pixels per tile (k of one stage): 64, or 48 where the image has a multiple of 48 but not of 64 pixels (layer 4: 16 x 9 = 144).
// LDS image of one tile (64 KB, two of them): [dy hi | dy lo | x hi | x lo], every plane as TWO sub-planes of 32 channels with 64-byte rows
// ([half][pixel][32 ch]): a wave works on one half of dy (its 32 co) and one half of x (its 32 ci), the four pixel rows of a transpose read are
// then 256 contiguous bytes -- conflict-free with NO swizzle, so a fragment address is lane base + tap offset (one add; the first version's
// XOR-swizzled 128-byte rows cost ~7 VALU per MFMA in address arithmetic and the kernel was instruction-issue bound at 3x the MFMA floor).
constexpr int PW_ROWS = 192;                                   // patch rows reserved per sub-plane (>= rows * (W + 2); checked by the host)
constexpr int X_SUB = PW_ROWS * 64;                            // bytes of one x sub-plane

// Twelve waves: wave = corner (32 co x 32 ci of the 64 x 64 block) + 4 * tap row (kh = 0, 1, 2: three taps each).  Three waves per SIMD share
// the matrix pipe.  (Four waves with all nine taps each = 144 accumulator registers: rocprofv3 showed the pipe 36 % busy, a third of the wave's
// time parked at the tile boundary with nothing else to issue from.)  Every wave issues a FIXED number of DMA instructions per tile (one
// 16-row piece of the patch in its four sub-planes, two pieces of the dy strip): with a data-dependent count hipcc put s_waitcnt vmcnt(0) in
// front of every fragment read, i.e. waited for the NEXT tile's DMA before computing on this one.
// NP = 2: hi + lo planes (three MFMAs per product); NP = 1: x and dy ARE bf16 tensors (bf16-storage mode): one plane, one MFMA.
template <int WT, int NP>
__global__ __launch_bounds__(768) void pconv_wgrad_kernel(PWgradP p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char wsm[];
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
    // NP = 3 (round 4, fp32-class): three pieces per operand and six MFMAs per product.  Three x planes of a 64-channel block no longer fit twice
    // into the LDS, so a workgroup owns a 64 (co) x 32 (ci) block of dW (ONE 32-channel half of x: CIH = 1) and the twelve waves are
    // 2 co halves x 3 tap rows x 2 K HALVES: the waves of k half h run the 16-pixel chunks h * NKC/2 .. of every tile and write their own partial
    // slab (the reduce adds slabs anyway) -- 36 MFMAs per wave and tile, 60 KB of DMA per tile, as many as the two-piece form has.
    constexpr int CIH = NP == 3 ? 1 : 2;                       // 32-channel halves of x per workgroup
    constexpr int KSP = NP == 3 ? 2 : 1;                       // k (pixel chunk) split over wave groups
    constexpr int NKC = WT / 16;
    constexpr int DY_SUB = ((WT + 15) / 16) * 1024;            // bytes of one dy sub-plane (WT rows of 64 bytes, in 16-row DMA pieces)
    constexpr int DY_ALL = 2 * NP * DY_SUB, BUF = DY_ALL + CIH * NP * X_SUB;
    constexpr int NDY = 2 * NP * (WT / 16);                    // dy DMA pieces per tile (plane, half, 16 pixels): 24 / 18 (NP = 3), 16 / 12 (NP = 2), 8 / 6 (NP = 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NP <= 2: wave = corner (wm, wn) + 4 * tap row.  NP = 3: wave = wm + 2 * tap row + 6 * k half, wn = 0.
    const int wm = NP == 3 ? (wave & 1) : ((wave >> 1) & 1), wn = NP == 3 ? 0 : (wave & 1);
    const int kh = NP == 3 ? ((wave >> 1) % 3) : (wave >> 2), khalf = NP == 3 ? wave / 6 : 0;
    const int l31 = lane & 31, lhi = lane >> 5, g4 = lane >> 4, q16 = lane & 15;
    const int krow = 8 * (g4 >> 1) + (q16 >> 2), moff = 16 * (g4 & 1) + 4 * (q16 & 3);
    const int HW = p.H * p.W, PW = p.W + 2;
    constexpr int CIB = 32 * CIH;                               // ci block of a workgroup
    const int ncit = (p.Cin + CIB - 1) / CIB;
    const int co0 = ((int)blockIdx.y / ncit) << 6, ci0 = ((int)blockIdx.y % ncit) * CIB;
    const int tiles = p.N * p.tpi;
    const int t_beg = (int)blockIdx.x * p.tiles_per_chunk;
    const int t_end = t_beg + p.tiles_per_chunk < tiles ? t_beg + p.tiles_per_chunk : tiles;
    const unsigned short* zero = g_zero_page;
    const long xps = p.x_lo - p.x_hi, dps = p.dy_lo - p.dy_hi;  // plane strides (elements); piece q at hi + q * stride (equally spaced planes)

    auto tr = [](const unsigned char* ptr) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(__attribute__((address_space(3))) const unsigned char*)ptr);
    };
    auto frag = [&](const unsigned char* q0, const unsigned char* q1) {          // k rows krow .. +3 (q0) and krow + 4 .. + 7 (q1)
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(tr(q0), tr(q1), 0, 1, 2, 3, 4, 5, 6, 7));
    };

    const int drow = lane >> 2, dpc = lane & 3;                                   // DMA: 16 rows x 4 sixteen-byte pieces per wave instruction
    // this lane's patch pixel: piece `wave`, row drow -> padded coordinates (fixed for every tile: the patch geometry depends on W only)
    const int pp = wave * 16 + drow;
    const int ppr = pp / PW, ppx = pp - ppr * PW;
    auto stage = [&](int tile, int buf) {
        const int img = tile / p.tpi, p0 = (tile - img * p.tpi) * WT;
        const int pend = p0 + WT < HW ? p0 + WT : HW;
        const int r0 = p0 / p.W, rows = (pend - 1) / p.W - r0 + 3;              // + the halo row above and below
        unsigned char* dst = wsm + buf * BUF;
        // dy strip: NDY pieces = (plane, half, 16 pixels); wave w stages pieces w and w + 12 (the waves past the end repeat their first)
#pragma unroll
        for (int i = 0; i < (NDY + 11) / 12; ++i) {
            const int q = (wave + 12 * i) < NDY ? wave + 12 * i : wave % NDY;     // = sub * (WT / 16) + piece, sub = plane * 2 + half
            const int sub = q / (WT / 16), piece = q - sub * (WT / 16);
            const int pix = p0 + piece * 16 + drow;
            const bool on = pix < HW && co0 + (sub & 1) * 32 < p.Cout;           // Cout = 32: the upper half is zeros
            const long o = ((long)img * HW + pix) * p.Cout + co0 + (sub & 1) * 32 + dpc * 8;
            const unsigned short* g = on ? p.dy_hi + (sub >> 1) * dps + o : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_ptr_t)(dst + sub * DY_SUB + piece * 1024), 16, 0, 0);
        }
        // x patch: piece `wave` (16 padded pixels) in its CIH * NP sub-planes; pieces past rows * PW carry zeros (never read)
        {
            const int gy = r0 - 1 + ppr;
            const bool on = ppr < rows && gy >= 0 && gy < p.H && ppx >= 1 && ppx <= p.W;
            const bool on1 = on && ci0 + 32 < p.Cin;                               // Cin = 32: the upper half is zeros
            const long o = (((long)img * p.H + gy) * p.W + (ppx - 1)) * p.Cin + ci0 + dpc * 8;
            unsigned char* d = dst + DY_ALL + wave * 1024;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                __builtin_amdgcn_global_load_lds((gptr_t)(on ? p.x_hi + q * xps + o : zero), (lds_ptr_t)(d + (q * CIH) * X_SUB), 16, 0, 0);
                if (CIH == 2) __builtin_amdgcn_global_load_lds((gptr_t)(on1 ? p.x_hi + q * xps + o + 32 : zero), (lds_ptr_t)(d + (q * CIH + 1) * X_SUB), 16, 0, 0);
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment sets of one 16-pixel k chunk: the dy fragment pieces and this wave's three taps' x fragment pieces.  NP <= 2: two sets (chunk
    // kc + 1 is read from LDS while chunk kc's MFMAs run); NP = 3: one set (three waves per SIMD leave 168 registers per lane).
    struct FragSet { bf16x8_t a[NP], b[3][NP]; };
    const int a_lane = wm * DY_SUB + 2 * moff;                                    // this wave's dy half + this lane's channel offset
    const int b_lane = DY_ALL + wn * X_SUB + 2 * moff + kh * PW * 64;             // its x half, channel offset and tap row
    auto load_set = [&](FragSet& f, int cur, int p0, int r0, int kc) {
        const unsigned char* base = wsm + cur * BUF;
        const int k0 = 16 * kc + krow, k1 = k0 + 4;                               // this lane's two tile-local pixels
        const unsigned char* a0 = base + a_lane + k0 * 64;
        const unsigned char* a1 = base + a_lane + k1 * 64;
#pragma unroll
        for (int q = 0; q < NP; ++q) f.a[q] = frag(a0 + 2 * q * DY_SUB, a1 + 2 * q * DY_SUB);
        int q0 = p0 + k0, q1 = p0 + k1;
        if (q0 >= HW) q0 = HW - 1;                                                 // clamp: stays inside the patch; dy is zero there
        if (q1 >= HW) q1 = HW - 1;
        const int y0 = q0 / p.W, y1 = q1 / p.W;
        const unsigned char* b0 = base + b_lane + ((y0 - r0) * PW + (q0 - y0 * p.W)) * 64;   // patch row of tap (kh, 0)
        const unsigned char* b1 = base + b_lane + ((y1 - r0) * PW + (q1 - y1 * p.W)) * 64;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int q = 0; q < NP; ++q) f.b[t][q] = frag(b0 + t * 64 + CIH * q * X_SUB, b1 + t * 64 + CIH * q * X_SUB);
    };
    // passes over the taps, smallest products first: consecutive MFMAs go to DIFFERENT accumulators; each accumulator still sees its products
    // in the same order (NP = 2: lo*hi, hi*lo, hi*hi as in round 3; NP = 3: the six products of pconv_kernel)
    auto mfma_set = [&](const FragSet& f) {
        if constexpr (NP == 3) {
            constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[QA[u]], f.b[t][QB[u]], acc[t], 0, 0, 0);
        } else {
            if constexpr (NP == 2) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[NP - 1], f.b[t][0], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[t][NP - 1], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[t][0], acc[t], 0, 0, 0);
        }
    };

    if (t_beg < t_end) stage(t_beg, 0);
    wait_vm_lds_barrier<0>();            // explicit: the DMA of every wave has landed (do not lean on the compiler's alias wait)
    FragSet fs0, fs1;
    constexpr int KC0 = NKC / KSP + (NKC % KSP);                // chunks of k half 0 (NKC = 3: 2 + 1)
    const int kc_beg = khalf == 0 ? 0 : KC0, kc_end = (KSP == 1 || khalf == 1) ? NKC : KC0;
    for (int tile = t_beg; tile < t_end; ++tile) {
        const int cur = (tile - t_beg) & 1;
        if (tile + 1 < t_end && !(p.dbg & 1)) stage(tile + 1, cur ^ 1);
        const int p0 = (tile % p.tpi) * WT;
        const int r0 = p0 / p.W;
        if constexpr (NP == 3) {
            for (int kc = kc_beg; kc < kc_end; ++kc) {
                load_set(fs0, cur, p0, r0, kc);
                mfma_set(fs0);
            }
        } else {
            load_set(fs0, cur, p0, r0, 0);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                FragSet& use = (kc & 1) ? fs1 : fs0;
                FragSet& nxt = (kc & 1) ? fs0 : fs1;
                if (kc + 1 < NKC) load_set(nxt, cur, p0, r0, kc + 1);
                mfma_set(use);
            }
        }
        wait_vm_lds_barrier<0>();        // the next tile has landed and every wave is done with this one
    }
    // ---- this wave's three taps of its corner of the chunk's slab: part[chunk][co][tap][ci] ----
    // NP = 3 (round 6): the two k halves of a workgroup are added HERE, through the dead staging buffers (half 1's accumulators -> LDS, barrier, half 0
    // adds them onto its own: fixed order), and ONE slab per workgroup leaves the CU -- the wide reduce reads half as many slabs (37.7 -> 18.9 MB per
    // launch, 0.87 GB per train step less written and re-read).
    if constexpr (KSP == 2) {
        float* xl = reinterpret_cast<float*>(wsm) + (long)(wave % 6) * 3 * 16 * 64 + lane;
        if (khalf == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) xl[(t * 16 + r) * 64] = acc[t][r];
        }
        __syncthreads();
        if (khalf == 1) return;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += xl[(t * 16 + r) * 64];
    }
    float* out = p.part + (long)blockIdx.x * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (co < p.Cout && ci0 + wn * 32 + l31 < p.Cin) out[((long)co * 9 + kh * 3 + t) * p.Cin + ci0 + wn * 32 + l31] = acc[t][r];
        }
}

#define HA2G_PCONV_RING_DEFAULT 2
static int g_pdbg = 0;
static int g_planes = 1;         // ha2g_conv_planes_enable: 0 = callers keep the round-2 kernels (A/B switch, HA2G_PLANES=0)

}  // namespace

// Partials of the plane-based weight gradient: returns the number of dW-shaped slabs written to `part` (the caller reduces them),
// -100 when the geometry is not served (caller keeps the implicit GEMM), < 0 on error.
static int pwgrad_tile(int HW) { return (HW % 64 != 0 && HW % 48 == 0) ? 48 : 64; }
int g_side_cus = 256;      // persistent weight-gradient kernels (side stream) size their grids for at most this many compute units

int pconv_wgrad_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!(g_planes && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin % 32 == 0 && Cout % 32 == 0 && (long)H * W >= 16)) return 0;
    const int wt = pwgrad_tile(H * W);
    return ((wt + W - 2) / W + 1 + 2) * (W + 2) <= PW_ROWS;
}
// np = 3: a workgroup owns a 64 x 32 block (one 32-channel half of x) and writes two slabs (the two k halves of its waves)
static int pwgrad_pairs(int Cin, int Cout, int np) { return ((Cin + (np == 3 ? 31 : 63)) / (np == 3 ? 32 : 64)) * ((Cout + 63) / 64); }
long pconv_wgrad_workspace_bytes_np(int N, int H, int W, int Cin, int Cout, int np) {
    const int wt = pwgrad_tile(H * W);
    const long tiles = (long)N * (((long)H * W + wt - 1) / wt);
    const int npairs = pwgrad_pairs(Cin, Cout, np);
    long nchunks = 256 / npairs < 1 ? 1 : 256 / npairs;
    if (nchunks > tiles) nchunks = tiles;
    return nchunks * (np == 3 ? 2 : 1) * (long)Cout * 9 * Cin * 4;
}
long pconv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {          // enough for every piece count
    const long a = pconv_wgrad_workspace_bytes_np(N, H, W, Cin, Cout, 2), b = pconv_wgrad_workspace_bytes_np(N, H, W, Cin, Cout, 3);
    return a > b ? a : b;
}
// np pieces per operand: piece q of x at x + q * x_ps, of dy at dy + q * dy_ps (elements); np = 1: bf16-storage mode (the tensors themselves)
int pconv_wgrad_launch_np(const void* x, long x_ps, const void* dy, long dy_ps, int np, float* part, int N, int H, int W, int Cin, int Cout,
                          hipStream_t st) {
    PWgradP p{};
    p.x_hi = (const unsigned short*)x; p.x_lo = p.x_hi + x_ps;
    p.dy_hi = (const unsigned short*)dy; p.dy_lo = p.dy_hi + dy_ps;
    p.part = part; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.dbg = g_pdbg;
    const int HW = H * W, wt = pwgrad_tile(HW);
    p.tpi = (HW + wt - 1) / wt;
    const long tiles = (long)N * p.tpi;
    const int npairs = pwgrad_pairs(Cin, Cout, np);
    int cus = hw_cu_count(), dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus > 256) cus = 256;                                   // the workspace query assumes at most 256 chunks x pairs
    if (cus > g_side_cus) cus = g_side_cus;                     // ha2g_side_cus: leave compute units to the main queue's kernels
    long nchunks = cus / npairs < 1 ? 1 : cus / npairs;
    if (nchunks > tiles) nchunks = tiles;
    p.tiles_per_chunk = (int)((tiles + nchunks - 1) / nchunks);
    nchunks = (tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
    const int rows_max = (wt + W - 2) / W + 1 + 2;
    p.patch_rows = rows_max * (W + 2);
    if (p.patch_rows > PW_ROWS) return -100;                     // the patch does not fit the reserved sub-planes
    const int cih = np == 3 ? 1 : 2;
    const size_t lds = (size_t)2 * (2 * np * (size_t)((wt + 15) / 16) * 1024 + cih * np * (size_t)X_SUB);
    static bool attr_set[64] = {false};                          // per device
    if (!attr_set[dev]) {
        const void* fns[6] = {reinterpret_cast<const void*>(pconv_wgrad_kernel<64, 2>), reinterpret_cast<const void*>(pconv_wgrad_kernel<48, 2>),
                              reinterpret_cast<const void*>(pconv_wgrad_kernel<64, 1>), reinterpret_cast<const void*>(pconv_wgrad_kernel<48, 1>),
                              reinterpret_cast<const void*>(pconv_wgrad_kernel<64, 3>), reinterpret_cast<const void*>(pconv_wgrad_kernel<48, 3>)};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
                return ha2g_set_error(-2, "pconv_wgrad: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    const dim3 grid((unsigned)nchunks, npairs);
    if (np == 3) {
        if (wt == 64) hipLaunchKernelGGL((pconv_wgrad_kernel<64, 3>), grid, dim3(768), lds, st, p);
        else hipLaunchKernelGGL((pconv_wgrad_kernel<48, 3>), grid, dim3(768), lds, st, p);
    } else if (np == 2) {
        if (wt == 64) hipLaunchKernelGGL((pconv_wgrad_kernel<64, 2>), grid, dim3(768), lds, st, p);
        else hipLaunchKernelGGL((pconv_wgrad_kernel<48, 2>), grid, dim3(768), lds, st, p);
    } else {
        if (wt == 64) hipLaunchKernelGGL((pconv_wgrad_kernel<64, 1>), grid, dim3(768), lds, st, p);
        else hipLaunchKernelGGL((pconv_wgrad_kernel<48, 1>), grid, dim3(768), lds, st, p);
    }
    HA2G_CHECK_LAUNCH("pconv_wgrad");
    return (int)nchunks;                                         // slabs written (np = 3: the two k halves of a workgroup are added in the kernel)
}
// x_lo == dy_lo == nullptr: bf16-storage mode (one plane)
int pconv_wgrad_launch(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* part, int N, int H, int W, int Cin, int Cout,
                       hipStream_t st) {
    const int np = (x_lo == nullptr && dy_lo == nullptr) ? 1 : 2;
    return pconv_wgrad_launch_np(x_hi, np == 1 ? 0 : (const unsigned short*)x_lo - (const unsigned short*)x_hi, dy_hi,
                                 np == 1 ? 0 : (const unsigned short*)dy_lo - (const unsigned short*)dy_hi, np, part, N, H, W, Cin, Cout, st);
}

// parity classes of a data gradient (stride 1: one class with every tap)
static int dgrad_classes(PConvP& p, int N, int H, int W, int KH, int KW, int stride, int pad, int OWd) {
    int maxM = 0;
    p.ncls = 0;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            PClass c{};
            c.py = py; c.px = px;
            c.OHc = (H - py + stride - 1) / stride; c.OWc = (W - px + stride - 1) / stride;
            c.M = N * c.OHc * c.OWc;
            c.ntaps = 0; c.kh0 = -1; c.kw0 = -1;
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) {
                    if ((py + pad - kh) % stride != 0 || (px + pad - kw) % stride != 0) continue;       // operands >= -2: C's % keeps the sign, 0 stays 0
                    if (c.kh0 < 0) { c.kh0 = kh; c.kw0 = kw; }
                    c.tap[c.ntaps] = kh * KW + kw;
                    c.doff[c.ntaps] = -(((kh - c.kh0) / stride) * OWd + (kw - c.kw0) / stride);
                    ++c.ntaps;
                }
            if (c.ntaps == 0 || c.M == 0) continue;
            if (c.M > maxM) maxM = c.M;
            p.cls[p.ncls++] = c;
        }
    return maxM;
}
// tile choice by the output width: 128 x 128 when N is a multiple of 128, else 256 x 64 (N = 32: the upper half of the column tile idles)
static int g_ring = 0;       // ha2g_conv_planes_ring: 0 = default depth per tile shape, 2 / 3 / 4 = forced (A/B)

template <int BM, int BN, int WM, int WN, int NP, int OUT, int S>
static int pconv_launch_s(const PConvP& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)S * NP * (BM * 64 + BN * 64);
    static bool attr_set[64] = {false};                          // per instantiation and per device (the attribute belongs to the device's code object)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(pconv_kernel<BM, BN, WM, WN, NP, OUT, S>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ha2g_set_error(-2, "pconv: cannot raise the dynamic LDS limit to %zu bytes", lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pconv_kernel<BM, BN, WM, WN, NP, OUT, S>), grid, dim3(64 * WM * WN), lds, st, p);
    return 0;
}
template <int BM, int BN, int WM, int WN, int NP, int OUT>
static int pconv_launch(const PConvP& p, dim3 grid, hipStream_t st) {
    constexpr int per_stage = NP * (BM * 64 + BN * 64);
    constexpr int SMAX = 160 * 1024 / per_stage >= 4 ? 4 : (160 * 1024 / per_stage >= 3 ? 3 : 2);
    int s = g_ring ? g_ring : HA2G_PCONV_RING_DEFAULT;
    if (s > SMAX) s = SMAX;
    if (s >= 4) { if constexpr (SMAX >= 4) return pconv_launch_s<BM, BN, WM, WN, NP, OUT, 4>(p, grid, st); }
    if (s == 3) { if constexpr (SMAX >= 3) return pconv_launch_s<BM, BN, WM, WN, NP, OUT, 3>(p, grid, st); }
    return pconv_launch_s<BM, BN, WM, WN, NP, OUT, 2>(p, grid, st);
}
static int g_waves = 4;      // ha2g_conv_planes_waves: 8 = eight-wave workgroups with twice the tile (256 x 128 / 512 x 64), one per CU

static int g_tile3 = 0;      // ha2g_conv_planes_tile3: tile of the three-piece kernel, 0 = default per shape, 1 = 128 x 128, 2 = 256 x 64, 3 = 128 x 64,
                             // 4 = the ping-pong kernel (256 x 128 / 256 x 64, eight waves in two anti-phase groups)
template <int BN, int NP, int OUT>
static int pconv_pp_launch(const PConvP& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)4 * NP * 128 * 64 + (size_t)2 * NP * BN * 64;
    static bool attr_set[64] = {false};                          // per device (the attribute is per device, not per process)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pconv_pp_kernel<BN, NP, OUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ha2g_set_error(-2, "pconv_pp: cannot raise the dynamic LDS limit to %zu bytes", lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pconv_pp_kernel<BN, NP, OUT>), grid, dim3(512), lds, st, p);
    return 0;
}
static int g_q_classes = 1;  // stride-2 data gradients (up to four parity classes over grid.z) on the q kernel too (ha2g_conv_planes_tile3(7) = the 32x32 kernels, A/B)
static int g_qbuf = 1;       // buffer-addressed DMA in the q kernel (ha2g_conv_planes_bufaddr(0) = flat addresses + zero page, A/B)
static int g_kmaj = 1;       // k order of the q kernel's split products: 1 = channel-major (the nine taps of a 32-channel slice back to back), 0 = tap-major
static int g_q_kernel = 1;   // the quantisation-free 16x16 kernel where its tile choice fills the CUs better (ha2g_conv_planes_tile3(5) forces, (6) = off)
template <int MT, int BN, int NP>
static int pconv_q_launch(const PConvP& p, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)4 * NP * (16 * MT) * 64 + (size_t)2 * NP * BN * 64;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pconv_q_kernel<MT, BN, NP, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ha2g_set_error(-2, "pconv_q: cannot raise the dynamic LDS limit to %zu bytes", lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pconv_q_kernel<MT, BN, NP, 0>), grid, dim3(512), lds, st, p);
    return 0;
}
// Tile choice: rows per workgroup 2 x 16 MT (MT = 7, 8, 9) x BN columns (64 / 128) -- the combination that keeps the CUs fullest over the launch
// (workgroups / (rounds x CUs), times the useful fraction of the padded rows), larger tiles on ties.  -100: not served (caller falls back).
// tile plan of the q kernel for an M x N output computed in `ksplit` k slices: (MT, BN) that keeps the CUs fullest; returns the efficiency (0 = none)
static double pconv_q_plan(int M, int N, int ksplit, int* pmt, int* pbn) {
    const int cus = hw_cu_count();
    double best = -1.0; int bmt = 0, bbn = 0;
    for (int bn = 128; bn >= 64; bn -= 64) {
        const long tn = (N + bn - 1) / bn;
        for (int mt = 9; mt >= 7; --mt) {
            const long rows = 32L * mt, tm = (M + rows - 1) / rows, wgs = tm * tn * ksplit;
            const long rounds = (wgs + cus - 1) / cus;
            double eff = (double)wgs / (double)(rounds * cus) * ((double)M / (double)(tm * rows)) * ((double)N / (double)(tn * bn));
            eff *= (bn == 128 ? 1.0 : 0.93) * (mt == 9 ? 1.0 : (mt == 8 ? 0.985 : 0.97));        // longer COMPUTE phases amortise the per-phase overhead
            if (eff > best + 1e-9) { best = eff; bmt = mt; bbn = bn; }
        }
    }
    *pmt = bmt; *pbn = bbn;
    return best;
}
// element counts of one piece plane of A / B -> byte sizes for the q kernel's buffer resources (0 = too large for 32-bit byte offsets: flat path)
static void set_plane_bytes(PConvP& p, long a_elems, long b_elems) {
    const long lim = (1L << 30) - 64;
    p.a_bytes = (a_elems > 0 && a_elems < lim) ? (int)(a_elems * 2) : 0;
    p.b_bytes = (b_elems > 0 && b_elems < lim) ? (int)(b_elems * 2) : 0;
}
static int g_r_kernel = 1;   // patch-resident kernel for the 3x3 / stride-1 trunk convolutions (ha2g_conv_planes_tile3(8) = off: the q kernel, A/B)
template <int MT, int BN, int PPX, int WPS>
static int pconv_r_launch(const PConvP& p, const RGeo& g, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * PPX * 64;
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pconv_r_kernel<MT, BN, PPX, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ha2g_set_error(-2, "pconv_r: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pconv_r_kernel<MT, BN, PPX, WPS>), grid, dim3(256), lds, st, p, g);
    return 0;
}
// -100: geometry not served (the caller keeps the q kernel): 3x3 / stride 1 / pad 1 with one class of nine taps, N a multiple of 64 with aligned
// vector stores, no bias / activation epilogue (the convolution callers pass relu only), the patch inside one of the two reserved plane sizes
// tile plan of the patch-resident kernel for a 3x3 / stride-1 / pad-1 convolution of `imgs` images of H x W pixels with N output channels:
// false = geometry not served.  Depends on the geometry and the device's CU count only (ha2g_conv2d_fwd_planes_stat_blocks reports its tile count).
static bool pconv_r_plan(int imgs, int H, int W, int N, int& bmt, int& bbn, RGeo& bg);
static int pconv_r_dispatch(const PConvP& p_in, int imgs, hipStream_t st) {
    PConvP p = p_in;
    const PClass& c0 = p.cls[0];
    if (!g_r_kernel || p.ncls != 1 || c0.ntaps != 9 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.ksplit > 1 || p.bias || p.act) return -100;
    if (p.GH != p.OH || p.GW != p.OW || p.GC % 32 != 0 || p.N % 64 != 0 || p.a_bytes <= 0 || p.b_bytes <= 0) return -100;
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!(p.ldc % 4 == 0 && al16(p.C))) return -100;
    int bmt = 0, bbn = 0; RGeo bg{};
    if (!pconv_r_plan(imgs, p.GH, p.GW, p.N, bmt, bbn, bg)) return -100;
    if (p.stat != nullptr && p.stat_nblk != bg.ntiles) return ha2g_set_error(-1, "conv2d_fwd_planes: statistics buffer sized for %d row tiles, the kernel writes %d", p.stat_nblk, bg.ntiles);
    for (int t = 0; t < 9; ++t) {                                // forward: source pixel (oy - 1 + kh, ox - 1 + kw); data gradient: (oy + 1 - kh, ox + 1 - kw)
        const int kh = c0.tap[t] / 3, kw = c0.tap[t] % 3;
        bg.toff64[t] = (p.fwd ? kh * bg.PW + kw : (2 - kh) * bg.PW + (2 - kw)) * 64;
    }
    const dim3 grid((unsigned)bg.ntiles, (unsigned)(p.N / bbn), 1);
    const bool big = bg.patch_px == 320;
    if (bbn == 128) {
        if (bmt == 9) return big ? pconv_r_launch<9, 128, 320, 2>(p, bg, grid, st) : pconv_r_launch<9, 128, 272, 2>(p, bg, grid, st);
        return big ? pconv_r_launch<7, 128, 320, 2>(p, bg, grid, st) : pconv_r_launch<7, 128, 272, 2>(p, bg, grid, st);
    }
    if (bmt == 9) return big ? pconv_r_launch<9, 64, 320, 2>(p, bg, grid, st) : pconv_r_launch<9, 64, 272, 2>(p, bg, grid, st);
    return big ? pconv_r_launch<7, 64, 320, 2>(p, bg, grid, st) : pconv_r_launch<7, 64, 272, 3>(p, bg, grid, st);
}
static bool pconv_r_plan(int imgs, int H, int W, int N, int& bmt, int& bbn, RGeo& bg) {
    const int HW = H * W;
    bmt = bbn = 0;
    if (W < 2 || HW < 16 || N % 64 != 0 || imgs <= 0) return false;
    struct { int N; } p{N};
    const int cus = hw_cu_count();
    // tile choice: 16 MT pixels (MT = 7 or 9) x BN columns per workgroup -- the combination that keeps the CUs fullest, among those whose patch fits one
    // of the two reserved plane sizes (272 / 320 patch pixels); resident workgroups per CU by LDS (160 KB) and registers (MT = 7, BN = 64: three waves per SIMD)
    double best = -1.0;
    for (int bn = 128; bn >= 64; bn -= 64) {
        if (p.N % bn != 0) continue;
        for (int mt = 9; mt >= 7; mt -= 2) {
            const int gm = 16 * mt, gpi = (HW + gm - 1) / gm;
            int rows_max = 1;
            for (int t = 0; t < gpi; ++t) {
                const int a = t * gm, b = (a + gm < HW ? a + gm : HW) - 1;
                const int r = b / W - a / W + 1;
                if (r > rows_max) rows_max = r;
            }
            RGeo g{};
            g.gpi = gpi; g.ntiles = imgs * gpi; g.PW = W + 8;
            const int need = (rows_max + 2) * g.PW;
            if (need > 320) continue;
            g.patch_px = need <= 272 ? 272 : 320;
            g.nchunks = (need + 15) / 16;
            const int per_cu = (mt == 7 && bn == 64 && g.patch_px == 272) ? 3 : 2;
            const long wgs = (long)g.ntiles * (p.N / bn), slots = (long)cus * per_cu, rounds = (wgs + slots - 1) / slots;
            double eff = (double)wgs / (double)(rounds * slots) * ((double)HW / (double)(gpi * gm));
            eff *= (bn == 128 ? 1.0 : 0.93) * (mt == 9 ? 1.0 : 0.97);
            if (eff > best + 1e-9) { best = eff; bmt = mt; bbn = bn; bg = g; }
        }
    }
    if (bmt == 0) return false;
    bg.mgW = (unsigned)(((1ULL << 32) + (unsigned)W - 1) / (unsigned)W);
    bg.mgPW = (unsigned)(((1ULL << 32) + (unsigned)bg.PW - 1) / (unsigned)bg.PW);
    bg.mg_gpi = bg.gpi > 1 ? (unsigned)(((1ULL << 32) + (unsigned)bg.gpi - 1) / (unsigned)bg.gpi) : 0u;
    return true;
}

template <int NP>
static int pconv_q_dispatch(const PConvP& p_in, int maxM, hipStream_t st) {
    PConvP p = p_in;
    p.kmaj = g_kmaj;
    // buffer addressing needs 32-bit byte offsets inside a plane and the element counts of the planes (a: GH x GW x GC per image x images; b: N x K);
    // callers that know them set a_bytes / b_bytes, otherwise (0) the flat-address path runs
    p.buf = (g_qbuf && p.a_bytes > 0 && p.b_bytes > 0) ? 1 : 0;
    for (int ci = 0; ci < p.ncls; ++ci) {
        PClass& c = p.cls[ci];
        for (int t = 0; t < c.ntaps && t < 9; ++t) { c.khs[t] = (unsigned char)(c.tap[t] / p.KW); c.kws[t] = (unsigned char)(c.tap[t] % p.KW); }
        c.mg_ow = c.OWc > 1 ? (unsigned)(((1ULL << 32) + (unsigned)c.OWc - 1) / (unsigned)c.OWc) : 0u;
        c.mg_oh = c.OHc > 1 ? (unsigned)(((1ULL << 32) + (unsigned)c.OHc - 1) / (unsigned)c.OHc) : 0u;
    }
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    p.vec = (p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && (!p.bias || al16(p.bias)) && (p.ksplit <= 1 || al16(p.ws))) ? 1 : 0;
    if (p.ncls != 1 && (p.ksplit > 1 || !g_q_classes)) return -100;      // grid.z is the parity class OR the k slice
    int bmt = 0, bbn = 0;
    pconv_q_plan(maxM, p.N, p.ksplit > 1 ? p.ksplit : 1, &bmt, &bbn);
    if (bmt == 0) return -100;
    const dim3 grid((unsigned)((maxM + 32 * bmt - 1) / (32 * bmt)), (unsigned)((p.N + bbn - 1) / bbn), (unsigned)(p.ksplit > 1 ? p.ksplit : p.ncls));
    if (p.stat != nullptr && p.bsx != nullptr) return ha2g_set_error(-1, "conv2d_dgrad_planes_bnstats: the geometry is not served by the patch-resident kernel");
    if (p.stat != nullptr && !(p.fwd && p.vec && p.ncls == 1 && p.ksplit <= 1 && !p.bias && !p.act && p.beta == 0.f && p.stat_nblk == 2 * (int)grid.x))
        return ha2g_set_error(-1, "conv2d_fwd_planes: statistics buffer sized for %d row blocks, the q kernel writes %d", p.stat_nblk, 2 * (int)grid.x);
    if (bbn == 128) {
        if (bmt == 9) return pconv_q_launch<9, 128, NP>(p, grid, st);
        if (bmt == 8) return pconv_q_launch<8, 128, NP>(p, grid, st);
        return pconv_q_launch<7, 128, NP>(p, grid, st);
    }
    if (bmt == 9) return pconv_q_launch<9, 64, NP>(p, grid, st);
    if (bmt == 8) return pconv_q_launch<8, 64, NP>(p, grid, st);
    return pconv_q_launch<7, 64, NP>(p, grid, st);
}

template <int NP, int OUT>
static int pconv_dispatch(const PConvP& p, int maxM, hipStream_t st) {
    if constexpr (NP == 3) {
        // three pieces: 48 KB (128 x 128) / 60 KB (256 x 64) / 36 KB (128 x 64) per LDS stage -- the 128 x 64 tile keeps two workgroups per CU
        // default: the eight-wave ping-pong kernel where the column tile is 128 wide (layers 3 / 4: 168 vs 227 us at C = 256, 194 vs 199 at C = 128,
        // profiles/r04_bwd_matrix_bench_np3_v2.txt), the 128 x 64 tile (two workgroups per CU) for the 64-channel layer (179 vs 186 us)
        int t = g_tile3 ? g_tile3 : (p.N % 128 == 0 ? 4 : 3);
        if (g_tile3 == 5 || (g_tile3 == 0 && g_q_kernel)) {
            if constexpr (OUT == 0) {
                int rc = g_tile3 == 0 ? pconv_r_dispatch(p, p.cls[0].OHc > 0 && p.cls[0].OWc > 0 ? p.cls[0].M / (p.cls[0].OHc * p.cls[0].OWc) : 0, st) : -100;
                if (rc != -100) return rc;
                if (p.rsd != nullptr) return ha2g_set_error(-1, "conv2d_dgrad_planes_resid: the geometry is not served by the patch-resident kernel");
                rc = pconv_q_dispatch<NP>(p, maxM, st);
                if (rc != -100) return rc;
            }
            t = g_tile3 == 5 ? 4 : (p.N % 128 == 0 ? 4 : 3);
        }
        if (t == 4) {
            if (p.N % 128 == 0) return pconv_pp_launch<128, NP, OUT>(p, dim3(ceil_div(maxM, 256), p.N / 128, p.ncls), st);
            return pconv_pp_launch<64, NP, OUT>(p, dim3(ceil_div(maxM, 256), ceil_div(p.N, 64), p.ncls), st);
        }
        if (t == 1 && p.N % 128 != 0) t = 3;
        if (t == 1) return pconv_launch<128, 128, 2, 2, NP, OUT>(p, dim3(ceil_div(maxM, 128), p.N / 128, p.ncls), st);
        if (t == 2) return pconv_launch<256, 64, 4, 1, NP, OUT>(p, dim3(ceil_div(maxM, 256), ceil_div(p.N, 64), p.ncls), st);
        return pconv_launch<128, 64, 2, 2, NP, OUT>(p, dim3(ceil_div(maxM, 128), ceil_div(p.N, 64), p.ncls), st);
    }
    if (g_waves == 8) {
        if (p.N % 128 == 0) return pconv_launch<256, 128, 4, 2, NP, OUT>(p, dim3(ceil_div(maxM, 256), p.N / 128, p.ncls), st);
        return pconv_launch<512, 64, 8, 1, NP, OUT>(p, dim3(ceil_div(maxM, 512), ceil_div(p.N, 64), p.ncls), st);
    }
    if (p.N % 128 == 0) return pconv_launch<128, 128, 2, 2, NP, OUT>(p, dim3(ceil_div(maxM, 128), p.N / 128, p.ncls), st);
    return pconv_launch<256, 64, 4, 1, NP, OUT>(p, dim3(ceil_div(maxM, 256), ceil_div(p.N, 64), p.ncls), st);
}


// Dense product on the quantisation-free plane kernel: C [M][N] (fp32, row stride ldc) = act(A B^T + bias) + beta C with A = piece planes
// [M][lda], B = piece planes [N][ldb] (three pieces each, k contiguous, zero-padded to whole 32-wide k tiles: lda, ldb >= round_up(K, 32)) -- a 1x1
// "convolution" over M pixels with lda channels.  ksplit > 1: k slices over grid.z write raw partial slabs ws [ksplit][M][N] (the caller reduces).
int plane_gemm_plan(int M, int N, int ksplit, int* mt, int* bn) { return pconv_q_plan(M, N, ksplit, mt, bn) > 0.0 ? 0 : -100; }
int plane_gemm_launch(const void* a, long a_ps, long lda, const void* b, long b_ps, long ldb, int M, int N, int K, float* C, long ldc, float beta,
                      const float* bias, int act, float* ws, int ksplit, int* tickets, hipStream_t st) {
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)a, a_ps}; p.b = PlaneSet{(const unsigned short*)b, b_ps};
    p.C = C; p.ldc = ldc; p.beta = beta; p.fwd = 1; p.bias = bias; p.act = act;
    p.N = N; p.K = (int)ldb;                                     // row stride of the B planes
    const int kpad = (K + 31) / 32 * 32;
    p.GH = 1; p.GW = M; p.GC = (int)lda; p.OH = 1; p.OW = M; p.KH = 1; p.KW = 1; p.pad = 0; p.stride = 1;
    p.dbg = g_pdbg;
    PClass c{};
    c.OHc = 1; c.OWc = M; c.M = M; c.ntaps = 1; c.tap[0] = 0; c.doff[0] = 0;
    p.ncls = 1; p.cls[0] = c;
    // the kernel walks GC / 32 k tiles per tap: only the first kpad / 32 carry data -- hand it a class whose channel count is kpad
    const int nkt = kpad / 32;
    p.ksplit = ksplit > 1 ? ksplit : 1;
    p.kt_per = (nkt + p.ksplit - 1) / p.ksplit;
    p.ws = ws;
    p.tickets = p.ksplit > 1 ? tickets : nullptr;                // in-kernel reduction of the k slices (the caller checked the tile count against the ticket buffer)
    if (M == 0 || N == 0) return 0;
    // GC is the row stride AND (>> 5) the k-tile count of the kernel: when lda > kpad the surplus tiles would multiply zeros by zeros; forbid it
    if (lda != kpad) return ha2g_set_error(-1, "plane_gemm: lda %ld must equal K rounded up to 32 (%d)", lda, kpad);
    set_plane_bytes(p, (long)M * lda, (long)N * ldb);
    if (int rc = pconv_q_dispatch<3>(p, M, st)) return rc;
    HA2G_CHECK_LAUNCH("plane_gemm");
    return 0;
}

extern "C" {

// x [rows][ldx] fp32 -> np = 3 piece planes for the plane GEMM: transpose = 0: [rows][ldp], `cols` valid columns, zeros behind; transpose = 1:
// [cols][ldp] holding x^T, `rows` valid columns.  ldp % 32 == 0; piece q at planes + q * ps elements.
int ha2g_f32_to_planes_2d_np(const float* x, long ldx, long rows, int cols, void* planes, long ps, long ldp, int np, int transpose, void* stream) {
    HA2G_REQUIRE(np == 3, "f32_to_planes_2d: np = %d (3)", np);
    HA2G_REQUIRE(ldp % 32 == 0 && ldp >= (transpose ? rows : cols), "f32_to_planes_2d: ldp %ld", ldp);
    if (rows == 0 || cols == 0) return 0;
    if (!transpose) {
        const long total = rows * (ldp / 4);
        const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        hipLaunchKernelGGL(f32_to_planes_pad_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned short*)planes, ps, ldp, rows, cols);
    } else {
        hipLaunchKernelGGL(f32_to_planes_t_kernel<3>, dim3((unsigned)(ldp / 32), (unsigned)((cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           (unsigned short*)planes, ps, ldp, rows, cols);
    }
    HA2G_CHECK_LAUNCH("f32_to_planes_2d");
    return 0;
}

// n <= 16 two-dimensional splits in one launch (HOST arrays of device pointers / sizes); see f32_to_planes_2d_multi_kernel.  Every job writes a window of
// the same plane set: piece q of job i at planes[i] + q * ps, row stride ldp.  transpose = 0: window = rows[i] x wcols[i]; 1: cols[i] x wcols[i].
int ha2g_f32_to_planes_2d_multi_np(const void* const* x, const long* ldx, const int* rows, const int* cols, void* const* planes, const int* wcols, long ps, long ldp,
                                   int n, int np, int transpose, void* stream) {
    HA2G_REQUIRE(np == 3, "f32_to_planes_2d_multi: np = %d (3)", np);
    HA2G_REQUIRE(n >= 0 && n <= 16, "f32_to_planes_2d_multi: %d jobs (max 16)", n);
    if (n == 0) return 0;
    Split2dJobs jb{};
    jb.ps = ps; jb.ldp = ldp; jb.tr = transpose ? 1 : 0;
    long most_r = 0; int most_c = 0;
    for (int i = 0; i < n; ++i) {
        HA2G_REQUIRE(wcols[i] % 4 == 0 && wcols[i] <= ldp && wcols[i] >= (transpose ? rows[i] : cols[i]), "f32_to_planes_2d_multi: window of %d columns", wcols[i]);
        HA2G_REQUIRE(((uintptr_t)planes[i] & 7) == 0 && ldp % 4 == 0 && ps % 4 == 0, "f32_to_planes_2d_multi: plane windows must be 8-byte aligned");
        jb.x[i] = (const float*)x[i]; jb.pl[i] = (unsigned short*)planes[i]; jb.ldx[i] = ldx[i]; jb.rows[i] = rows[i]; jb.cols[i] = cols[i]; jb.wcols[i] = wcols[i];
        const long work = transpose ? wcols[i] : (long)rows[i] * (wcols[i] / 4);
        if (work > most_r) most_r = work;
        if (cols[i] > most_c) most_c = cols[i];
    }
    dim3 grid;
    if (transpose) grid = dim3((unsigned)((most_r + 31) / 32), (unsigned)((most_c + 31) / 32), (unsigned)n);
    else { const long b = (most_r + 255) / 256; grid = dim3((unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b)), 1, (unsigned)n); }
    hipLaunchKernelGGL(f32_to_planes_2d_multi_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, jb);
    HA2G_CHECK_LAUNCH("f32_to_planes_2d_multi");
    return 0;
}

void ha2g_conv_planes_enable(int on) { g_planes = on; }
void ha2g_conv_planes_debug(int bits) { g_pdbg = bits; }
void ha2g_conv_planes_ring(int depth) { g_ring = depth; }
void ha2g_conv_planes_waves(int n) { g_waves = n == 8 ? 8 : 4; }
void ha2g_conv_planes_bufaddr(int on) { g_qbuf = on ? 1 : 0; }
void ha2g_conv_planes_korder(int channel_major) { g_kmaj = channel_major ? 1 : 0; }
void ha2g_conv_planes_tile3(int t) {
    g_q_classes = 1;
    g_r_kernel = 1;
    if (t == 8) { g_r_kernel = 0; t = 0; }                        // the q kernel for the 3x3 / stride-1 convolutions too (round 4's default; A/B)
    if (t == 7) { g_q_classes = 0; g_q_kernel = 1; g_tile3 = 0; }
    else if (t == 6) { g_q_kernel = 0; g_tile3 = 0; }
    else { g_q_kernel = 1; g_tile3 = (t >= 0 && t <= 5) ? t : 0; }
}

// fp32 -> np bf16 piece planes of the same shape (piece q at planes + q * ps elements); n % 4 == 0, 16-byte aligned, ps % 8 == 0
int ha2g_f32_to_planes_np(const float* x, void* planes, long ps, int np, long n, void* stream) {
    HA2G_REQUIRE(n % 4 == 0, "f32_to_planes: n %% 4");
    HA2G_REQUIRE(np == 2 || np == 3, "f32_to_planes: np = %d (2 or 3)", np);
    if (n == 0) return 0;
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    if (np == 3) hipLaunchKernelGGL(f32_to_planes_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)planes, ps, n4);
    else hipLaunchKernelGGL(f32_to_planes_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)planes, ps, n4);
    HA2G_CHECK_LAUNCH("f32_to_planes");
    return 0;
}
// n <= 48 tensors in one launch; x / planes / ps / numel are HOST arrays; numel[i] % 4 == 0
int ha2g_f32_to_planes_multi_np(const void* const* x, void* const* planes, const long* ps, const long* numel, int n, int np, void* stream) {
    HA2G_REQUIRE(n >= 0 && n <= 48, "f32_to_planes_multi: %d tensors (max 48)", n);
    HA2G_REQUIRE(np == 2 || np == 3, "f32_to_planes_multi: np = %d (2 or 3)", np);
    if (n == 0) return 0;
    SplitBatch b{};
    long most = 0;
    for (int i = 0; i < n; ++i) {
        HA2G_REQUIRE(numel[i] % 4 == 0, "f32_to_planes_multi: numel %% 4");
        b.x[i] = (const float*)x[i]; b.pl[i] = (unsigned short*)planes[i]; b.ps[i] = ps[i]; b.n4[i] = numel[i] / 4;
        if (b.n4[i] > most) most = b.n4[i];
    }
    const int gx = (int)((most + 255) / 256 > 512 ? 512 : (most + 255) / 256);
    if (np == 3) hipLaunchKernelGGL(f32_to_planes_multi_kernel<3>, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(f32_to_planes_multi_kernel<2>, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, b);
    HA2G_CHECK_LAUNCH("f32_to_planes_multi");
    return 0;
}
// the two-plane form of round 3: (hi, lo) = pieces 0, 1
int ha2g_f32_to_planes(const float* x, void* hi, void* lo, long n, void* stream) {
    return ha2g_f32_to_planes_np(x, hi, (const unsigned short*)lo - (const unsigned short*)hi, 2, n, stream);
}

// w [Cout][KH][KW][Cin] fp32 -> planes of [Cin][KH][KW][Cout] (what ha2g_conv2d_weight_ohwi_to_ihwo_f32 + a split would give)
int ha2g_conv2d_weight_ihwo_planes(const float* w, void* wt_hi, void* wt_lo, int Cout, int KH, int KW, int Cin, void* stream) {
    HA2G_REQUIRE(Cout % 2 == 0, "weight_ihwo_planes: Cout %% 2");
    const long total = (long)Cout * KH * KW * Cin / 2;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(weight_ihwo_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wt_hi, (unsigned short*)wt_lo,
                       Cout, KH * KW, Cin);
    HA2G_CHECK_LAUNCH("weight_ihwo_planes");
    return 0;
}

// n <= 48 weights in one launch; the arrays are HOST arrays (of device pointers / of sizes); piece q of tensor i at wt[i] + q * ps[i] elements
int ha2g_conv2d_weight_ihwo_planes_multi_np(const void* const* w, void* const* wt, const long* ps, const int* cout, const int* kk, const int* cin,
                                            int n, int np, void* stream) {
    HA2G_REQUIRE(n >= 0 && n <= WPB_MAX, "weight_ihwo_planes_multi: %d tensors (max %d)", n, WPB_MAX);
    HA2G_REQUIRE(np == 2 || np == 3, "weight_ihwo_planes_multi: np = %d (2 or 3)", np);
    if (n == 0) return 0;
    WPlanesBatch b{};
    long most = 0;
    for (int i = 0; i < n; ++i) {
        HA2G_REQUIRE(cout[i] % 2 == 0, "weight_ihwo_planes_multi: Cout %% 2");
        b.w[i] = (const float*)w[i]; b.pl[i] = (unsigned short*)wt[i]; b.ps[i] = ps[i];
        b.cout[i] = cout[i]; b.kk[i] = kk[i]; b.cin[i] = cin[i];
        const long tot = (long)cout[i] * kk[i] * cin[i] / 2;
        if (tot > most) most = tot;
    }
    const int gx = (int)((most + 255) / 256 > 512 ? 512 : (most + 255) / 256);
    if (np == 3) hipLaunchKernelGGL(weight_ihwo_planes_multi_kernel<3>, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(weight_ihwo_planes_multi_kernel<2>, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, b);
    HA2G_CHECK_LAUNCH("weight_ihwo_planes_multi");
    return 0;
}
int ha2g_conv2d_weight_ihwo_planes_multi(const void* const* w, void* const* wt_hi, void* const* wt_lo, const int* cout, const int* kk, const int* cin,
                                         int n, void* stream) {
    HA2G_REQUIRE(n >= 0 && n <= WPB_MAX, "weight_ihwo_planes_multi: %d tensors (max %d)", n, WPB_MAX);
    long ps[WPB_MAX];
    for (int i = 0; i < n; ++i) ps[i] = (const unsigned short*)wt_lo[i] - (const unsigned short*)wt_hi[i];
    return ha2g_conv2d_weight_ihwo_planes_multi_np(w, wt_hi, ps, cout, kk, cin, n, 2, stream);
}

// 1 when ha2g_conv2d_dgrad_planes_f32 serves this geometry in the current arithmetic mode (split-bf16 data gradients on, planes enabled):
// 3x3 / pad 1 or 1x1 / pad 0, stride 1 or 2, Cout % 32 == 0 (k tiles of 32 channels), Cin >= 32 and a multiple of 32
int ha2g_conv2d_dgrad_planes_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const bool geom = (KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0 && stride == 2);
    return g_planes && gemm_split_dgrad_enabled() && geom && (stride == 1 || stride == 2) && Cout % 32 == 0 && Cin % 32 == 0 && Cin >= 32 &&
           !(stride == 1 && Cin < 64);                              // the 32 -> 32 stride-1 layer has its direct LDS-patch kernel
}

// dx [N,H,W,Cin] = beta * dx + conv_transpose(dy, w) from the bf16 planes of dy [N,OH,OW,Cout] and of wt [Cin][KH][KW][Cout]
// (ha2g_conv2d_weight_ihwo_planes).  Bit-identical to ha2g_conv2d_dgrad_f32 in the default arithmetic mode on the fp32 tensors the planes
// were split from.  Stride 2: pixels that no tap reaches (1x1 kernel: three of the four parity classes) are NOT written: with beta = 0 the
// caller zero-fills dx first (ha2g_amd.wav_engine does).
// np = 2 or 3 piece planes: dy pieces at dy + q * dy_ps, transposed-weight pieces at wt + q * wt_ps (elements)
int ha2g_conv2d_dgrad_planes_np_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W,
                                    int Cin, int Cout, int KH, int KW, int stride, int pad, float beta, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_dgrad_planes_supported(Cin, Cout, KH, KW, stride, pad), "conv2d_dgrad_planes: unsupported geometry / mode");
    HA2G_REQUIRE(np == 2 || np == 3, "conv2d_dgrad_planes: np = %d (2 or 3)", np);
    const int OHd = (H + 2 * pad - KH) / stride + 1, OWd = (W + 2 * pad - KW) / stride + 1;      // the dy grid
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)dy, dy_ps};
    p.b = PlaneSet{(const unsigned short*)wt, wt_ps};
    p.C = dx; p.ldc = Cin; p.beta = beta;
    p.N = Cin; p.K = KH * KW * Cout;
    p.GH = OHd; p.GW = OWd; p.GC = Cout; p.OH = H; p.OW = W; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    if ((long)N * H * W == 0) return 0;
    const int maxM = dgrad_classes(p, N, H, W, KH, KW, stride, pad, OWd);
    set_plane_bytes(p, (long)N * OHd * OWd * Cout, (long)Cin * KH * KW * Cout);
    if (int rc = (np == 3 ? pconv_dispatch<3, 0>(p, maxM, (hipStream_t)stream) : pconv_dispatch<2, 0>(p, maxM, (hipStream_t)stream))) return rc;
    HA2G_CHECK_LAUNCH("conv2d_dgrad_planes");
    return 0;
}
// ... + (decision bit ? resid : 0) in the epilogue instead of beta = 1 onto a materialised dres = dout * (out > 0) (round 6): resid [N,H,W,Cin] fp32,
// resid_bits = ha2g_se_bn_scale_add_relu_mask_np_f32's words over the same tensor.  Patch-resident kernel only: ha2g_conv2d_dgrad_planes_resid_supported.
int ha2g_conv2d_dgrad_planes_resid_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!ha2g_conv2d_dgrad_planes_supported(Cin, Cout, KH, KW, stride, pad)) return 0;
    if (!(g_tile3 == 0 && g_q_kernel && g_r_kernel && gemm_bwd_pieces() == 3)) return 0;
    if (!(KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cout % 32 == 0 && Cin % 64 == 0)) return 0;
    int bmt = 0, bbn = 0; RGeo g{};
    return pconv_r_plan(N, H, W, Cin, bmt, bbn, g) ? 1 : 0;
}
int ha2g_conv2d_dgrad_planes_np_resid_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad, const float* resid, const void* resid_bits, void* stream) {
    HA2G_REQUIRE(np == 3, "conv2d_dgrad_planes_resid: np = %d (3)", np);
    HA2G_REQUIRE(ha2g_conv2d_dgrad_planes_resid_supported(N, H, W, Cin, Cout, KH, KW, stride, pad), "conv2d_dgrad_planes_resid: unsupported geometry / mode");
    HA2G_REQUIRE(resid != nullptr && resid_bits != nullptr && dx != resid && (((uintptr_t)resid | (uintptr_t)dx) & 15) == 0,
                 "conv2d_dgrad_planes_resid: null / misaligned residual, or dx aliases it");
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)dy, dy_ps};
    p.b = PlaneSet{(const unsigned short*)wt, wt_ps};
    p.C = dx; p.ldc = Cin; p.beta = 0.f; p.rsd = resid; p.rsd_bits = (const unsigned*)resid_bits;
    p.N = Cin; p.K = KH * KW * Cout;
    p.GH = H; p.GW = W; p.GC = Cout; p.OH = H; p.OW = W; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    if ((long)N * H * W == 0) return 0;
    const int maxM = dgrad_classes(p, N, H, W, KH, KW, stride, pad, W);
    set_plane_bytes(p, (long)N * H * W * Cout, (long)Cin * KH * KW * Cout);
    if (int rc = pconv_dispatch<3, 0>(p, maxM, (hipStream_t)stream)) return rc;
    HA2G_CHECK_LAUNCH("conv2d_dgrad_planes_resid");
    return 0;
}
int ha2g_conv2d_dgrad_planes_f32(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, float beta, void* stream) {
    return ha2g_conv2d_dgrad_planes_np_f32(dy_hi, (const unsigned short*)dy_lo - (const unsigned short*)dy_hi, wt_hi,
                                           (const unsigned short*)wt_lo - (const unsigned short*)wt_hi, 2, dx, N, H, W, Cin, Cout, KH, KW, stride, pad,
                                           beta, stream);
}

// Round 6: the data gradient of a 3x3 / stride-1 / pad-1 convolution whose OUTPUT is the dy of a BatchNorm backward (conv2's data gradient feeds bn1,
// model/ResNetBlocks.py:24-29 under autograd) leaves that backward's tile sums behind: stat_part [2][Cin][stat_nblk] doubles = per row tile the sums of
// dx and of dx * xhat, xhat = (x_bn - mean) * invstd with x_bn [N,H,W,Cin] the BatchNorm's input.  ha2g_bn_bwd_planes_np_partials_f32 finishes them:
// the column pass over dx and x_bn (col_partial_kernel<1>) is not run.  stat_nblk = ha2g_conv2d_dgrad_planes_stat_blocks(...) > 0: the geometry is
// served by the patch-resident kernel in the current configuration (else the caller keeps the column pass).
int ha2g_conv2d_dgrad_planes_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!ha2g_conv2d_dgrad_planes_supported(Cin, Cout, KH, KW, stride, pad)) return 0;
    if (!(g_tile3 == 0 && g_q_kernel && g_r_kernel && gemm_bwd_pieces() == 3)) return 0;
    if (!(KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cout % 32 == 0 && Cin % 64 == 0)) return 0;
    int bmt = 0, bbn = 0; RGeo g{};
    return pconv_r_plan(N, H, W, Cin, bmt, bbn, g) ? g.ntiles : 0;
}
int ha2g_conv2d_dgrad_planes_np_bnstats_f32(const void* dy, long dy_ps, const void* wt, long wt_ps, int np, float* dx, int N, int H, int W, int Cin, int Cout,
                                            int KH, int KW, int stride, int pad, const float* x_bn, const float* mean, const float* invstd, void* stat_part,
                                            int stat_nblk, void* stream) {
    HA2G_REQUIRE(np == 3, "conv2d_dgrad_planes_bnstats: np = %d (3)", np);
    HA2G_REQUIRE(x_bn && mean && invstd && stat_part, "conv2d_dgrad_planes_bnstats: null operand");
    HA2G_REQUIRE(stat_nblk > 0 && stat_nblk == ha2g_conv2d_dgrad_planes_stat_blocks(N, H, W, Cin, Cout, KH, KW, stride, pad),
                 "conv2d_dgrad_planes_bnstats: stat_nblk = %d is not what ha2g_conv2d_dgrad_planes_stat_blocks reports for this geometry", stat_nblk);
    HA2G_REQUIRE((((uintptr_t)x_bn | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)dx) & 15) == 0, "conv2d_dgrad_planes_bnstats: 16-byte aligned operands");
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)dy, dy_ps};
    p.b = PlaneSet{(const unsigned short*)wt, wt_ps};
    p.C = dx; p.ldc = Cin; p.beta = 0.f;
    p.N = Cin; p.K = KH * KW * Cout;
    p.GH = H; p.GW = W; p.GC = Cout; p.OH = H; p.OW = W; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    p.stat = (double*)stat_part; p.stat_nblk = stat_nblk; p.bsx = x_bn; p.bsmean = mean; p.bsinv = invstd;
    if ((long)N * H * W == 0) return 0;
    const int maxM = dgrad_classes(p, N, H, W, KH, KW, stride, pad, W);
    set_plane_bytes(p, (long)N * H * W * Cout, (long)Cin * KH * KW * Cout);
    if (int rc = pconv_dispatch<3, 0>(p, maxM, (hipStream_t)stream)) return rc;
    HA2G_CHECK_LAUNCH("conv2d_dgrad_planes_bnstats");
    return 0;
}

// ---- FORWARD convolution on three-piece planes (round 4): y [N,OH,OW,Cout] (fp32) = [relu](conv(x, w)) from the piece planes of x [N,H,W,Cin]
//      (written by x's producer: ha2g_bn_apply_planes_np_f32 / ha2g_se_scale_add_relu_planes_np_f32) and of w [Cout][KH][KW][Cin] (OHWI,
//      ha2g_f32_to_planes_multi_np).  Six bf16 MFMAs per product on all 24 mantissa bits of both operands = the accuracy of the fp32 MFMA
//      chain (tests/test_gpu_np3.py), at 6 x 8 instead of 8 x 16 matrix-pipe passes.  3x3 / pad 1 or 1x1 / pad 0, stride 1 or 2, Cin % 32 == 0,
//      Cout % 64 == 0.  Replaces nn.Conv2d forward of the SE-ResNet trunk layers 2-4 (model/ResNetBlocks.py:24-29, ResNetSE34V2.py:96-116). ----
int ha2g_conv2d_fwd_planes_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const bool geom = (KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0);
    return g_planes && geom && (stride == 1 || stride == 2) && Cin % 32 == 0 && Cin >= 32 && Cout % 64 == 0;
}
// Row tiles per channel of the BatchNorm statistics this forward convolution can leave behind (ha2g_conv2d_fwd_planes_np_stats_f32): the tile
// count of the patch-resident kernel when it serves the geometry (3x3 / stride 1 / pad 1) in the current configuration, else 0 (the caller runs
// ha2g_bn_stats_f32 on the output).  Host arithmetic only.
int ha2g_conv2d_fwd_planes_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!ha2g_conv2d_fwd_planes_supported(Cin, Cout, KH, KW, stride, pad)) return 0;
    if (!(g_tile3 == 0 && g_q_kernel)) return 0;
    int bmt = 0, bbn = 0; RGeo g{};
    if (g_r_kernel && KH == 3 && KW == 3 && stride == 1 && pad == 1 && pconv_r_plan(N, H, W, Cout, bmt, bbn, g)) return g.ntiles;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;     // the q kernel: two row groups per workgroup
    const long M = (long)N * OH * OW;
    if (M <= 0 || M > 0x7fffffffL) return 0;
    pconv_q_plan((int)M, Cout, 1, &bmt, &bbn);
    return bmt ? 2 * (int)((M + 32 * bmt - 1) / (32 * bmt)) : 0;
}
// > 0: the statistics blocks are tiles INSIDE one image (the patch-resident kernel), that many per image in image order -- per-image column sums
// (the SE squeeze, ha2g_bn_pool_from_partials_f32) are sums of consecutive blocks; 0: they are not (the q kernel's row groups straddle images) or none
int ha2g_conv2d_fwd_planes_stat_tiles_per_image(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (!ha2g_conv2d_fwd_planes_supported(Cin, Cout, KH, KW, stride, pad) || !(g_tile3 == 0 && g_q_kernel)) return 0;
    int bmt = 0, bbn = 0; RGeo g{};
    return (g_r_kernel && KH == 3 && KW == 3 && stride == 1 && pad == 1 && pconv_r_plan(N, H, W, Cout, bmt, bbn, g)) ? g.gpi : 0;
}
static int conv2d_fwd_planes_impl(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride, int pad, int relu, double* stat, int stat_nblk, void* stream);
int ha2g_conv2d_fwd_planes_np_f32(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride, int pad, int relu, void* stream) {
    return conv2d_fwd_planes_impl(x, x_ps, w, w_ps, np, y, N, H, W, Cin, Cout, KH, KW, stride, pad, relu, nullptr, 0, stream);
}
// ... and the statistics of the BatchNorm that follows (conv -> [ReLU] -> BatchNorm, ResNetBlocks.py:24-29,81-83) from the same launch:
// stat_part [2][Cout][stat_nblk] doubles receives per row tile the sum and the sum of squares of the STORED output (after the ReLU);
// stat_nblk = ha2g_conv2d_fwd_planes_stat_blocks(...) > 0.  ha2g_bn_stats_finalize_f32 turns them into mean / invstd / running statistics.
int ha2g_conv2d_fwd_planes_np_stats_f32(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout,
                                        int KH, int KW, int stride, int pad, int relu, void* stat_part, int stat_nblk, void* stream) {
    HA2G_REQUIRE(stat_part != nullptr && stat_nblk > 0 && stat_nblk == ha2g_conv2d_fwd_planes_stat_blocks(N, H, W, Cin, Cout, KH, KW, stride, pad),
                 "conv2d_fwd_planes_stats: stat_nblk = %d is not what ha2g_conv2d_fwd_planes_stat_blocks reports for this geometry", stat_nblk);
    return conv2d_fwd_planes_impl(x, x_ps, w, w_ps, np, y, N, H, W, Cin, Cout, KH, KW, stride, pad, relu, (double*)stat_part, stat_nblk, stream);
}
static int conv2d_fwd_planes_impl(const void* x, long x_ps, const void* w, long w_ps, int np, float* y, int N, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride, int pad, int relu, double* stat, int stat_nblk, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_fwd_planes_supported(Cin, Cout, KH, KW, stride, pad), "conv2d_fwd_planes: unsupported geometry");
    HA2G_REQUIRE(np == 3, "conv2d_fwd_planes: np = %d (the forward runs on three pieces only: fp32-class)", np);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)x, x_ps}; p.b = PlaneSet{(const unsigned short*)w, w_ps};
    p.C = y; p.ldc = Cout; p.beta = 0.f; p.relu = relu; p.fwd = 1;
    p.N = Cout; p.K = KH * KW * Cin;
    p.GH = H; p.GW = W; p.GC = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    PClass c{};
    c.OHc = OH; c.OWc = OW; c.M = N * OH * OW;
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw) { c.tap[c.ntaps] = kh * KW + kw; c.doff[c.ntaps] = kh * W + kw; ++c.ntaps; }
    p.ncls = 1; p.cls[0] = c;
    p.stat = stat; p.stat_nblk = stat_nblk;
    if (c.M == 0) return 0;
    set_plane_bytes(p, (long)N * H * W * Cin, (long)Cout * KH * KW * Cin);
    if (int rc = pconv_dispatch<3, 0>(p, c.M, (hipStream_t)stream)) return rc;
    HA2G_CHECK_LAUNCH("conv2d_fwd_planes");
    return 0;
}

// ---- bf16-storage mode (BASELINE config 5, `bench.py --bf16`): activations and the dY stream of the audio tower live in HBM as bf16; the
//      convolutions read them as they are (one plane, one MFMA per product, fp32 accumulate) and write bf16.  3x3 / pad 1, 1x1 / pad 0,
//      stride 1 or 2, channel counts multiples of 32.  Replaces nn.Conv2d and its autograd backward (model/ResNetBlocks.py:24-29,
//      model/ResNetSE34V2.py:96-116) in that mode. ----
// Tuning knob: the persistent weight-gradient kernels of the side stream (plane kernel, direct 32-channel kernel) launch at most one (two)
// workgroup(s) per compute unit on `n` units instead of all of them, so that the main queue's bandwidth-bound passes find free units.
void ha2g_side_cus(int n) { g_side_cus = n < 8 ? 8 : (n > 256 ? 256 : n); }

int ha2g_conv2d_b16_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const bool geom = (KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0);
    return geom && (stride == 1 || stride == 2) && Cin % 32 == 0 && Cout % 32 == 0;
}
// y [N,OH,OW,Cout] (bf16) = [relu](conv(x [N,H,W,Cin] (bf16), w [Cout][KH][KW][Cin] (bf16)))
int ha2g_conv2d_fwd_b16(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                        void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_b16_supported(Cin, Cout, KH, KW, stride, pad), "conv2d_fwd_b16: unsupported geometry");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)x, 0}; p.b = PlaneSet{(const unsigned short*)w, 0};
    p.C = (float*)y; p.ldc = Cout; p.beta = 0.f; p.relu = relu; p.fwd = 1;
    p.N = Cout; p.K = KH * KW * Cin;
    p.GH = H; p.GW = W; p.GC = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    PClass c{};
    c.OHc = OH; c.OWc = OW; c.M = N * OH * OW;
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw) { c.tap[c.ntaps] = kh * KW + kw; c.doff[c.ntaps] = kh * W + kw; ++c.ntaps; }
    p.ncls = 1; p.cls[0] = c;
    if (c.M == 0) return 0;
    if (int rc = pconv_dispatch<1, 1>(p, c.M, (hipStream_t)stream)) return rc;
    HA2G_CHECK_LAUNCH("conv2d_fwd_b16");
    return 0;
}
// dx [N,H,W,Cin] (bf16) = beta * dx + conv_transpose(dy [N,OH,OW,Cout] (bf16), wt [Cin][KH][KW][Cout] (bf16)); stride 2 with a 1x1 kernel and
// beta = 0: zero-fill dx first (three of four pixels are not reached)
int ha2g_conv2d_dgrad_b16(const void* dy, const void* wt, void* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          float beta, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_b16_supported(Cin, Cout, KH, KW, stride, pad), "conv2d_dgrad_b16: unsupported geometry");
    const int OHd = (H + 2 * pad - KH) / stride + 1, OWd = (W + 2 * pad - KW) / stride + 1;
    PConvP p{};
    p.a = PlaneSet{(const unsigned short*)dy, 0}; p.b = PlaneSet{(const unsigned short*)wt, 0};
    p.C = (float*)dx; p.ldc = Cin; p.beta = beta;
    p.N = Cin; p.K = KH * KW * Cout;
    p.GH = OHd; p.GW = OWd; p.GC = Cout; p.OH = H; p.OW = W; p.KH = KH; p.KW = KW; p.pad = pad; p.stride = stride;
    p.dbg = g_pdbg;
    if ((long)N * H * W == 0) return 0;
    const int maxM = dgrad_classes(p, N, H, W, KH, KW, stride, pad, OWd);
    if (int rc = pconv_dispatch<1, 1>(p, maxM, (hipStream_t)stream)) return rc;
    HA2G_CHECK_LAUNCH("conv2d_dgrad_b16");
    return 0;
}

}  // extern "C"
