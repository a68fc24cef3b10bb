// fp32 MFMA GEMM family for gfx950 (MI355X): dense GEMM (NT / NN / TN) and NHWC implicit-GEMM
// convolution (forward, data-gradient, weight-gradient) share one LDS-tiled kernel.
//
//   C[M,N] = act(alpha * A[M,K] * B[K,N] + beta * C + bias[n])
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles / SIMD).  A block is 4 waves (WMxWN);
// each wave owns an (32*MI) x (32*NI) output tile; K advances 16 per LDS stage.  LDS tiles are k-major
// (As[k][m], Bs[k][n]) so an MFMA operand read is 32 consecutive floats per half-wave (conflict free).
// Global->LDS staging goes through registers (double-buffered LDS, one barrier per K-step).
//
// Operand addressing modes:
//   A: KC  a(m,k) = A[m*lda + k]            (activations, row-major)
//      MC  a(m,k) = A[k*lda + m]            (transposed: dY^T in weight gradients)
//      IM  a(m,k) = gather from an NHWC tensor; m = (img, oy, ox), k = (kh, kw, c)   (conv fwd / dgrad)
//   B: NC  b(k,n) = B[k*ldb + n]
//      KC  b(k,n) = B[n*ldb + k]            (torch Linear weight [N,K]; conv weight [co][(kh,kw,ci)])
//      IM  b(k,n) = gather from NHWC x; k = output pixel, n = (kh, kw, ci)           (conv wgrad)
// Split-K (grid.z): partial tiles go to a caller-provided workspace and ha2g reduce kernel applies
// the epilogue -- deterministic (no float atomics).
#include "common.h"

namespace {

enum { A_KC = 0, A_MC = 1, A_IM = 2 };
enum { B_NC = 0, B_KC = 1, B_IM = 2 };

struct ConvGeom {
    int GH, GW, GC;   // gathered tensor: height, width, channels (NHWC)
    int OH, OW;       // output pixel grid the GEMM rows (A_IM) / K index (B_IM) run over
    int KH, KW, stride, pad;
    int transposed;   // A_IM only: 1 = data-gradient gather (oy + pad - kh must be divisible by stride)
};

struct GemmP {
    int M, N, K;
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    float alpha, beta;
    const float* bias;
    int act;          // 0 none, 1 relu, 2 leaky-relu(0.01), 3 sigmoid
    int splits, kchunk;
    float* ws;        // [splits][M][N] when splits > 1
    ConvGeom g;
    // A_MC (weight-gradient shape, A = dY stored [K][M]) only: csum[m] = csum_beta * csum[m] + sum_k A[k][m] -- the bias gradient of the
    // layer from the dY tiles this launch stages anyway.  Partials of a split-K launch go to ws + splits*M*N as [splits][M].
    float* csum; float csum_beta;
    // Grouped launch (gemm_kernel only): `groups` independent problems of one shape, group = blockIdx.z / splits.  Per-group base pointers
    // (stacked operands: base + g * stride, or separate tensors); split-K slabs [group][split][M][N], then the bias partials [group][split][M].
    int groups;
    const float* Ag[8]; const float* Bg[8]; float* Cg[8]; const float* biasg[8]; float* csumg[8];
    // in-kernel split-K reduction (common.h): one arrival ticket per (group, output tile); null = the caller launches splitk_reduce_kernel
    int* tickets;
};


__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : 0.01f * v;
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    return v;
}

// The last-arriving k slice of an output tile (splitk_last_arriver) adds the `splits` raw slabs of rows [m0, m0 + BM) x columns [n0, n0 + BN) in
// slice order -- double accumulation, then alpha / bias / beta / activation: element for element the arithmetic of splitk_reduce_kernel --
// and, on the weight-gradient shape, the bias-gradient partials of its rows (the by == 0 tile).
template <int BM, int BN>
__device__ __forceinline__ void splitk_finish_tile(const GemmP& p, const float* __restrict__ gws, const float* __restrict__ gwsb, float* __restrict__ gC,
                                                   const float* __restrict__ gbias, float* __restrict__ gcsum, bool csum_on, int m0, int n0) {
    const long gMN = (long)p.M * p.N;
    const bool vec = (p.N & 3) == 0 && ((reinterpret_cast<uintptr_t>(gws) & 15) == 0);
    const bool vecc = vec && (p.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(gC) & 15) == 0);
    for (int idx = threadIdx.x; idx < BM * (BN / 4); idx += blockDim.x) {
        const int row = m0 + idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
        if (row >= p.M || col >= p.N) continue;
        double sd[4];
        int nv = p.N - col < 4 ? p.N - col : 4;
        const float* src = gws + (long)row * p.N + col;
        if (vec) splitk_ordered_sum4(src, gMN, p.splits, sd);
        else for (int c = 0; c < nv; ++c) sd[c] = splitk_ordered_sum(src + c, gMN, p.splits);
        float* dst = gC + (long)row * p.ldc + col;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float x = 0.f;
            if (c < nv) {
                x = p.alpha * (float)sd[c] + (gbias ? gbias[col + c] : 0.f);
                if (p.beta != 0.f) x += p.beta * dst[c];
                x = apply_act(x, p.act);
            }
            v[c] = x;
        }
        if (vecc) *reinterpret_cast<f32x4*>(dst) = v;
        else for (int c = 0; c < nv; ++c) dst[c] = v[c];
    }
    if (csum_on) {
        for (int r = threadIdx.x; r < BM; r += blockDim.x) {
            const int m = m0 + r;
            if (m >= p.M) continue;
            const double sd = splitk_ordered_sum(gwsb + m, p.M, p.splits);
            gcsum[m] = (p.csum_beta != 0.f ? p.csum_beta * gcsum[m] : 0.f) + (float)sd;
        }
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2_t v = {a, b};
    bf16x2_t h = __builtin_convertvector(v, bf16x2_t);                 // v_cvt_pk_bf16_f32, round-to-nearest-even
    hi = __builtin_bit_cast(unsigned, h);
    f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
    lo = __builtin_bit_cast(unsigned, l);
}

// one fp32 value -> a 32-bit word {hi bf16 | lo bf16}: hi = bf16(x) in the upper half, lo = bf16(x - hi) in the lower half
__device__ __forceinline__ float4 pack_hilo4(float4 v) {
    unsigned h0, l0, h1, l1;
    split_bf16x2(v.x, v.y, h0, l0); split_bf16x2(v.z, v.w, h1, l1);
    float4 r;
    r.x = __uint_as_float(__builtin_amdgcn_perm(h0, l0, 0x05040100u));      // {hi.lower | lo.lower}   (v_perm_b32: one op per word)
    r.y = __uint_as_float(__builtin_amdgcn_perm(h0, l0, 0x07060302u));      // {hi.upper | lo.upper}
    r.z = __uint_as_float(__builtin_amdgcn_perm(h1, l1, 0x05040100u));
    r.w = __uint_as_float(__builtin_amdgcn_perm(h1, l1, 0x07060302u));
    return r;
}


// ---- implicit-GEMM gather (A_IM): per-slot state computed ONCE per block ---------------------------------------------------
// For the pixel a staging slot serves, the element offset of tap (kh, kw) is affine:  base + sign * ((kh' * GW + kw') * GC)
// with kh' = kh (forward, stride-1 data gradient) or kh >> 1 (stride-2 data gradient, parity-matched taps only), and the
// border / parity test of every tap is folded into one bit of a 32-bit mask.  The k loop then needs one scalar tap offset per
// tile and one 64-bit add + one bit test per load instead of re-deriving coordinates (measured: VALU 7.2 -> ~4 per MFMA).
struct ImSlot { long base; unsigned mask; };

__device__ __forceinline__ ImSlot im_slot(const ConvGeom& g, int m, bool on) {
    ImSlot s; s.base = 0; s.mask = 0u;
    if (!on) return s;
    const int ox = m % g.OW; const int t = m / g.OW; const int oy = t % g.OH; const int img = t / g.OH;
    if (!g.transposed) {
        const int y0 = oy * g.stride - g.pad, x0 = ox * g.stride - g.pad;
        s.base = (((long)img * g.GH + y0) * g.GW + x0) * g.GC;
        for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw) {
                const int iy = y0 + kh, ix = x0 + kw;
                if (iy >= 0 && iy < g.GH && ix >= 0 && ix < g.GW) s.mask |= 1u << (kh * g.KW + kw);
            }
    } else {
        const int u = oy + g.pad, v = ox + g.pad;
        const int sh = g.stride == 2 ? 1 : 0;
        s.base = (((long)img * g.GH + (u >> sh)) * g.GW + (v >> sh)) * g.GC;
        for (int kh = 0; kh < g.KH; ++kh)
            for (int kw = 0; kw < g.KW; ++kw) {
                const int ty = u - kh, tx = v - kw;
                bool ok = ty >= 0 && tx >= 0;
                if (sh) ok = ok && !((ty | tx) & 1);
                ok = ok && (ty >> sh) < g.GH && (tx >> sh) < g.GW;
                if (ok) s.mask |= 1u << (kh * g.KW + kw);
            }
    }
    return s;
}
// scalar (block-uniform) cursor over the k axis = (tap, channel): advanced by one tile depth per load_tile call
struct ImCursor {
    int tap, c0, kh, kw;
    __device__ __forceinline__ void init(const ConvGeom& g, int k) { tap = k / g.GC; c0 = k % g.GC; kh = tap / g.KW; kw = tap % g.KW; }
    __device__ __forceinline__ long tap_offset(const ConvGeom& g) const {
        if (!g.transposed) return ((long)kh * g.GW + kw) * g.GC;
        const int sh = g.stride == 2 ? 1 : 0;
        return -((long)(kh >> sh) * g.GW + (kw >> sh)) * g.GC;
    }
    __device__ __forceinline__ void advance(const ConvGeom& g, int bk) {
        c0 += bk;
        if (c0 >= g.GC) { c0 -= g.GC; ++tap; if (++kw == g.KW) { kw = 0; ++kh; } }
    }
};

template <bool VEC>
__device__ __forceinline__ float4 ld4_guard(const float* p, int valid) {   // valid = number of in-range elements (<=4)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4) {
        if (VEC) return *reinterpret_cast<const float4*>(p);
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
        return v;
    }
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    return v;
}

// Workgroup -> tile mapping.  The hardware dispatches workgroups in blockIdx order (x fastest) and deals them round-robin over the 8 XCDs,
// each with its own 4 MB L2.  With tile = (blockIdx.x, blockIdx.y) the concurrently running workgroups walk down ONE column of tiles: every
// A panel is fetched by all XCDs and comes round again only after the whole column (PMC on the 13056 x 900 x 600 projection: 1.86 GB of L2
// requests and 600 MB of L2 misses per launch for 80 MB of operands -- the kernel was bound by that, MFMA busy 71 %).  Here XCD x owns a
// contiguous eighth of the tile sequence, and the sequence runs along the dimension whose shared panel is the larger one (n fastest when
// B is the smaller operand: consecutive tiles reuse the A panel out of L2 and all of B stays resident).  Pure scheduling: results unchanged.
struct TileId { int bx, by; };
__device__ __forceinline__ TileId tile_of_block(int M, int N) {
    const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
    const int lin = blockIdx.y * nbx + blockIdx.x;
    const int xcd = lin & 7, per = total >> 3, rem = total & 7;
    const int seq = xcd * per + (xcd < rem ? xcd : rem) + (lin >> 3);          // XCD x: tiles [x*per + min(x, rem), ...)
    TileId t;
    if (N <= M) { t.bx = seq / nby; t.by = seq - t.bx * nby; }                  // n fastest
    else { t.by = seq / nbx; t.bx = seq - t.by * nbx; }                         // m fastest
    return t;
}

// SPLIT = 1: the LDS tiles hold {hi|lo} bf16 pairs instead of fp32 and the inner product runs as three bf16 MFMAs
// (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate) -- same staging, same tile shapes, every loader mode; 4e-6 rms-rel
// per GEMM instead of 4e-7.  Used (by default) only for WEIGHT gradients, whose error goes straight to the optimizer and
// does not compound through the network.
template <int MI, int NI, int WM, int WN, int AMODE, int BMODE, bool VEC, int BKT, int SPLIT = 0>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr bool A_KCONT = (AMODE != A_MC);          // staged float4 spans 4 consecutive k
    constexpr bool B_KCONT = (BMODE == B_KC);
    constexpr int KQ = BKT / 4;                        // float4 quads per tile row along k
    constexpr int SA = BM * KQ, SB = BN * KQ;          // staging slots (float4) of the A / B tile
    constexpr int NA = (SA + 255) / 256, NB = (SB + 255) / 256;
    // SPLIT = 3 (weight-gradient shapes: both operands arrive with k as the SLOW dimension): the tile is stored as bf16 hi / lo PLANES
    // [k][m] (a staged float4 = four consecutive m -> one 8-byte store per plane) and the MFMA fragments -- 8 consecutive k per lane --
    // come from the gfx950 transpose read ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block row-major and every lane
    // receives one column (probed on hardware: out[lane i][elem j] = in[lane 4j + i/4][elem i%4], tools/probe/tr_probe.hip).  Two such
    // reads form a 32x32x16 operand with no VALU at all; SPLIT = 1 rebuilt every fragment from packed {hi|lo} words with 8 ds_read_b32
    // and 8 v_perm_b32 per fragment, in every wave that needed it -- those kernels were VALU-bound.  Same hi / lo values, same MFMA
    // order: bit-identical to SPLIT = 1.  Plane row strides are = 32 (mod 64) bf16 so that the four k rows of a read hit disjoint banks.
    constexpr bool PLANES = SPLIT == 3 || SPLIT == 4;
    constexpr int PNP = SPLIT == 4 ? 3 : 2;             // SPLIT = 4 (round 4): THREE pieces per operand, six MFMAs per product (fp32-class), same plane layout
    // A k-contiguous operand (A_KC: the dense data gradients dX = dY W) keeps [m][k] planes read with plain 16-byte loads (row stride 24
    // bf16: 16 consecutive rows hit 16 distinct 16-byte slots); its n-contiguous B goes through the transpose reads like above.
    static_assert(!PLANES || (AMODE != A_IM && BMODE != B_KC && BKT == 16), "plane staging: A_MC / A_KC with a k-slow B operand");
    constexpr int LDPA = (BM % 64 == 0) ? BM + 32 : BM, LDPB = (BN % 64 == 0) ? BN + 32 : BN;      // bf16 elements
    constexpr int LDKA = BKT + 8;                                                                  // A_KC: [m][k] plane row
    constexpr int PLA = AMODE == A_MC ? BKT * LDPA : BM * LDKA;                                    // one A plane
    constexpr int PL_BUF = PNP * PLA + PNP * BKT * LDPB;                                           // per buffer: the A pieces, then the B pieces
    constexpr int SMEM_F32 = 2 * BKT * (LDA + LDB), SMEM_PL = (2 * PL_BUF + 1) / 2;
    __shared__ __attribute__((aligned(16))) float smem[PLANES ? (SMEM_PL > 16 * BM ? SMEM_PL : 16 * BM) : SMEM_F32];
    float* As = smem;                      // [2][BKT][LDA]
    float* Bs = smem + 2 * BKT * LDA;      // [2][BKT][LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const TileId tl = tile_of_block(p.M, p.N);
    const int m0 = tl.bx * BM, n0 = tl.by * BN;
    // grouped launch: this block's problem and split; gA .. gcsum are that problem's operands (plain launch: group 0 = the GemmP fields)
    const bool grouped = p.groups > 1;
    const int grp = grouped ? (int)blockIdx.z / p.splits : 0;
    const int zsp = grouped ? (int)blockIdx.z - grp * p.splits : (int)blockIdx.z;
    const float* __restrict__ gA = grouped ? p.Ag[grp] : p.A;
    const float* __restrict__ gB = grouped ? p.Bg[grp] : p.B;
    float* __restrict__ gC = grouped ? p.Cg[grp] : p.C;
    const float* __restrict__ gbias = grouped ? p.biasg[grp] : p.bias;
    float* __restrict__ gcsum = grouped ? p.csumg[grp] : p.csum;
    const long gMN = (long)p.M * p.N;
    float* gws = p.ws + (long)grp * p.splits * gMN;                                                      // this group's slabs
    float* gwsb = p.ws + (long)(grouped ? p.groups : 1) * p.splits * gMN + (long)grp * p.splits * p.M;    // its bias partials
    const int kbeg = zsp * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + BKT - 1) / BKT;
    const ConvGeom g = p.g;

    // ---- per-thread staging slots ----
    int a_r[NA], a_c[NA];   // KCONT: r = tile row (m), c = k offset (0,4,8,12);  MCONT: r = k row, c = m offset
    bool a_on[NA];
    ImSlot a_im[NA];                          // A_IM: gather base offset + per-tap validity mask of the slot's pixel
    ImCursor a_cur;
    if (AMODE == A_IM) a_cur.init(g, kbeg);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int s = tid + i * 256;
        a_on[i] = s < SA;
        if (A_KCONT) { a_r[i] = s / KQ; a_c[i] = (s % KQ) * 4; }
        else { a_r[i] = s / (BM / 4); a_c[i] = (s % (BM / 4)) * 4; }
        if (AMODE == A_IM) {
            int m = m0 + a_r[i];
            a_on[i] = a_on[i] && m < p.M;
            a_im[i] = im_slot(g, m, a_on[i]);
            a_im[i].base += a_c[i];
        }
    }
    int b_r[NB], b_c[NB];
    bool b_on[NB];
    int b_kh[NB], b_kw[NB], b_ci[NB];   // B_IM: filter tap / channel of the slot's columns (fixed for the block)
    int b_ox[NB], b_oy[NB], b_img[NB];  // B_IM: pixel coordinates of the slot's current k row, advanced incrementally
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int s = tid + i * 256;
        b_on[i] = s < SB;
        if (B_KCONT) { b_r[i] = s / KQ; b_c[i] = (s % KQ) * 4; }
        else { b_r[i] = s / (BN / 4); b_c[i] = (s % (BN / 4)) * 4; }
        if (BMODE == B_IM) {
            int n = n0 + b_c[i];
            b_on[i] = b_on[i] && n < p.N;
            int nn = b_on[i] ? n : 0;
            int tap = nn / g.GC;
            b_ci[i] = nn % g.GC; b_kh[i] = tap / g.KW; b_kw[i] = tap % g.KW;
            int pix = kbeg + b_r[i];                         // one division per block; load_tile steps by BKT afterwards
            b_ox[i] = pix % g.OW; int t = pix / g.OW; b_oy[i] = t % g.OH; b_img[i] = t / g.OH;
        }
    }

    float4 ra[NA], rb[NB];
    float4 bsum[NA];                          // A_MC + p.csum: this thread's column sums of the A tiles (k rows a_r + 16 j)
    const bool csum_on = AMODE == A_MC && gcsum != nullptr && tl.by == 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tile = [&](int kt) {
        const int k0 = kbeg + kt * BKT;
        // ---- A ----
        if (AMODE == A_IM) {
            // a BKT-wide k tile never straddles a tap (GC % BKT == 0); tiles are visited in order, so the (tap, channel) cursor
            // is advanced instead of divided out
            const long koff = a_cur.tap_offset(g) + a_cur.c0;
            const unsigned bit = 1u << a_cur.tap;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((a_im[i].mask & bit) && k0 + a_c[i] < kend) v = *reinterpret_cast<const float4*>(gA + (a_im[i].base + koff));
                ra[i] = v;
            }
            a_cur.advance(g, BKT);
        } else if (AMODE == A_KC) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int m = m0 + a_r[i], k = k0 + a_c[i];
                int valid = (a_on[i] && m < p.M) ? (kend - k) : 0;
                ra[i] = ld4_guard<VEC>(gA + (long)m * p.lda + k, valid);
            }
        } else {   // A_MC
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int k = k0 + a_r[i], m = m0 + a_c[i];
                int valid = (a_on[i] && k < kend) ? (p.M - m) : 0;
                ra[i] = ld4_guard<VEC>(gA + (long)k * p.lda + m, valid);
            }
        }
        // ---- B ----
        if (BMODE == B_IM) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                int pix = k0 + b_r[i];
                if (b_on[i] && pix < kend) {
                    int iy = b_oy[i] * g.stride - g.pad + b_kh[i], ix = b_ox[i] * g.stride - g.pad + b_kw[i];
                    if (iy >= 0 && iy < g.GH && ix >= 0 && ix < g.GW)
                        v = *reinterpret_cast<const float4*>(gB + (((long)b_img[i] * g.GH + iy) * g.GW + ix) * g.GC + b_ci[i]);
                }
                rb[i] = v;
                // advance this slot's pixel by one tile depth (tiles are visited in order): no div/mod in the loop
                b_ox[i] += BKT;
                while (b_ox[i] >= g.OW) { b_ox[i] -= g.OW; if (++b_oy[i] == g.OH) { b_oy[i] = 0; ++b_img[i]; } }
            }
        } else if (BMODE == B_KC) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int n = n0 + b_r[i], k = k0 + b_c[i];
                int valid = (b_on[i] && n < p.N) ? (kend - k) : 0;
                rb[i] = ld4_guard<VEC>(gB + (long)n * p.ldb + k, valid);
            }
        } else {   // B_NC
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int k = k0 + b_r[i], n = n0 + b_c[i];
                int valid = (b_on[i] && k < kend) ? (p.N - n) : 0;
                rb[i] = ld4_guard<VEC>(gB + (long)k * p.ldb + n, valid);
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* as = As + buf * BKT * LDA;
        float* bs = Bs + buf * BKT * LDB;
        if (AMODE == A_MC && csum_on) {
#pragma unroll
            for (int i = 0; i < NA; ++i) { bsum[i].x += ra[i].x; bsum[i].y += ra[i].y; bsum[i].z += ra[i].z; bsum[i].w += ra[i].w; }
        }
        if constexpr (PLANES) {
            unsigned short* pl = reinterpret_cast<unsigned short*>(smem) + buf * PL_BUF;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (!(tid + i * 256 < SA)) continue;
                unsigned u[PNP], w[PNP];
                splitn_bf16<PNP>(ra[i].x, ra[i].y, u); splitn_bf16<PNP>(ra[i].z, ra[i].w, w);
                unsigned short* d = pl + (AMODE == A_MC ? a_r[i] * LDPA + a_c[i] : a_r[i] * LDKA + a_c[i]);    // A_KC: four consecutive k of row a_r
#pragma unroll
                for (int q = 0; q < PNP; ++q) *reinterpret_cast<uint2*>(d + q * PLA) = make_uint2(u[q], w[q]);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (!(tid + i * 256 < SB)) continue;
                unsigned u[PNP], w[PNP];
                splitn_bf16<PNP>(rb[i].x, rb[i].y, u); splitn_bf16<PNP>(rb[i].z, rb[i].w, w);
                unsigned short* d = pl + PNP * PLA + b_r[i] * LDPB + b_c[i];
#pragma unroll
                for (int q = 0; q < PNP; ++q) *reinterpret_cast<uint2*>(d + q * BKT * LDPB) = make_uint2(u[q], w[q]);
            }
            return;
        }
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = pack_hilo4(ra[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = pack_hilo4(rb[i]);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!(tid + i * 256 < SA)) continue;
            if (A_KCONT) {
                float* d = as + a_c[i] * LDA + a_r[i];
                d[0] = ra[i].x; d[LDA] = ra[i].y; d[2 * LDA] = ra[i].z; d[3 * LDA] = ra[i].w;
            } else {
                *reinterpret_cast<float4*>(as + a_r[i] * LDA + a_c[i]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!(tid + i * 256 < SB)) continue;
            if (B_KCONT) {
                float* d = bs + b_c[i] * LDB + b_r[i];
                d[0] = rb[i].x; d[LDB] = rb[i].y; d[2 * LDB] = rb[i].z; d[3 * LDB] = rb[i].w;
            } else {
                *reinterpret_cast<float4*>(bs + b_r[i] * LDB + b_c[i]) = rb[i];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* as = As + cur * BKT * LDA + wm * (32 * MI) + l31;
        const float* bs = Bs + cur * BKT * LDB + wn * (32 * NI) + l31;
        if constexpr (PLANES) {
            typedef short s16x4_t __attribute__((ext_vector_type(4)));
            typedef short s16x8_t __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
            const int g4 = lane >> 4, q16 = lane & 15;
            const int krow = 8 * (g4 >> 1) + (q16 >> 2), moff = 16 * (g4 & 1) + 4 * (q16 & 3);
            const unsigned short* pl = reinterpret_cast<const unsigned short*>(smem) + cur * PL_BUF;
            auto frag = [&](const unsigned short* ptr, int ld) {      // k rows krow .. krow+3 and krow+4 .. krow+7 of this lane's k half
                const s16x4_t f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(__attribute__((address_space(3))) const unsigned short*)ptr);
                const s16x4_t f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(__attribute__((address_space(3))) const unsigned short*)(ptr + 4 * ld));
                return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(f0, f1, 0, 1, 2, 3, 4, 5, 6, 7));
            };
            bf16x8_t af[PNP][MI], bf[PNP][NI];
#pragma unroll
            for (int q = 0; q < PNP; ++q) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if constexpr (AMODE == A_MC) {
                        af[q][i] = frag(pl + q * PLA + krow * LDPA + wm * (32 * MI) + i * 32 + moff, LDPA);
                    } else {
                        af[q][i] = *reinterpret_cast<const bf16x8_t*>(pl + q * PLA + (wm * (32 * MI) + i * 32 + l31) * LDKA + 8 * lhi);
                    }
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[q][j] = frag(pl + PNP * PLA + q * BKT * LDPB + krow * LDPB + wn * (32 * NI) + j * 32 + moff, LDPB);
            }
            if constexpr (PNP == 3) {                        // six products, smallest first (conv_planes.hip, pconv_kernel<.., NP = 3>)
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PNP - 1][i], bf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[PNP - 1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (!SPLIT) {
#pragma unroll
            for (int kk = 0; kk < BKT / 2; ++kk) {
                float a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = as[(2 * kk + lhi) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = bs[(2 * kk + lhi) * LDB + j * 32];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // v_mfma_f32_32x32x16_bf16: lane (row = l&31, half = l>>5) supplies k = 8*half .. 8*half+7 of a 16-deep chunk
#pragma unroll
            for (int kc = 0; kc < BKT / 16; ++kc) {
                bf16x8_t ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    unsigned w[8], hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(as[(kc * 16 + 8 * lhi + e) * LDA + i * 32]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x07060302u); lo[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x05040100u); }   /* v_perm_b32: the two upper / lower halves in one op each */
                    ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<uint4*>(hi));
                    al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<uint4*>(lo));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    unsigned w[8], hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(bs[(kc * 16 + 8 * lhi + e) * LDB + j * 32]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x07060302u); lo[q] = __builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x05040100u); }   /* v_perm_b32: the two upper / lower halves in one op each */
                    bh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<uint4*>(hi));
                    bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<uint4*>(lo));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if constexpr (SPLIT == 1) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);   // SPLIT == 2: plain bf16 operands
                    }
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const bool partial = p.splits > 1;
    float* out = partial ? gws + (long)zsp * gMN : gC;
    const long ldo = partial ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (32 * NI) + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (!partial && gbias) ? gbias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= p.M) continue;
                float v = acc[i][j][r];
                float* dst = out + (long)row * ldo + col;
                if (!partial) {
                    v = p.alpha * v + bv;
                    if (p.beta != 0.f) v += p.beta * *dst;
                    v = apply_act(v, p.act);
                }
                *dst = v;
            }
        }
    if constexpr (AMODE == A_MC) {
        if (csum_on) {                                   // block-uniform: combine the 16 k-row slots per column through LDS, fixed order
            float* red = smem;                           // [BKT][BM]; the tiles are dead (last loop iteration ended with a barrier)
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (tid + i * 256 < SA) *reinterpret_cast<float4*>(red + a_r[i] * BM + a_c[i]) = bsum[i];
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < BKT; ++r) t += red[r * BM + tid];
                if (partial) gwsb[(long)zsp * p.M + m0 + tid] = t;
                else gcsum[m0 + tid] = (p.csum_beta != 0.f ? p.csum_beta * gcsum[m0 + tid] : 0.f) + t;
            }
        }
    }
    if (partial && p.tickets != nullptr) {                // in-kernel reduction: the last of this tile's k slices to arrive finishes the tile
        int* sh = reinterpret_cast<int*>(smem);
        if (splitk_last_arriver(p.tickets + ((long)grp * gridDim.y + tl.by) * gridDim.x + tl.bx, p.splits, sh))
            splitk_finish_tile<BM, BN>(p, gws, gwsb, gC, gbias, gcsum, AMODE == A_MC && csum_on, m0, n0);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") core: fp32-class results from the bf16 matrix pipe.
// Each fp32 operand is split at LDS-staging time into hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits kept, residual
// <= 2^-18 |x|); the product a*b is accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on
// v_mfma_f32_32x32x16_bf16.  Three bf16 MFMAs (3 x 32 cycles per 32x32x16) replace eight fp32 MFMAs (8 x 64 cycles), i.e.
// 5.3x less matrix-pipe time; the dropped a_lo*b_lo term and the residuals bound the per-product error at ~1.1e-5 |ab|
// (typ. 4e-6), the same class as the rounding of a K ~ 1e3 fp32 accumulation chain -- parity tests use unchanged tolerances.
// Operands whose staged float4 runs along k only: A = KC or IM (im2col), B = KC.  LDS tiles are [row][k] bf16 with an
// 80-byte row stride (16 consecutive rows hit 16 distinct 16-byte slots: conflict-free ds_read_b128 fragment loads).

// Split-bf16 matrix core for k-contiguous operands (forward GEMMs: y = x W^T, and the forward / data-gradient implicit-GEMM
// convolutions).  Each fp32 value is written as NP bf16 pieces (x = p0 + p1 (+ p2), each the bf16 rounding of what is left),
// staged as NP planes of [row][k] bf16 so that one ds_read_b128 yields a whole v_mfma_f32_32x32x16_bf16 fragment:
//   NP = 3 ("x6"): p0 p0' + p0 p1' + p1 p0' + p1 p1' + p0 p2' + p2 p0'  -- the dropped terms are <= 2^-24 of the product, i.e.
//                  the result is as accurate as the fp32 MFMA chain (three pieces hold all 24 mantissa bits); 6 bf16 MFMAs
//                  (6 x 8 passes) replace 8 fp32 MFMAs (8 x 16 passes) per 32x32x16 block.  Opt-in (mode bit 3).
//   NP = 2 ("x3"): p0 p0' + p0 p1' + p1 p0' -- ~4e-6 rms-rel per GEMM; opt-in for forward work (mode bit 0).
template <int NP> struct X3Cfg { static constexpr int BK = (NP == 3) ? 16 : 32; static constexpr int LDK = BK + 8; };   // NP = 1: plain bf16 operands

template <int NP>
__device__ __forceinline__ void split_pieces(const float4& v, uint2* out /* [NP] : 4 bf16 each */) {
    unsigned h0, h1;
    {
        f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
        h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t));
        h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t));
    }
    out[0] = make_uint2(h0, h1);
    if constexpr (NP == 1) return;
    f32x2_t r0 = {v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u)};
    f32x2_t r1 = {v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u)};
    unsigned m0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf16x2_t));
    unsigned m1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
    if constexpr (NP >= 2) out[1] = make_uint2(m0, m1);
    if constexpr (NP == 3) {
        f32x2_t q0 = {r0[0] - __uint_as_float(m0 << 16), r0[1] - __uint_as_float(m0 & 0xffff0000u)};
        f32x2_t q1 = {r1[0] - __uint_as_float(m1 << 16), r1[1] - __uint_as_float(m1 & 0xffff0000u)};
        out[2] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(q0, bf16x2_t)),
                            __builtin_bit_cast(unsigned, __builtin_convertvector(q1, bf16x2_t)));
    }
}

template <int MI, int NI, int WM, int WN, int AMODE, int NP>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmP p) {
    constexpr int X3_BK = X3Cfg<NP>::BK, X3_LDK = X3Cfg<NP>::LDK;
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    constexpr int KQ = X3_BK / 4;
    constexpr int SA = BM * KQ, SB = BN * KQ;
    constexpr int NA = (SA + 255) / 256, NB = (SB + 255) / 256;
    constexpr int PLANE_A = BM * X3_LDK, PLANE_B = BN * X3_LDK;         // bf16 elements per piece plane
    constexpr int BUF = NP * (PLANE_A + PLANE_B);
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const TileId tl = tile_of_block(p.M, p.N);
    const int m0 = tl.bx * BM, n0 = tl.by * BN;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg + X3_BK - 1) / X3_BK;
    const ConvGeom g = p.g;

    int a_r[NA], a_c[NA]; bool a_on[NA];
    ImSlot a_im[NA];
    ImCursor a_cur;
    if (AMODE == A_IM) a_cur.init(g, kbeg);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int s = tid + i * 256;
        a_on[i] = s < SA;
        a_r[i] = s / KQ; a_c[i] = (s % KQ) * 4;
        if (AMODE == A_IM) {
            int m = m0 + a_r[i];
            a_on[i] = a_on[i] && m < p.M;
            a_im[i] = im_slot(g, m, a_on[i]);
            a_im[i].base += a_c[i];
        }
    }
    int b_r[NB], b_c[NB]; bool b_on[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int s = tid + i * 256;
        b_on[i] = s < SB;
        b_r[i] = s / KQ; b_c[i] = (s % KQ) * 4;
    }
    float4 ra[NA], rb[NB];

    auto load_tile = [&](int kt) {
        const int k0 = kbeg + kt * X3_BK;
        if (AMODE == A_IM) {
            const long koff = a_cur.tap_offset(g) + a_cur.c0;   // a k tile stays inside one filter tap (GC % X3_BK == 0)
            const unsigned bit = 1u << a_cur.tap;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((a_im[i].mask & bit) && k0 + a_c[i] < kend) v = *reinterpret_cast<const float4*>(p.A + (a_im[i].base + koff));
                ra[i] = v;
            }
            a_cur.advance(g, X3_BK);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int m = m0 + a_r[i], k = k0 + a_c[i];
                int valid = (a_on[i] && m < p.M) ? (kend - k) : 0;
                ra[i] = ld4_guard<true>(p.A + (long)m * p.lda + k, valid);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int n = n0 + b_r[i], k = k0 + b_c[i];
            int valid = (b_on[i] && n < p.N) ? (kend - k) : 0;
            rb[i] = ld4_guard<true>(p.B + (long)n * p.ldb + k, valid);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned short* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!(tid + i * 256 < SA)) continue;
            uint2 pc[NP];
            split_pieces<NP>(ra[i], pc);
            const int o = a_r[i] * X3_LDK + a_c[i];
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(base + q * PLANE_A + o) = pc[q];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!(tid + i * 256 < SB)) continue;
            uint2 pc[NP];
            split_pieces<NP>(rb[i], pc);
            const int o = NP * PLANE_A + b_r[i] * X3_LDK + b_c[i];
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(base + q * PLANE_B + o) = pc[q];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) { load_tile(0); store_tile(0); }
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned short* ap = smem + cur * BUF + (wm * 32 * MI + l31) * X3_LDK + 8 * lhi;
        const unsigned short* bp = smem + cur * BUF + NP * PLANE_A + (wn * 32 * NI + l31) * X3_LDK + 8 * lhi;
#pragma unroll
        for (int kc = 0; kc < X3_BK / 16; ++kc) {
            bf16x8_t a[NP][MI], b[NP][NI];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[q][i] = *reinterpret_cast<const bf16x8_t*>(ap + q * PLANE_A + i * 32 * X3_LDK + kc * 16);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[q][j] = *reinterpret_cast<const bf16x8_t*>(bp + q * PLANE_B + j * 32 * X3_LDK + kc * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // smallest products first
                    if constexpr (NP == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP == 3 ? 2 : 0][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[NP == 3 ? 2 : 0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP == 3 ? 1 : 0][i], b[NP == 3 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                    }
                    if constexpr (NP >= 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NP >= 2 ? 1 : 0][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[NP >= 2 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    const bool partial = p.splits > 1;
    float* out = partial ? p.ws + (long)blockIdx.z * p.M * p.N : p.C;
    const long ldo = partial ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (32 * NI) + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (!partial && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= p.M) continue;
                float v = acc[i][j][r];
                float* dst = out + (long)row * ldo + col;
                if (!partial) {
                    v = p.alpha * v + bv;
                    if (p.beta != 0.f) v += p.beta * *dst;
                    v = apply_act(v, p.act);
                }
                *dst = v;
            }
        }
    if (partial && p.tickets != nullptr) {                // in-kernel split-K reduction (see gemm_kernel)
        int* sh = reinterpret_cast<int*>(smem);
        if (splitk_last_arriver(p.tickets + (long)tl.by * gridDim.x + tl.bx, p.splits, sh))
            splitk_finish_tile<BM, BN>(p, p.ws, nullptr, p.C, p.bias, nullptr, false, m0, n0);
    }
}

static int g_c32_dbg = 0;
static int g_c32_fwd3 = 1;      // layer-1 forward on the three-piece direct kernel in the default mode (ha2g_conv_debug_direct_c32 bit 6 clears it)
static int g_direct_c32_dgrad = 0;
static int g_direct_c32_x3 = 1;  // data gradient of the 32->32 channel 3x3 convolutions on the split-bf16 direct kernel (debug bit 2 = off)
static int g_direct_c32 = 1;     // 32->32 channel 3x3 stride-1 forward convolutions on the direct LDS-patch kernel (conv_c32.hip): DEFAULT since it is bit-identical to the implicit GEMM (round 2: same k pairing and order).
                                 // 231 vs 312 us per convolution (-0.5 ms/step) and exact to 2e-6 vs float64, but its different fp32
                                 // summation order moves the chaotic B=4 BatchNorm case (cfg1) to 1.04x its tolerance (3x the
                                 // reference's nine-run fp32 scatter) on one of 1 000 tensors, so the implicit GEMM stays the default.
static int g_np3 = 0;           // mode bit 6 (round 4, DEFAULT ON through ha2g_amd/_lib.py): the split backward products use THREE bf16 pieces per
                                // operand and six MFMAs (all 24 mantissa bits: fp32-class, the reference's arithmetic) instead of two pieces / three
                                // MFMAs (16-bit operand mantissa).  Families without a three-piece kernel run the exact fp32 MFMA in this mode.
static int g_split_dgrad = 1;   // data-gradient GEMMs / convolutions on the split-bf16 inner product (bit 2)
static int g_split_wgrad = 1;   // weight-gradient GEMMs / convolutions on the split-bf16 inner product (ha2g_gemm_set_mode bit 1)
static int g_bf16 = 0;    // every vectorisable GEMM / convolution with plain bf16 operands (1 MFMA per product), fp32 accumulate: mode bit 4.
                          // NOT fp32-class (8 mantissa bits per operand): the `--bf16` bench mode for BASELINE config 5, never the default.
static int g_x6 = 0;      // forward k-contiguous GEMMs / convolutions on the 3-piece split (fp32-accurate), mode bit 3: OPT-IN.
                          // Measured 1.1-1.45x over the fp32 MFMA per GEMM but only 1.4 % of the step (LDS-bandwidth bound), and
                          // being a DIFFERENT fp32-level rounding it lands elsewhere in the reference's own run-to-run scatter.
static int g_x6_min_n = 33;
static int g_x6_dense = 0;   // mode bit 5 (opt-in; measured 60.1 vs 58.0 ms/step): the 3-piece split for forward DENSE GEMMs only (GRU input projections, TCN /
                             // discriminator im2col GEMMs, generator head) -- never the convolutions: the audio tower's forward arithmetic stays frozen
                             // (its own dense GEMMs have N <= 32 or K < 64 and stay on the fp32 MFMA; tests/test_gpu_kernels.py pins that bitwise)
static int g_x3 = 0;      // split-bf16 core: OPT-IN (ha2g_gemm_set_mode(1)).  It is 1.5-2.5x faster on K-contiguous GEMMs / convs
                          // with >= 64 channels but ~10x noisier than the fp32 MFMA chain (4e-6 vs 4e-7 rms-rel per GEMM), which the
                          // reference-derived parity tolerances of the deep audio encoder do not absorb -> exact fp32 is the default.

static int g_direct_c32_wgrad = 1;  // 32 -> 32 channel 3x3 weight gradients on the direct transpose-read kernel (conv_c32.hip); ha2g_conv_debug_cfg(40000) = off
static int g_wgrad_planes = 1;      // weight-gradient shapes: bf16 planes + transpose reads (SPLIT = 3) instead of packed words (SPLIT = 1); bit-identical
static int g_wgrad_wide = 1;        // Cout <= 32 weight gradients: one 32 x 384 tile spans all 9*Cin columns (dy tile staged once instead of 3 times)
static int g_wgrad_blocks = 0;      // 0 = per-shape default (see ha2g_conv2d_wgrad_workspace_bytes); else forced target
static int g_plane_ksplit_model = 1, g_plane_ksplit_force = 0;   // ha2g_gemm_debug_plane_ksplit
static long g_inkernel_bytes = 3L << 19;   // ha2g_gemm_debug_inkernel_bytes
static int g_split_tiles = 192;   // swept on the full step: <=100 is 5-30 % slower, >=192 flat

// sum of p[z * stride] for z = z0, z0 + step, ... < n, accumulated in double IN THAT ORDER; eight loads are issued before the first add (the
// plain loop compiles to load -> s_waitcnt vmcnt(0) -> add per partial: one full memory latency per split)
__device__ __forceinline__ double ordered_sum(const float* __restrict__ p, long stride, int z0, int n, int step) {
    double sd = 0.0;
    int z = z0;
    for (; z + 7 * step < n; z += 8 * step) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(long)(z + j * step) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) sd += (double)v[j];
    }
    for (; z < n; z += step) sd += (double)p[(long)z * stride];
    return sd;
}

// per-group outputs of a (possibly grouped, grid.y = group) reduce launch
struct ReduceOut { float* C[8]; const float* bias[8]; float* csum[8]; int groups; };

__global__ void splitk_reduce_kernel(const float* ws, int splits, long MN, int N, ReduceOut ro, long ldc, float alpha,
                                     float beta, int act, float csum_beta, int M) {
    const int grp = blockIdx.y;
    float* C = ro.C[grp]; const float* bias = ro.bias[grp]; float* csum = ro.csum[grp];
    const float* wg = ws + (long)grp * splits * MN;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) {                                        // the fused bias gradient's partials: [group][splits][M] behind the slabs
        const long m = i - MN;
        if (csum == nullptr || m >= M) return;
        const float* wb = ws + (long)ro.groups * splits * MN + (long)grp * splits * M;
        const double sd = ordered_sum(wb + m, M, 0, splits, 1);
        csum[m] = (csum_beta != 0.f ? csum_beta * csum[m] : 0.f) + (float)sd;
        return;
    }
    const double sd = ordered_sum(wg + i, MN, 0, splits, 1);
    const float s = (float)sd;
    int col = (int)(i % N);
    long row = i / N;
    float v = alpha * s + (bias ? bias[col] : 0.f);
    float* dst = C + row * ldc + col;
    if (beta != 0.f) v += beta * *dst;
    *dst = apply_act(v, act);
}

// Many splits over a small output (weight gradients of the 32/64-channel convolutions: up to ~340 partials of 9-37 k
// floats): one block per 64 outputs, its 4 waves stride the partials, fixed-order LDS combine (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const float* ws, int splits, long MN, int N, ReduceOut ro, long ldc,
                                                                 float alpha, float beta, int act, float csum_beta, int M) {
    __shared__ double part[4][64];
    const int grp = blockIdx.y;
    float* C = ro.C[grp]; const float* bias = ro.bias[grp]; float* csum = ro.csum[grp];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long nmain = (MN + 63) / 64;                    // blocks past the output tiles reduce the fused bias gradient's partials
    const bool tail = blockIdx.x >= nmain;
    const long i = tail ? (long)(blockIdx.x - nmain) * 64 + lane : (long)blockIdx.x * 64 + lane;
    const long cnt = tail ? M : MN;
    const float* src = tail ? ws + (long)ro.groups * splits * MN + (long)grp * splits * M : ws + (long)grp * splits * MN;
    double sd = 0.0;
    if (i < cnt) sd = ordered_sum(src + i, cnt, w, splits, 4);
    part[w][lane] = sd;
    __syncthreads();
    if (w != 0 || i >= cnt) return;
    const float s = (float)((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
    if (tail) { csum[i] = (csum_beta != 0.f ? csum_beta * csum[i] : 0.f) + s; return; }
    int col = (int)(i % N);
    long row = i / N;
    float v = alpha * s + (bias ? bias[col] : 0.f);
    float* dst = C + row * ldc + col;
    if (beta != 0.f) v += beta * *dst;
    *dst = apply_act(v, act);
}

template <int MI, int NI, int WM, int WN, int AMODE, int BMODE, bool VEC>
int launch(const GemmP& p_in, hipStream_t st) {
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    GemmP p = p_in;
    const int groups = p.groups > 1 ? p.groups : 1;
    dim3 grid(ceil_div(p.M, BM), ceil_div(p.N, BN), p.splits * groups);
    p.tickets = nullptr;
    // in-kernel reduction where ONE workgroup can add a tile's slabs in a few microseconds (it reads at ~70 GB/s: <= 1.5 MB); hundreds of slabs over a
    // small output (implicit-GEMM convolution weight gradients) keep the chip-wide reduce launch below
    if (p.splits > 1 && (long)grid.x * grid.y * groups <= HA2G_SPLITK_TICKETS && (long)p.splits * BM * BN * 4 <= g_inkernel_bytes) p.tickets = splitk_tickets_for(st);
    // tile depth BKT = 16.  (BKT = 32 was measured 5-20 % slower on MI355X: fewer resident blocks per CU, more staging
    // registers; the template parameter stays for future tuning.)
    constexpr bool X3_SHAPE = (AMODE == A_KC || AMODE == A_IM) && BMODE == B_KC && VEC;
    bool use_x3 = false;
    if constexpr (X3_SHAPE) if (groups == 1) {
        const bool bwd = AMODE == A_IM && p.g.transposed;       // conv data gradient: handled by the split-bf16 inner product below
        // two-piece planes (b128 fragment reads): opt-in for forward work, DEFAULT for the conv data gradient (1.4x faster there
        // than the packed-word inner product below, same 3-MFMA arithmetic)
        if (g_bf16 && p.K >= 64 && (AMODE != A_IM || p.g.GC % 32 == 0)) {     // bf16 operands, fp32 accumulate (mode bit 4)
            hipLaunchKernelGGL((gemm_x3_kernel<MI, NI, WM, WN, AMODE, 1>), grid, dim3(256), 0, st, p);
            use_x3 = true;
        } else if (bwd && g_split_dgrad && g_np3 && !g_x3) {
            // fp32-class backward: the three-piece core where it serves the shape, else the exact fp32 MFMA below (never two pieces)
            if (p.kchunk >= 64 && p.g.GC % 16 == 0 && p.N > 32) {
                hipLaunchKernelGGL((gemm_x3_kernel<MI, NI, WM, WN, AMODE, 3>), grid, dim3(256), 0, st, p);
                use_x3 = true;
            }
        } else if ((g_x3 || (bwd && g_split_dgrad)) && p.K >= 64 && p.N > 32 && (AMODE != A_IM || p.g.GC % 32 == 0)) {
            hipLaunchKernelGGL((gemm_x3_kernel<MI, NI, WM, WN, AMODE, 2>), grid, dim3(256), 0, st, p);
            use_x3 = true;
        } else if ((g_x6 || (g_x6_dense && AMODE == A_KC)) && !bwd && p.kchunk >= 64 && (AMODE != A_IM || p.g.GC % 16 == 0) && p.N >= g_x6_min_n) {
            hipLaunchKernelGGL((gemm_x3_kernel<MI, NI, WM, WN, AMODE, 3>), grid, dim3(256), 0, st, p);
            use_x3 = true;
        }
    }
    // split-bf16 inner product by role: transposed-A shapes are the weight gradients (dW = dY^T X, conv wgrad; default on),
    // n-contiguous B / the transposed conv gather are the data gradients (mode bit 2)
    bool use_split = false;
    if constexpr (VEC) {
        constexpr bool WGRAD_SHAPE = AMODE == A_MC;
        constexpr bool DGRAD_DENSE = AMODE == A_KC && BMODE == B_NC;
        bool dgrad_conv = AMODE == A_IM && p.g.transposed;
        use_split = !use_x3 && p.kchunk >= 64 &&
                    ((WGRAD_SHAPE && g_split_wgrad) || ((DGRAD_DENSE || dgrad_conv) && g_split_dgrad));
        // three-piece plane tiles that fit the 64 KB of static LDS (the widest dense tiles do not: those shapes run the exact fp32 MFMA)
        constexpr bool P3_FITS = 2 * 2 * 3 * ((AMODE == A_MC ? 16 * ((BM % 64 == 0) ? BM + 32 : BM) : BM * 24) + 16 * ((BN % 64 == 0) ? BN + 32 : BN)) <= 65536;
        if (g_bf16 && !use_x3 && p.kchunk >= 32) {
            hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16, 2>), grid, dim3(256), 0, st, p);
            use_split = true;
        } else if constexpr (WGRAD_SHAPE || DGRAD_DENSE) {
            if (use_split && g_np3) {
                use_split = false;
                if constexpr (P3_FITS) {
                    hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16, 4>), grid, dim3(256), 0, st, p);
                    use_split = true;
                }
            } else if (use_split) {
                if (g_wgrad_planes) hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16, 3>), grid, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16, 1>), grid, dim3(256), 0, st, p);
            }
        } else if constexpr (AMODE == A_IM) {
            if (use_split && g_np3) use_split = false;           // packed-word two-piece core only: fp32-class mode takes the exact fp32 MFMA
            else if (use_split) hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16, 1>), grid, dim3(256), 0, st, p);
        }
    }
    if (!use_x3 && !use_split) hipLaunchKernelGGL((gemm_kernel<MI, NI, WM, WN, AMODE, BMODE, VEC, 16>), grid, dim3(256), 0, st, p);
    HA2G_CHECK_LAUNCH("gemm");
    if (p.splits > 1 && p.tickets == nullptr) {
        long MN = (long)p.M * p.N;
        ReduceOut ro{};
        ro.groups = groups;
        bool cs = false;
        for (int g = 0; g < groups; ++g) {
            ro.C[g] = groups > 1 ? p.Cg[g] : p.C; ro.bias[g] = groups > 1 ? p.biasg[g] : p.bias;
            ro.csum[g] = AMODE == A_MC ? (groups > 1 ? p.csumg[g] : p.csum) : nullptr;
            cs = cs || ro.csum[g] != nullptr;
        }
        if (p.splits >= 16 && MN <= (1 << 20))
            hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64) + (cs ? ceil_div(p.M, 64) : 0), groups), dim3(256), 0, st, p.ws,
                               p.splits, MN, p.N, ro, p.ldc, p.alpha, p.beta, p.act, p.csum_beta, p.M);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(MN + (cs ? p.M : 0), 256), groups), dim3(256), 0, st, p.ws, p.splits, MN, p.N,
                               ro, p.ldc, p.alpha, p.beta, p.act, p.csum_beta, p.M);
        HA2G_CHECK_LAUNCH("splitk_reduce");
    }
    return 0;
}

// Compute units of the current device (256 on an unpartitioned MI355X; fewer under CPX/DPX partitioning or CU masking), queried once per
// device: the tile / split-K heuristics balance work over THIS many CUs.  The arithmetic a shape runs is therefore a function of the CU
// count: the parity fixtures and the bitwise run-to-run guarantees are stated for one device configuration.
static int cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        n[dev] = c;
    }
    return n[dev];
}

// Pick split-K so that a launch has enough workgroups to fill the CUs; returns splits and sets kchunk.
int choose_splits(int M, int N, int K, int BM, int BN, long ws_floats, int* kchunk, int groups = 1) {
    // Split K when the tile grid cannot fill the 256 CUs a couple of times over (measured on the whole train step:
    // splitting below 192 tiles is worth 5-30 % of the step; the workspace round trip + reduce launch is cheap).
    long tiles = (long)ceil_div(M, BM) * ceil_div(N, BN) * groups;
    int splits = 1;
    if (tiles < g_split_tiles && K >= 512) {
        splits = (int)((512 + tiles - 1) / tiles);
        int maxs = K / 128;
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
        while (splits > 1 && (long)groups * splits * M * (N + 1) > ws_floats) --splits;
    }
    int kc = ceil_div(K, splits);
    kc = ceil_div(kc, 32) * 32;
    splits = ceil_div(K, kc);
    *kchunk = kc;
    return splits;
}

// Dense tile shapes.  0-4 serve every operand layout; 5-8 (wider / 96- and 160-column tiles) only the 16-byte-aligned (VEC) paths.
// The arithmetic of one output element does not depend on the tile shape (same k order, same MFMA chain per accumulator), only on the
// split-K count: the split decision below is the round-1 rule (so results are bit-identical to round 1's), the tile shape of the
// k-contiguous-A GEMMs (forward and data-gradient products) is then picked by a time model fitted to tools/gemm_tile_sweep.py on MI355X
// (profiles/r02_gemm_tile_sweep.txt): launch time = (most loaded CU's block count) x block time / latency-hiding factor, where the
// factor drops when a CU holds only one or two blocks (nothing to overlap the tile loads with).
static const int kTileBM[9] = {128, 64, 128, 64, 128, 128, 128, 64, 128}, kTileBN[9] = {128, 128, 64, 64, 32, 192, 160, 192, 96};
static const double kTileEff[5] = {1.00, 0.97, 0.93, 0.88, 0.70};
struct TileModel { double kov, f1s, f2s, f1b, f2b, eff[9]; };
static const TileModel kModelF32   = {0.0,  0.67, 0.88, 0.84, 0.98, {0.71, 0.77, 0.85, 0.83, 0.71, 0.73, 0.71, 0.76, 0.80}};   // fp32 MFMA inner product
static const TileModel kModelSplit = {44.0, 0.69, 0.82, 0.59, 0.97, {1.34, 1.27, 1.33, 1.16, 0.96, 1.49, 1.35, 1.30, 1.31}};   // split-bf16 inner product
static int g_tile_force = -1, g_splits_force = 0;      // ha2g_gemm_debug_tile: tools/gemm_tile_sweep.py
static int g_tile_model = 1;                           // ha2g_gemm_debug_tile(-2, 0) = round-1 tile rule

template <int AMODE, int BMODE, bool VEC>
int dispatch_tile(GemmP& p, long ws_floats, hipStream_t st) {
    // Round-1 rule: every CU's SIMDs share one MFMA pipe, so a launch lasts as long as its most loaded CU: score = per-tile
    // efficiency x useful fraction of the padded tiles x load balance over the CUs (split-K fills the chip when the grid is small,
    // so small grids are scored as balanced).  It still fixes the split-K count.
    int best = 0; double bs = -1.0;
    const int ncu = cu_count();
    const int groups = p.groups > 1 ? p.groups : 1;          // a grouped launch fills the chip with groups x tiles workgroups
    for (int c = 0; c < 5; ++c) {
        long tm = ceil_div(p.M, kTileBM[c]), tn = ceil_div(p.N, kTileBN[c]), tiles = tm * tn * groups;
        double useful = ((double)p.M * p.N * groups) / ((double)tiles * kTileBM[c] * kTileBN[c]);
        double balance = tiles >= g_split_tiles ? ((double)tiles / ncu) / (double)((tiles + ncu - 1) / ncu) : 0.95;
        double score = kTileEff[c] * useful * balance;
        if (score > bs) { bs = score; best = c; }
    }
    p.splits = choose_splits(p.M, p.N, p.K, kTileBM[best], kTileBN[best], ws_floats, &p.kchunk, groups);
    constexpr int NC = VEC ? 9 : 5;
    if (VEC && AMODE == A_KC && g_tile_model && !g_bf16 && !g_x3 && !g_x6 && !g_x6_dense) {
        const TileModel& m = (BMODE == B_NC && g_split_dgrad && p.kchunk >= 64) ? kModelSplit : kModelF32;
        double bt = 1e300;
        const bool np3_dgrad = g_np3 && BMODE == B_NC && g_split_dgrad && p.kchunk >= 64;
        for (int c = 0; c < NC; ++c) {
            if (np3_dgrad && (c == 0 || c == 5 || c == 6)) continue;      // three-piece [m][k] planes of these tiles exceed the 64 KB of static LDS
            long tiles = (long)ceil_div(p.M, kTileBM[c]) * ceil_div(p.N, kTileBN[c]) * groups;
            long load = (tiles * p.splits + ncu - 1) / ncu;
            const bool big = kTileBM[c] * kTileBN[c] >= 128 * 96;
            double occ = load >= 3 ? 1.0 : (load == 2 ? (big ? m.f2b : m.f2s) : (big ? m.f1b : m.f1s));
            double t = (double)load * kTileBM[c] * kTileBN[c] * (p.kchunk + m.kov) / (m.eff[c] * occ);
            if (t < bt) { bt = t; best = c; }
        }
    }
    if (AMODE == A_KC && BMODE == B_NC && g_np3 && g_split_dgrad && best == 0) best = 2;      // (rule / forced-model-off path) see np3_dgrad above
    if (g_tile_force >= 0 && g_tile_force < NC) best = g_tile_force;
    if (g_splits_force > 0 && (long)groups * g_splits_force * p.M * (p.N + 1) <= ws_floats) {
        int kc = ceil_div(ceil_div(p.K, g_splits_force), 32) * 32;
        p.splits = ceil_div(p.K, kc); p.kchunk = kc;
    }
    switch (best) {
        case 0: return launch<2, 2, 2, 2, AMODE, BMODE, VEC>(p, st);
        case 1: return launch<1, 2, 2, 2, AMODE, BMODE, VEC>(p, st);
        case 2: return launch<1, 2, 4, 1, AMODE, BMODE, VEC>(p, st);
        case 3: return launch<1, 1, 2, 2, AMODE, BMODE, VEC>(p, st);
        case 4: return launch<1, 1, 4, 1, AMODE, BMODE, VEC>(p, st);
        default: break;
    }
    if constexpr (VEC && AMODE == A_KC) {
        switch (best) {
            case 5: return launch<2, 3, 2, 2, AMODE, BMODE, VEC>(p, st);
            case 6: return launch<1, 5, 4, 1, AMODE, BMODE, VEC>(p, st);
            case 7: return launch<1, 3, 2, 2, AMODE, BMODE, VEC>(p, st);
            default: return launch<1, 3, 4, 1, AMODE, BMODE, VEC>(p, st);
        }
    }
    return launch<2, 2, 2, 2, AMODE, BMODE, VEC>(p, st);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int g_conv_cfg = -1;     // -1 = heuristic; 0: 256x32, 1: 128x64, 2: 128x128, 3: 64x64, 4: 64x128   (tools/conv_bench.py)

template <int AMODE, int BMODE>
static int launch_conv_cfg(int cfg, const GemmP& p, hipStream_t st) {
    switch (cfg) {
        case 0: return launch<2, 1, 4, 1, AMODE, BMODE, true>(p, st);
        case 1: return launch<1, 2, 4, 1, AMODE, BMODE, true>(p, st);
        case 2: return launch<2, 2, 2, 2, AMODE, BMODE, true>(p, st);
        case 3: return launch<1, 1, 2, 2, AMODE, BMODE, true>(p, st);
        default: return launch<1, 2, 2, 2, AMODE, BMODE, true>(p, st);
    }
}

// workgroups a CU can hold at once are limited, and every CU shares one MFMA pipe per SIMD: a launch runs as long as
// its most loaded CU.  Pick the tile whose (estimated per-tile efficiency x load balance over 256 CUs) is best.
static int pick_conv_cfg(int M, int N) {
    if (g_conv_cfg >= 0) return g_conv_cfg;
    static const int bm[5] = {256, 128, 128, 64, 64}, bn[5] = {32, 64, 128, 64, 128};
    static const double eff[5] = {0.70, 0.95, 0.90, 0.95, 1.00};    // measured on the four trunk shapes (tools/conv_bench.py)
    int best = 2; double bs = -1.0;
    for (int c = 0; c < 5; ++c) {
        if (bn[c] > 32 && N <= 32 && c != 0) continue;
        long tiles = (long)ceil_div(M, bm[c]) * ceil_div(N, bn[c]);
        double waste_n = (double)N / (ceil_div(N, bn[c]) * bn[c]);
        const int ncu = cu_count();
        double per_cu = (double)tiles / ncu;
        double balance = per_cu / (double)((tiles + ncu - 1) / ncu);
        double score = eff[c] * balance * waste_n;
        if (score > bs) { bs = score; best = c; }
    }
    return best;
}

int gemm_split_dgrad_enabled() { return g_split_dgrad && !g_bf16; }
int gemm_bwd_pieces() { return g_np3 ? 3 : 2; }

extern "C" {

/* bit 0: forward GEMMs / convolutions on the split-bf16 core (default 0 = exact fp32: the error compounds through 34 layers
   and breaks parity); bit 1: weight gradients, bit 2: data gradients on the split-bf16 inner product (default 1: the parity
   margins of the full step are unchanged, see tools/margins.py) */
void ha2g_gemm_set_mode(int mode) { g_x3 = mode & 1; g_split_wgrad = (mode >> 1) & 1; g_split_dgrad = (mode >> 2) & 1; g_x6 = (mode >> 3) & 1; g_bf16 = (mode >> 4) & 1; g_x6_dense = (mode >> 5) & 1; g_np3 = (mode >> 6) & 1; }
int ha2g_gemm_bwd_pieces(void) { return (g_bf16 || !(g_split_wgrad || g_split_dgrad)) ? 0 : gemm_bwd_pieces(); }
void ha2g_gemm_debug_x6_min_n(int n) { g_x6_min_n = n; }
void ha2g_gemm_debug_inkernel_bytes(long n) { g_inkernel_bytes = n; }
void ha2g_gemm_debug_plane_ksplit(int model, int force) { g_plane_ksplit_model = model; g_plane_ksplit_force = force; }
void ha2g_gemm_debug_tile(int cfg, int splits) { g_tile_model = cfg != -2; g_tile_force = cfg == -2 ? -1 : cfg; g_splits_force = splits; }
void ha2g_conv_debug_direct_c32(int on) { g_direct_c32 = on & 1; g_direct_c32_dgrad = (on >> 1) & 1; g_direct_c32_x3 = !((on >> 2) & 1); g_c32_dbg = on & 0x30; g_c32_fwd3 = !((on >> 6) & 1); }
void ha2g_conv_debug_cfg(int cfg) { if (cfg >= 40000) g_direct_c32_wgrad = cfg - 40000; else if (cfg >= 30000) g_wgrad_planes = cfg - 30000; else if (cfg >= 20000) g_wgrad_wide = cfg - 20000; else if (cfg >= 10000) g_wgrad_blocks = cfg - 10000; else if (cfg >= 1000) g_split_tiles = cfg - 1000; else g_conv_cfg = cfg; }   /* 1000+n: split-K tile threshold n; 10000+n: wgrad block target n */

// Dense GEMM, row-major.  transa/transb follow BLAS meaning on row-major storage:
//   transa = 0: A is [M,K] (lda >= K);  1: A is stored [K,M] (lda >= M)
//   transb = 0: B is [K,N] (ldb >= N);  1: B is stored [N,K] (ldb >= K)   <- torch Linear weights
// ws / ws_bytes: optional split-K workspace (may be null: no split-K).
int ha2g_gemm_f32(int transa, int transb, int M, int N, int K, float alpha, const float* A, long lda, const float* B,
                  long ldb, float beta, float* C, long ldc, const float* bias, int act, float* ws, long ws_bytes,
                  void* stream) {
    HA2G_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
    if (M == 0 || N == 0) return 0;
    GemmP p{};
    p.M = M; p.N = N; p.K = K; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta; p.bias = bias; p.act = act; p.ws = ws; p.splits = 1; p.kchunk = K;
    hipStream_t st = (hipStream_t)stream;
    long wsf = ws ? ws_bytes / 4 : 0;
    // vector path needs 16-byte aligned rows along the contiguous dimension of both operands
    bool va = aligned16(A) && (lda % 4 == 0) && ((transa ? M : K) % 4 == 0);
    bool vb = aligned16(B) && (ldb % 4 == 0) && ((transb ? K : N) % 4 == 0);
    bool vec = va && vb;
    if (!transa && transb) return vec ? dispatch_tile<A_KC, B_KC, true>(p, wsf, st) : dispatch_tile<A_KC, B_KC, false>(p, wsf, st);
    if (!transa && !transb) return vec ? dispatch_tile<A_KC, B_NC, true>(p, wsf, st) : dispatch_tile<A_KC, B_NC, false>(p, wsf, st);
    if (transa && !transb) return vec ? dispatch_tile<A_MC, B_NC, true>(p, wsf, st) : dispatch_tile<A_MC, B_NC, false>(p, wsf, st);
    return ha2g_set_error(-1, "gemm: transa=1,transb=1 is not used on this path");
}

// Dense product on THREE-PIECE PLANES (round 4): C [M][N] = act(A B^T + bias) + beta C, A = piece planes [M][lda], B = piece planes [N][ldb]
// (ha2g_f32_to_planes_2d_np; lda = ldb = K rounded up to 32, zero padded), on the quantisation-free plane kernel of conv_planes.hip: six bf16
// MFMAs per product on all 24 mantissa bits = the accuracy of the fp32 MFMA GEMM, ~2x its speed on the GRU projection shapes.  Small tile grids
// are split over k (raw slabs in ws, reduced in double by the split-K reduce).  Replaces nn.Linear / the GRU's input projections and their
// autograd backward (model/hierarchy_net.py:87-93,144-147) where ha2g_amd.ops.gemm routes a product here.
int ha2g_gemm_planes_np_f32(const void* a, long a_ps, long lda, const void* b, long b_ps, long ldb, int np, int M, int N, int K, float beta,
                            float* C, long ldc, const float* bias, int act, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(np == 3, "gemm_planes: np = %d (3)", np);
    HA2G_REQUIRE(act >= 0 && act <= 2, "gemm_planes: act %d (0 none, 1 relu, 2 leaky-relu)", act);
    if (M == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int nkt = (K + 31) / 32, ncu = cu_count();
    int ksplit = 1, mt = 0, bn = 0;
    int* tickets = splitk_tickets_for(st);
    if (g_plane_ksplit_model && tickets != nullptr) {
        // Split-K by a time model of the q kernel (round 6).  One workgroup takes ~2.4 us per 32-deep k tile whatever its tile shape (its LOAD phase --
        // the A tile's LDS-DMA -- bounds a k tile, so a 64-column tile is not faster than a 128-column one) plus ~6 us of prologue / epilogue, and a
        // launch lasts rounds x that: the GRU dX product [4352 x 600] x K = 1800 ran ONE round of 170 workgroups x 57 k tiles (150-166 us = 0.11 of the
        // roofline); three k slices of 19 tiles on 240 workgroups of 288 x 128 are one round of a third the depth.  The k slices are
        // added in the kernel by each tile's last arriver (~70 GB/s per workgroup): + 4 us + slab bytes / 70 GB/s.
        double best = 1e300;
        for (int ks = 1; ks <= 8 && (ks == 1 || ks <= nkt / 8); ++ks) {
            if ((long)ks * M * N * 4 > ws_bytes && ks > 1) break;
            int cmt = 0, cbn = 0;
            if (plane_gemm_plan(M, N, ks, &cmt, &cbn) != 0) continue;
            const long tiles = (long)ceil_div(M, 32 * cmt) * ceil_div(N, cbn);
            if (ks > 1 && tiles > HA2G_SPLITK_TICKETS) continue;
            const long rounds = (tiles * ks + ncu - 1) / ncu;
            const int kt_per = (nkt + ks - 1) / ks;
            double t = (double)rounds * (kt_per * 2.4 + 6.0);
            if (ks > 1) t += 4.0 + (double)ks * (32.0 * cmt * cbn * 4.0) / 70e3;
            if (t < best - 1e-9) { best = t; ksplit = ks; mt = cmt; bn = cbn; }
        }
    } else {
        // round 4-5 rule: split-K when the tile grid leaves most CUs idle and k is deep enough (raw slabs, reduced by a second launch)
        tickets = nullptr;
        plane_gemm_plan(M, N, 1, &mt, &bn);
        const long tiles = (long)ceil_div(M, 32 * mt) * ceil_div(N, bn);
        if (tiles * 2 <= ncu && nkt >= 16) {
            ksplit = (int)((ncu + tiles - 1) / tiles);
            if (ksplit > nkt / 8) ksplit = nkt / 8;
            while (ksplit > 1 && (long)ksplit * M * N * 4 > ws_bytes) --ksplit;
            if (ksplit < 1) ksplit = 1;
        }
    }
    if (g_plane_ksplit_force > 0 && g_plane_ksplit_force <= nkt / 2 && (long)g_plane_ksplit_force * M * N * 4 <= ws_bytes) ksplit = g_plane_ksplit_force;
    if (int rc = plane_gemm_launch(a, a_ps, lda, b, b_ps, ldb, M, N, K, C, ldc, beta, bias, act, ws, ksplit, tickets, st)) return rc;
    if (ksplit > 1 && tickets == nullptr) {
        const int kt_per = (nkt + ksplit - 1) / ksplit, used = (nkt + kt_per - 1) / kt_per;      // slabs the kernel wrote (the last slices may be empty)
        const long MN = (long)M * N;
        ReduceOut ro{};
        ro.groups = 1; ro.C[0] = C; ro.bias[0] = bias;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ceil_div(MN, 256), 1), dim3(256), 0, st, ws, used, MN, N, ro, ldc, 1.f, beta, act, 0.f, M);
        HA2G_CHECK_LAUNCH("gemm_planes reduce");
    }
    return 0;
}

// Weight and bias gradient of a linear layer in one launch: dW[M,N] = beta*dW + dY^T X and db[M] = bias_beta*db + column sums of dY,
// dY stored [K][M] (rows = samples), X [K][N].  The bias sums ride on the dY tiles the GEMM stages anyway (fp32 per block, double across
// split-K partials in the reduce launch).
int ha2g_gemm_wgrad_bias_f32(int M, int N, int K, const float* dY, long ldy, const float* X, long ldx, float beta, float* dW, long ldw,
                             float bias_beta, float* db, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_wgrad_bias: negative dimension");
    HA2G_REQUIRE(db != nullptr, "gemm_wgrad_bias: null bias gradient");
    if (M == 0 || N == 0) return 0;
    GemmP p{};
    p.M = M; p.N = N; p.K = K; p.A = dY; p.lda = ldy; p.B = X; p.ldb = ldx; p.C = dW; p.ldc = ldw;
    p.alpha = 1.f; p.beta = beta; p.bias = nullptr; p.act = 0; p.ws = ws; p.splits = 1; p.kchunk = K;
    p.csum = db; p.csum_beta = bias_beta;
    hipStream_t st = (hipStream_t)stream;
    long wsf = ws ? ws_bytes / 4 : 0;
    bool vec = aligned16(dY) && (ldy % 4 == 0) && (M % 4 == 0) && aligned16(X) && (ldx % 4 == 0) && (N % 4 == 0);
    return vec ? dispatch_tile<A_MC, B_NC, true>(p, wsf, st) : dispatch_tile<A_MC, B_NC, false>(p, wsf, st);
}

// `groups` (<= 8) independent GEMMs of ONE shape in one launch (+ one reduce launch when split-K is used): the same layer of several
// networks with their own weights -- the three / six generators' text encoders run in lockstep.  Semantics per group g as ha2g_gemm_f32 /
// ha2g_gemm_wgrad_bias_f32 with A = Ag[g] etc.; bias / csum arrays (or their entries) may be null; csum only with transa = 1, transb = 0.
int ha2g_gemm_grouped_f32(int groups, int transa, int transb, int M, int N, int K, float alpha, const float* const* Ag, long lda,
                          const float* const* Bg, long ldb, float beta, float* const* Cg, long ldc, const float* const* biasg, int act,
                          float* const* csumg, float csum_beta, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(groups >= 1 && groups <= 8, "gemm_grouped: 1..8 groups, got %d", groups);
    HA2G_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_grouped: negative dimension");
    HA2G_REQUIRE(!(transa && transb), "gemm_grouped: transa=1,transb=1 is not used on this path");
    HA2G_REQUIRE(csumg == nullptr || (transa && !transb), "gemm_grouped: column sums ride on the weight-gradient shape only");
    if (M == 0 || N == 0) return 0;
    GemmP p{};
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta; p.act = act; p.ws = ws; p.splits = 1; p.kchunk = K; p.csum_beta = csum_beta;
    p.groups = groups;
    bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && ((transa ? M : K) % 4 == 0) && ((transb ? K : N) % 4 == 0);
    for (int g = 0; g < groups; ++g) {
        p.Ag[g] = Ag[g]; p.Bg[g] = Bg[g]; p.Cg[g] = Cg[g];
        p.biasg[g] = biasg ? biasg[g] : nullptr; p.csumg[g] = csumg ? csumg[g] : nullptr;
        vec = vec && aligned16(Ag[g]) && aligned16(Bg[g]);
    }
    p.A = p.Ag[0]; p.B = p.Bg[0]; p.C = p.Cg[0]; p.bias = p.biasg[0]; p.csum = p.csumg[0];     // groups == 1 degenerates to the plain launch
    hipStream_t st = (hipStream_t)stream;
    long wsf = ws ? ws_bytes / 4 : 0;
    if (!transa && transb) return vec ? dispatch_tile<A_KC, B_KC, true>(p, wsf, st) : dispatch_tile<A_KC, B_KC, false>(p, wsf, st);
    if (!transa && !transb) return vec ? dispatch_tile<A_KC, B_NC, true>(p, wsf, st) : dispatch_tile<A_KC, B_NC, false>(p, wsf, st);
    return vec ? dispatch_tile<A_MC, B_NC, true>(p, wsf, st) : dispatch_tile<A_MC, B_NC, false>(p, wsf, st);
}

// NHWC convolution as implicit GEMM.  x [N,H,W,Cin], w [Cout][KH][KW][Cin] (torch channels_last weight),
// y [N,OH,OW,Cout].  Cin % 16 == 0 (the 1-channel stem has its own kernel in conv_misc.hip).
int ha2g_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, int act, void* stream) {
    HA2G_REQUIRE(Cin % 16 == 0, "conv2d_fwd: Cin=%d must be a multiple of 16", Cin);
    HA2G_REQUIRE(KH * KW <= 32, "conv2d_fwd: at most 32 filter taps");
    int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (g_direct_c32 && Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && bias == nullptr && act <= 1) {
        // fp32-class default mode: the layer-1 forward on three pieces as well (anti-phase direct kernel, conv_c32.hip: 160 vs 230 us); modes 0 / 6 and
        // shapes it does not serve keep the fp32 direct kernel that is bit-identical to the implicit GEMM
        if (g_np3 && g_c32_fwd3) {
            int rc = conv3x3_c32_x3_launch(x, w, y, N, H, W, 0, act, 0.f, (hipStream_t)stream);
            if (rc != -100) return rc;
        }
        int rc = conv3x3_c32_launch(x, w, y, N, H, W, 0, act | g_c32_dbg, 0.f, (hipStream_t)stream);
        if (rc != -100) return rc;
    }
    GemmP p{};
    p.M = N * OH * OW; p.N = Cout; p.K = KH * KW * Cin;
    p.A = x; p.B = w; p.ldb = p.K; p.C = y; p.ldc = Cout; p.alpha = 1.f; p.beta = 0.f; p.bias = bias; p.act = act;
    p.splits = 1; p.kchunk = p.K;
    p.g = ConvGeom{H, W, Cin, OH, OW, KH, KW, stride, pad, 0};
    hipStream_t st = (hipStream_t)stream;
    return launch_conv_cfg<A_IM, B_KC>(pick_conv_cfg(p.M, p.N), p, st);
}

// Data gradient: dx [N,H,W,Cin] = conv_transpose(dy [N,OH,OW,Cout], w).  wt is the weight permuted to
// [Cin][KH][KW][Cout] (ha2g_conv2d_weight_ohwi_to_ihwo).  Cout % 16 == 0.
int ha2g_conv2d_dgrad_f32(const float* dy, const float* wt, float* dx, int N, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, float beta, void* stream) {
    HA2G_REQUIRE(Cout % 16 == 0, "conv2d_dgrad: Cout=%d must be a multiple of 16", Cout);
    HA2G_REQUIRE(stride == 1 || stride == 2, "conv2d_dgrad: stride must be 1 or 2");
    HA2G_REQUIRE(KH * KW <= 32, "conv2d_dgrad: at most 32 filter taps");
    int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1) {
        // direct LDS-patch kernels (conv_c32.hip): split-bf16 by default like every other data gradient; fp32 form = debug bit 1
        int rc = -100;
        if (g_direct_c32_dgrad) rc = conv3x3_c32_launch(dy, wt, dx, N, H, W, 1, 0, beta, (hipStream_t)stream);
        else if (g_split_dgrad && g_direct_c32_x3) rc = conv3x3_c32_x3_launch(dy, wt, dx, N, H, W, 1, 0, beta, (hipStream_t)stream);      // two or three pieces by the mode
        if (rc != -100) return rc;
    }
    GemmP p{};
    p.M = N * H * W; p.N = Cin; p.K = KH * KW * Cout;
    p.A = dy; p.B = wt; p.ldb = p.K; p.C = dx; p.ldc = Cin; p.alpha = 1.f; p.beta = beta; p.bias = nullptr; p.act = 0;
    p.splits = 1; p.kchunk = p.K;
    p.g = ConvGeom{OH, OW, Cout, H, W, KH, KW, stride, pad, 1};
    hipStream_t st = (hipStream_t)stream;
    int cfg = pick_conv_cfg(p.M, p.N);
    // the plane-based split-bf16 kernel is staging-bound, not MFMA-bound: its best tiles differ from the fp32 kernel's
    // (tools/x3_cfg_sweep.py: 64-channel layers 64x64 141 vs 170 us, 256-channel layers 128x128 141 vs 151 us)
    if (g_conv_cfg < 0 && g_split_dgrad && p.N > 32 && Cout % 32 == 0) cfg = p.N <= 64 ? 3 : (p.N >= 256 ? 2 : cfg);
    return launch_conv_cfg<A_IM, B_KC>(cfg, p, st);
}

// dx = conv_transpose(dy, w) + (decision bit ? resid : 0): the data gradient of a block's conv1 with the identity shortcut's gradient added in the epilogue
// (round 6: what beta = 1 onto a materialised dres = dout * (out > 0) did).  32 -> 32 channels, 3x3, stride 1 on the anti-phase direct kernel only
// (ha2g_conv2d_dgrad_resid_supported); resid [N,H,W,32] fp32, resid_bits = ha2g_se_bn_scale_add_relu_mask_np_f32's words over the same tensor.
int ha2g_conv2d_dgrad_resid_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !g_direct_c32_dgrad && g_split_dgrad && g_direct_c32_x3 &&
           gemm_bwd_pieces() == 3 && conv3x3_c32pp_serves(H, W);
}
int ha2g_conv2d_dgrad_resid_f32(const float* dy, const float* wt, float* dx, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                const float* resid, const void* resid_bits, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_dgrad_resid_supported(H, W, Cin, Cout, KH, KW, stride, pad), "conv2d_dgrad_resid: unsupported geometry / mode");
    HA2G_REQUIRE(resid != nullptr && resid_bits != nullptr && dx != resid, "conv2d_dgrad_resid: null residual / bits, or dx aliases the residual");
    const int rc = conv3x3_c32_x3_launch(dy, wt, dx, N, H, W, 1, 0, 0.f, (hipStream_t)stream, resid, (const unsigned*)resid_bits);
    if (rc == -100) return ha2g_set_error(-1, "conv2d_dgrad_resid: the anti-phase 32-channel kernel does not serve H = %d, W = %d", H, W);
    return rc;
}

// Weight gradient: dw [Cout][KH][KW][Cin] (+)= dy^T * im2col(x); K = N*OH*OW output pixels, split over grid.z.
// ws must hold splits*Cout*KH*KW*Cin floats (query with ha2g_conv2d_wgrad_workspace_bytes).
long ha2g_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    long K = (long)N * OH * OW, MN = (long)Cout * KH * KW * Cin;
    int BM = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const bool wide = g_wgrad_wide && Cout <= 32 && KH * KW * Cin <= 384;
    long tiles = (long)ceil_div(Cout, BM) * ceil_div(KH * KW * Cin, wide ? 384 : 128);
    // target workgroups per launch, swept with the split-bf16 inner product (tools/wgrad_sweep.py): many small chunks for the
    // 32-channel layer (huge K, tiny output), fewer for the wide layers whose partial tiles are large
    const long target = g_wgrad_blocks > 0 ? g_wgrad_blocks : (wide ? 512 : (Cout <= 32 ? 1536 : (Cout <= 64 ? 1024 : 768)));
    long splits = (target + tiles - 1) / tiles;
    if (splits > K / 256) splits = K / 256;
    if (splits < 1) splits = 1;
    long bytes = splits * MN * 4;
    if (Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1) {          // direct kernel: one partial per workgroup
        const long direct = (long)conv3x3_c32_wgrad_blocks(N, H, W) * MN * 4;
        if (direct > bytes) bytes = direct;
    }
    return bytes;
}

int ha2g_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "conv2d_wgrad: channels must be multiples of 4");
    int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    GemmP p{};
    p.M = Cout; p.N = KH * KW * Cin; p.K = N * OH * OW;
    p.A = dy; p.lda = Cout; p.B = x; p.C = dw; p.ldc = p.N; p.alpha = 1.f; p.beta = beta; p.bias = nullptr; p.act = 0;
    p.g = ConvGeom{H, W, Cin, OH, OW, KH, KW, stride, pad, 0};
    long need = ha2g_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad);
    long MN = (long)p.M * p.N;
    hipStream_t st0 = (hipStream_t)stream;
    if (g_direct_c32_wgrad && g_split_wgrad && !g_bf16 && Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && ws &&
        ws_bytes >= need) {
        const int nblk = conv3x3_c32_wgrad_launch(x, dy, ws, N, H, W, st0);
        if (nblk != -100) {
            if (nblk < 0) return nblk;
            ReduceOut ro{};
            ro.groups = 1; ro.C[0] = dw;
            hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64), 1), dim3(256), 0, st0, ws, nblk, MN, p.N, ro, p.ldc, 1.f, beta, 0, 0.f,
                               p.M);
            HA2G_CHECK_LAUNCH("conv3x3_c32_wgrad reduce");
            return 0;
        }
    }
    {   // the implicit GEMM's own split count (the direct kernel may have asked for a larger workspace)
        int BM = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
        const bool wide = g_wgrad_wide && Cout <= 32 && KH * KW * Cin <= 384;
        long tiles = (long)ceil_div(Cout, BM) * ceil_div(KH * KW * Cin, wide ? 384 : 128);
        const long target = g_wgrad_blocks > 0 ? g_wgrad_blocks : (wide ? 512 : (Cout <= 32 ? 1536 : (Cout <= 64 ? 1024 : 768)));
        long sp = (target + tiles - 1) / tiles;
        if (sp > (long)p.K / 256) sp = p.K / 256;
        if (sp < 1) sp = 1;
        need = sp * MN * 4;
    }
    int splits = (int)(need / (MN * 4));
    HA2G_REQUIRE(splits == 1 || (ws && ws_bytes >= need), "conv2d_wgrad: workspace too small (%ld < %ld)", ws_bytes, need);
    int kc = ceil_div(p.K, splits);
    kc = ceil_div(kc, 32) * 32;
    p.kchunk = kc; p.splits = ceil_div(p.K, kc); p.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    if (g_wgrad_wide && Cout <= 32 && p.N <= 384) return launch<1, 3, 1, 4, A_MC, B_IM, true>(p, st);
    if (Cout <= 32) return launch<1, 1, 1, 4, A_MC, B_IM, true>(p, st);
    if (Cout <= 64) return launch<2, 1, 1, 4, A_MC, B_IM, true>(p, st);
    return launch<2, 2, 2, 2, A_MC, B_IM, true>(p, st);
}


// Plane-based weight gradient (conv_planes.hip): x and dy arrive as bf16 hi / lo planes; one DMA-staged launch writes `chunks` dW-shaped
// partial slabs into ws, the wide reduce adds them in double.  3x3 / stride 1 / pad 1, channels multiples of 64.
int ha2g_conv2d_wgrad_planes_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return g_split_wgrad && !g_bf16 && Cin % 64 == 0 && Cout % 64 == 0 && pconv_wgrad_supported(H, W, Cin, Cout, KH, KW, stride, pad);
}
// bf16-storage mode: dw (fp32) = beta*dw + dy^T im2col(x) with x and dy bf16 tensors (one plane, one MFMA per product); 3x3 / stride 1 / pad 1,
// channel counts multiples of 32
static bool c32_wgrad_b16_ok(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    // conv3x3_c32_wgrad_kernel<b16>: 32 -> 32 channels, 128-pixel tiles (layer 1's 128 x 70 images do not fit the plane kernel's LDS patch)
    if (!(Cin == 32 && Cout == 32 && KH == 3 && KW == 3 && stride == 1 && pad == 1)) return false;
    const int rows_max = (128 + W - 2) / W + 1 + 2;
    const long lds = ((long)2 * rows_max * (W + 2) * 32 + 2 * 128 * 32) * 2;
    return lds <= 78 * 1024 && (long)H * W >= 128 && W + 2 > 32;
}
int ha2g_conv2d_wgrad_b16_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return pconv_wgrad_supported(H, W, Cin, Cout, KH, KW, stride, pad) || c32_wgrad_b16_ok(H, W, Cin, Cout, KH, KW, stride, pad);
}
long ha2g_conv2d_wgrad_b16_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (pconv_wgrad_supported(H, W, Cin, Cout, 3, 3, 1, 1)) return pconv_wgrad_workspace_bytes(N, H, W, Cin, Cout);
    return (long)conv3x3_c32_wgrad_blocks(N, H, W) * Cout * 9 * Cin * 4;
}
int ha2g_conv2d_wgrad_b16(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          float beta, float* ws, long ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!pconv_wgrad_supported(H, W, Cin, Cout, KH, KW, stride, pad)) {
        HA2G_REQUIRE(c32_wgrad_b16_ok(H, W, Cin, Cout, KH, KW, stride, pad), "conv2d_wgrad_b16: unsupported geometry");
        const long MN = (long)Cout * 9 * Cin;
        HA2G_REQUIRE(ws && ws_bytes >= (long)conv3x3_c32_wgrad_blocks(N, H, W) * MN * 4, "conv2d_wgrad_b16: workspace too small");
        const int nblk = conv3x3_c32_wgrad_b16_launch(x, dy, ws, N, H, W, st);
        if (nblk < 0) return nblk == -100 ? ha2g_set_error(-1, "conv2d_wgrad_b16: the patch does not fit the LDS (W = %d)", W) : nblk;
        ReduceOut ro{};
        ro.groups = 1; ro.C[0] = dw;
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64), 1), dim3(256), 0, st, ws, nblk, MN, 9 * Cin, ro, (long)9 * Cin, 1.f, beta, 0, 0.f, Cout);
        HA2G_CHECK_LAUNCH("conv2d_wgrad_b16 (c32) reduce");
        return 0;
    }
    HA2G_REQUIRE(ws && ws_bytes >= pconv_wgrad_workspace_bytes(N, H, W, Cin, Cout), "conv2d_wgrad_b16: workspace too small");
    const int chunks = pconv_wgrad_launch(x, nullptr, dy, nullptr, ws, N, H, W, Cin, Cout, st);
    if (chunks == -100) return ha2g_set_error(-1, "conv2d_wgrad_b16: the patch does not fit the LDS (W = %d)", W);
    if (chunks < 0) return chunks;
    const long MN = (long)Cout * KH * KW * Cin;
    ReduceOut ro{};
    ro.groups = 1; ro.C[0] = dw;
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64), 1), dim3(256), 0, st, ws, chunks, MN, KH * KW * Cin, ro, (long)KH * KW * Cin, 1.f,
                       beta, 0, 0.f, Cout);
    HA2G_CHECK_LAUNCH("conv2d_wgrad_b16 reduce");
    return 0;
}
long ha2g_conv2d_wgrad_planes_workspace_bytes(int N, int H, int W, int Cin, int Cout) { return pconv_wgrad_workspace_bytes(N, H, W, Cin, Cout); }
// np = 2 or 3 equally spaced piece planes of x and dy (piece q at base + q * ps elements); np = 3 is the fp32-class default
int ha2g_conv2d_wgrad_planes_np_f32(const void* x, long x_ps, const void* dy, long dy_ps, int np, float* dw, int N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_wgrad_planes_supported(H, W, Cin, Cout, KH, KW, stride, pad), "conv2d_wgrad_planes: unsupported geometry / mode");
    HA2G_REQUIRE(np == 2 || np == 3, "conv2d_wgrad_planes: np = %d (2 or 3)", np);
    HA2G_REQUIRE(ws && ws_bytes >= pconv_wgrad_workspace_bytes(N, H, W, Cin, Cout), "conv2d_wgrad_planes: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = pconv_wgrad_launch_np(x, x_ps, dy, dy_ps, np, ws, N, H, W, Cin, Cout, st);
    if (chunks == -100) return ha2g_set_error(-1, "conv2d_wgrad_planes: the patch does not fit the LDS (W = %d)", W);
    if (chunks < 0) return chunks;
    const long MN = (long)Cout * KH * KW * Cin;
    ReduceOut ro{};
    ro.groups = 1; ro.C[0] = dw;
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64), 1), dim3(256), 0, st, ws, chunks, MN, KH * KW * Cin, ro, (long)KH * KW * Cin, 1.f,
                       beta, 0, 0.f, Cout);
    HA2G_CHECK_LAUNCH("conv2d_wgrad_planes reduce");
    return 0;
}
int ha2g_conv2d_wgrad_planes_f32(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N, int H, int W, int Cin,
                                 int Cout, int KH, int KW, int stride, int pad, float beta, float* ws, long ws_bytes, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_wgrad_planes_supported(H, W, Cin, Cout, KH, KW, stride, pad), "conv2d_wgrad_planes: unsupported geometry / mode");
    HA2G_REQUIRE(ws && ws_bytes >= pconv_wgrad_workspace_bytes(N, H, W, Cin, Cout), "conv2d_wgrad_planes: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = pconv_wgrad_launch(x_hi, x_lo, dy_hi, dy_lo, ws, N, H, W, Cin, Cout, st);
    if (chunks == -100) return ha2g_set_error(-1, "conv2d_wgrad_planes: the patch does not fit the LDS (W = %d)", W);
    if (chunks < 0) return chunks;
    const long MN = (long)Cout * KH * KW * Cin;
    ReduceOut ro{};
    ro.groups = 1; ro.C[0] = dw;
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(ceil_div(MN, 64), 1), dim3(256), 0, st, ws, chunks, MN, KH * KW * Cin, ro, (long)KH * KW * Cin, 1.f,
                       beta, 0, 0.f, Cout);
    HA2G_CHECK_LAUNCH("conv2d_wgrad_planes reduce");
    return 0;
}

}  // extern "C"
