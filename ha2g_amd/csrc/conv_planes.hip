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

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// zeros: the source of every masked-out 16-byte piece (zero padding of the convolution, rows past M)
__device__ __attribute__((aligned(64))) unsigned short g_zero_page[64];

__global__ void f32_to_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uint2 h, l;
        split2_bf16(v.x, v.y, h.x, l.x); split2_bf16(v.z, v.w, h.y, l.y);
        reinterpret_cast<uint2*>(hi)[i] = h;
        reinterpret_cast<uint2*>(lo)[i] = l;
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

struct PConvP {
    const unsigned short* a_hi; const unsigned short* a_lo;      // dy planes [img][GH][GW][GC]
    const unsigned short* b_hi; const unsigned short* b_lo;      // weight planes [N][K], K = KH*KW*GC
    float* C; long ldc; float beta;
    int M, N, K;
    int GH, GW, GC, OH, OW, KH, KW, pad;                         // gathered tensor / output pixel grid (stride 1)
    int dbg;                                                     // timing ablations (ha2g_conv_planes_debug): 1 = no DMA after tile 0, 2 = no MFMA
};

// XCD-aware workgroup -> tile mapping (same rule as gemm.hip's tile_of_block: XCD x owns a contiguous eighth of the tile sequence, n fastest)
__device__ __forceinline__ void ptile_of_block(int& bx, int& by) {
    const int nbx = gridDim.x, nby = gridDim.y, total = nbx * nby;
    const int lin = blockIdx.y * nbx + blockIdx.x;
    const int xcd = lin & 7, per = total >> 3, rem = total & 7;
    const int seq = xcd * per + (xcd < rem ? xcd : rem) + (lin >> 3);
    bx = seq / nby; by = seq - bx * nby;
}

// Data gradient of a stride-1 convolution as an implicit GEMM over planes:  dx[m][n] (+)= sum_{tap, co} dy[pix(m) - tap][co] * wt[n][tap][co].
// BM x BN output tile, 4 waves as WM x WN, wave tile (32 MI) x (32 NI), k tile = 32 channels of one filter tap, two LDS buffers,
// one barrier per k tile: the next tile's DMA is issued before the current tile's MFMAs and waited for after them.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void pconv_dgrad_kernel(PConvP p) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
    constexpr int RA = BM / 16, RB = BN / 16;                    // 16-row DMA pieces (1 KiB per wave instruction) per plane
    static_assert(RA % 4 == 0 && RB % 4 == 0, "every wave stages whole row blocks");
    constexpr int NA = RA / 4, NB = RB / 4;                      // row blocks per wave, per plane
    constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;          // bytes
    constexpr int BUF = 2 * PLANE_A + 2 * PLANE_B;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    ptile_of_block(bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int nkc = p.GC >> 5;                                   // k tiles per filter tap
    const int nk = p.KH * p.KW * nkc;

    // ---- staging state: this lane's rows (fixed for the whole k loop) ----
    const int srow = lane >> 2;                                  // row inside a 16-row piece
    long a_base[NA]; unsigned a_mask[NA]; int a_lc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + 4 * i) * 16 + srow;
        a_lc[i] = (lane & 3) ^ ((row >> 2) & 3);                  // logical 16-byte piece this lane fetches (its LDS slot is lane & 3)
        const int m = m0 + row;
        a_mask[i] = 0u; a_base[i] = 0;
        if (m < p.M) {
            const int ox = m % p.OW; const int t = m / p.OW; const int oy = t % p.OH; const int img = t / p.OH;
            const int u = oy + p.pad, v = ox + p.pad;
            a_base[i] = (((long)img * p.GH + u) * p.GW + v) * p.GC + a_lc[i] * 8;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw) {
                    const int ty = u - kh, tx = v - kw;
                    if (ty >= 0 && tx >= 0 && ty < p.GH && tx < p.GW) a_mask[i] |= 1u << (kh * p.KW + kw);
                }
        }
    }
    long b_off[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wave + 4 * i) * 16 + srow;
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        b_off[i] = n < p.N ? (long)n * p.K + lc * 8 : -1;
    }
    const unsigned short* zero = g_zero_page;

    auto stage = [&](int kt, int buf) {
        const int tap = kt / nkc, c0 = (kt - tap * nkc) << 5;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long koff = c0 - ((long)kh * p.GW + kw) * p.GC;
        const unsigned bit = 1u << tap;
        unsigned char* dst = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool on = (a_mask[i] & bit) != 0u;
            const long o = a_base[i] + koff;
            const unsigned short* gh = on ? p.a_hi + o : zero;
            const unsigned short* gl = on ? p.a_lo + o : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)gh, (lds_ptr_t)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)gl, (lds_ptr_t)(dst + PLANE_A + (wave + 4 * i) * 1024), 16, 0, 0);
        }
        const long kb = (long)kt * 32;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool on = b_off[i] >= 0;
            const unsigned short* gh = on ? p.b_hi + b_off[i] + kb : zero;
            const unsigned short* gl = on ? p.b_lo + b_off[i] + kb : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)gh, (lds_ptr_t)(dst + 2 * PLANE_A + (wave + 4 * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)gl, (lds_ptr_t)(dst + 2 * PLANE_A + PLANE_B + (wave + 4 * i) * 1024), 16, 0, 0);
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

    stage(0, 0);
    __syncthreads();                                             // with a DMA in flight the fence of __syncthreads() carries vmcnt(0): tile 0 has landed
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk && !(p.dbg & 1)) stage(kt + 1, cur ^ 1);
        const unsigned char* ab = smem + cur * BUF + a_row_off;
        const unsigned char* bb = smem + cur * BUF + 2 * PLANE_A + b_row_off;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int po = ((2 * kc + lhi) ^ sw) * 16;
            bf16x8_t ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8_t*>(ab + i * 2048 + po);
                al[i] = *reinterpret_cast<const bf16x8_t*>(ab + PLANE_A + i * 2048 + po);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8_t*>(bb + j * 2048 + po);
                bl[j] = *reinterpret_cast<const bf16x8_t*>(bb + PLANE_B + j * 2048 + po);
            }
            if (p.dbg & 2) {                                     // ablation: keep the fragment reads alive, skip the matrix pipe
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(ah[i]), "v"(al[i]));
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(bh[j]), "v"(bl[j]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        // vmcnt(0) + lgkmcnt(0) + barrier: the next tile has landed in every wave and every wave's reads of this one have completed.  (A raw
        // s_barrier is not a fence for the compiler: it sank this tile's second half of ds_reads below it.)  The MFMAs of the last k chunk
        // are register-only and may still be scheduled past the barrier, where they overlap the next tile's address arithmetic.
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (32 * NI) + j * 32 + l31;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= p.M) continue;
                float* dst = p.C + (long)row * p.ldc + col;
                float v = 1.0f * acc[i][j][r] + 0.f;                  // alpha = 1, no bias: the epilogue arithmetic of gemm_x3_kernel
                if (p.beta != 0.f) v += p.beta * *dst;
                *dst = v;
            }
        }
}

static int g_pdbg = 0;
static int g_planes = 1;         // ha2g_conv_planes_enable: 0 = callers keep the round-2 kernels (A/B switch, HA2G_PLANES=0)

}  // namespace

extern "C" {

void ha2g_conv_planes_enable(int on) { g_planes = on; }
void ha2g_conv_planes_debug(int bits) { g_pdbg = bits; }

// fp32 -> (hi, lo) bf16 planes of the same shape; n % 4 == 0, 16-byte aligned
int ha2g_f32_to_planes(const float* x, void* hi, void* lo, long n, void* stream) {
    HA2G_REQUIRE(n % 4 == 0, "f32_to_planes: n %% 4");
    if (n == 0) return 0;
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi, (unsigned short*)lo, n4);
    HA2G_CHECK_LAUNCH("f32_to_planes");
    return 0;
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

// 1 when ha2g_conv2d_dgrad_planes_f32 serves this geometry in the current arithmetic mode (split-bf16 data gradients on, planes enabled)
int ha2g_conv2d_dgrad_planes_supported(int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return g_planes && gemm_split_dgrad_enabled() && stride == 1 && KH == 3 && KW == 3 && pad == 1 && Cout % 32 == 0 && Cin % 64 == 0 && Cin >= 64;
}

// dx [N,H,W,Cin] = beta * dx + conv_transpose(dy, w) from the bf16 planes of dy [N,H,W,Cout] and of wt [Cin][KH][KW][Cout]
// (ha2g_conv2d_weight_ihwo_planes).  Bit-identical to ha2g_conv2d_dgrad_f32 in the default arithmetic mode on the fp32 tensors the planes
// were split from.
int ha2g_conv2d_dgrad_planes_f32(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, float beta, void* stream) {
    HA2G_REQUIRE(ha2g_conv2d_dgrad_planes_supported(Cin, Cout, KH, KW, stride, pad), "conv2d_dgrad_planes: unsupported geometry / mode");
    PConvP p{};
    p.a_hi = (const unsigned short*)dy_hi; p.a_lo = (const unsigned short*)dy_lo;
    p.b_hi = (const unsigned short*)wt_hi; p.b_lo = (const unsigned short*)wt_lo;
    p.C = dx; p.ldc = Cin; p.beta = beta;
    p.M = N * H * W; p.N = Cin; p.K = KH * KW * Cout;
    p.GH = H; p.GW = W; p.GC = Cout; p.OH = H; p.OW = W; p.KH = KH; p.KW = KW; p.pad = pad;
    p.dbg = g_pdbg;
    if (p.M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (Cin % 128 == 0) {
        dim3 grid(ceil_div(p.M, 128), Cin / 128);
        hipLaunchKernelGGL((pconv_dgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(ceil_div(p.M, 256), Cin / 64);
        hipLaunchKernelGGL((pconv_dgrad_kernel<256, 64, 4, 1>), grid, dim3(256), 0, st, p);
    }
    HA2G_CHECK_LAUNCH("conv2d_dgrad_planes");
    return 0;
}

}  // extern "C"
