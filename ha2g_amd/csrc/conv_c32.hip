// Direct 3x3 / stride 1 / pad 1 convolution for 32 -> 32 channels (the six layer1 convolutions of the SE-ResNet34 and their
// data gradients: 19 % of the tower's flops at the largest spatial size, 128 x 70).  With N = 32 output channels the
// implicit-GEMM kernel re-stages every input value 9 times through the loader for only 32 columns of reuse and stays at
// ~65 TFLOP/s; here
//   * a tile is 256 CONSECUTIVE pixels of one image (8 MFMA row blocks, one per wave), whose input patch -- the 4-5 image rows
//     they span plus one halo row above and below, full width plus a zero column left and right -- is ONE contiguous range
//     of the NHWC tensor: it is copied to LDS once (16-byte loads/stores, pixel stride 36 floats => conflict-free b128
//     fragment reads) and every tap of every pixel reads it there; padding is zeros in LDS, so the inner loop has no masks;
//   * the whole 32 x 288 filter lives in REGISTERS (144 per lane) in MFMA B-fragment order, loaded once per workgroup;
//     workgroups are persistent (one 8-wave workgroup per CU) and loop over tiles with the patch DOUBLE-BUFFERED: the next
//     tile's patch is fetched one 16-byte slot per thread per filter tap while the current tile is on the matrix cores;
//   * inner loop per (row block, tap): 4 ds_read_b128 -> 16 v_mfma_f32_32x32x2_f32.  One 16-byte read feeds 4 MFMAs AND the k order
//     is the implicit GEMM's (k = tap * 32 + channel ascending, lanes 0-31 the even k of a pair, lanes 32-63 the odd one): the patch is
//     stored with the channels of every 8-group interleaved as [0 2 4 6 | 1 3 5 7], so the 16 bytes a lane of half h reads are channels
//     8q + h, 8q + 2 + h, 8q + 4 + h, 8q + 6 + h and MFMA j consumes the pair (8q + 2j, 8q + 2j + 1).  Every accumulator therefore runs the
//     same MFMA chain on the same operands as gemm.hip's A_IM kernel: the two kernels are BIT-IDENTICAL (tests/test_gpu_kernels.py), which is
//     what lets the forward pass use this one by default without touching the parity fixtures.
// The data gradient is the same kernel on dy with the [ci][kh][kw][co] weight image and the taps flipped.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CH = 32, CS = 36, TP = 256, NT = 512;

struct Tile { int img, p0, r0, rows; };

__device__ __forceinline__ Tile tile_of(long tile, int tpi, int HW, int W) {
    Tile t;
    t.img = (int)(tile / tpi); t.p0 = (int)(tile % tpi) * TP;
    const int pend = min(t.p0 + TP, HW);
    t.r0 = t.p0 / W;
    t.rows = (pend - 1) / W - t.r0 + 3;                                  // + halo row above and below
    return t;
}

// Cursor over one thread's 8 patch slots of a tile (slot t = 16 bytes: padded pixel (tid >> 3) + 64 t of the patch, channel
// quad c4 = a permutation of tid & 7).  Successive slots are 64 padded pixels apart, i.e. at most one row wrap (W + 2 > 64 is checked by the host):
// no divisions or multiplications per slot.
struct SlotCursor {
    int pr, px;          // patch row / padded column of the current slot
    long g;              // element offset of the source pixel's channel quad in x (valid only when the slot is inside the image)
    int lo;              // LDS float offset of the slot (fp32 patch)
    int pp;              // padded pixel index pr * (W + 2) + px (bf16-plane patch)
    __device__ __forceinline__ void init(const Tile& t, int tid, int H, int W, int HW) {
        const int PW = W + 2, pi = tid >> 3, c4 = ((tid >> 2) & 1) | ((tid & 3) << 1);   // quad parity = lane bit 2 (see interleave_slot)
        pr = pi / PW; px = pi - pr * PW;
        g = ((long)t.img * HW + (long)(t.r0 - 1 + pr) * W + (px - 1)) * CH + 4 * c4;
        lo = (pr * PW + px) * CS + 4 * c4;
        pp = pr * PW + px;
    }
    __device__ __forceinline__ void advance(int W) {
        const int PW = W + 2;
        px += 64; g += 64 * CH; lo += 64 * CS; pp += 64;
        if (px >= PW) { px -= PW; ++pr; g -= 2 * CH; }                   // the LDS offset is linear in the padded pixel index
    }
    // loads the slot (zeros outside the image); returns its LDS offset or -1 past the tile's patch rows
    __device__ __forceinline__ int load(const float* __restrict__ x, const Tile& t, int H, int W, float4& v) const {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr >= t.rows) return -1;
        const int gy = t.r0 - 1 + pr;
        if (gy >= 0 && gy < H && px >= 1 && px <= W) v = *reinterpret_cast<const float4*>(x + g);
        return lo;
    }
};

// Channel interleave of the patch (see header).  Lanes l and l ^ 4 hold the two channel quads 8q..8q+3 / 8q+4..8q+7 of one pixel (slot
// mapping in SlotCursor::init: the quad's parity is bit 2 of the lane) and must store [0 2 4 6] / [1 3 5 7].  Four DPP moves do it with
// no selects and no LDS traffic: the bank mask (groups of 4 lanes) writes the neighbour's value only into the lanes of one parity, the
// others keep their own value, so the patch is still written with one 16-byte store per slot.  Called by all lanes.
__device__ __forceinline__ float dpp_keep(float keep, float src, bool from_above) {
    // from_above: even-parity lanes (banks 0, 2) take src of lane + 4 (row_shl:4); else odd-parity lanes (banks 1, 3) take src of lane - 4
    return from_above ? __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x104, 0xF, 0x5, false))
                      : __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x114, 0xF, 0xA, false));
}
__device__ __forceinline__ float4 interleave_slot(const float4& v) {
    return make_float4(dpp_keep(v.x, v.y, false),      // even: ch0 (own x)        odd: ch1 (neighbour's y)
                       dpp_keep(v.z, v.w, false),      // even: ch2 (own z)        odd: ch3 (neighbour's w)
                       dpp_keep(v.y, v.x, true),       // even: ch4 (neighbour's x) odd: ch5 (own y)
                       dpp_keep(v.w, v.z, true));      // even: ch6 (neighbour's z) odd: ch7 (own w)
}
__device__ __forceinline__ void store_slot(float* lds, int off, const float4& v) { *reinterpret_cast<float4*>(lds + off) = v; }

// 512 threads = 8 waves, one 32-pixel row block each; persistent over tiles; the patch is double-buffered in LDS and the next
// tile's patch is fetched one slot per filter tap while the current tile is on the matrix cores.
__global__ __launch_bounds__(NT, 1) void conv3x3_c32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, int N, int H, int W, int flip, int act,
                                                            float beta, int patch_floats, unsigned row_magic) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 x [rows_max][W + 2][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = H * W, PW = W + 2;
    const int tpi = (HW + TP - 1) / TP;
    const long tiles = (long)N * tpi;

    // ---- filter -> registers: breg[t][q] holds w[n = l31][tap t][channels 8q + lhi, 8q + 2 + lhi, 8q + 4 + lhi, 8q + 6 + lhi] ----
    float4 breg[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f0 = *reinterpret_cast<const float4*>(w + ((long)l31 * 9 + tt) * CH + 8 * q);
            const float4 f1 = *reinterpret_cast<const float4*>(w + ((long)l31 * 9 + tt) * CH + 8 * q + 4);
            breg[t][q] = lhi ? make_float4(f0.y, f0.w, f1.y, f1.w) : make_float4(f0.x, f0.z, f1.x, f1.z);
        }
    }

    long tile = blockIdx.x;
    if (tile >= tiles) return;
    Tile cur = tile_of(tile, tpi, HW, W);
    // prologue: first patch straight into buffer 0
    {
        SlotCursor c; c.init(cur, tid, H, W, HW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v;
            const int off = c.load(x, cur, H, W, v);
            v = interleave_slot(v);
            if (off >= 0) store_slot(lds, off, v);
            c.advance(W);
        }
    }
    __syncthreads();
    int buf = 0;
    for (;; ) {
        const long ntile = tile + gridDim.x;
        const bool has_next = ntile < tiles;
        Tile nxt = cur;
        if (has_next) nxt = tile_of(ntile, tpi, HW, W);
        const float* patch = lds + buf * patch_floats;
        float* npatch = lds + (buf ^ 1) * patch_floats;

        int p = cur.p0 + 32 * wave + l31;
        if (p >= HW) p = HW - 1;                                         // clamp: stays inside the patch, result discarded
        const int py = p / W, px = p - py * W;
        const int a0 = ((py - cur.r0) * PW + px) * CS + 4 * lhi;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // next tile's patch: slot t is fetched at tap t and stored to the other LDS buffer DEPTH taps later (global latency is
        // ~2 taps of MFMA time); the last DEPTH slots are stored after the tap loop
        constexpr int DEPTH = 3;
        SlotCursor nc; nc.init(nxt, tid, H, W, HW);
        float4 nv[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) nv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        int noff[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) noff[i] = -1;
        // fragment reads run one (tap, channel-group) step ahead of the MFMAs that consume them
        float4 av = *reinterpret_cast<const float4*>(patch + a0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t >= DEPTH) {                                              // block-uniform condition: every lane takes part in the swap
                const float4 sv = interleave_slot(nv[t % DEPTH]);
                if (noff[t % DEPTH] >= 0) store_slot(npatch, noff[t % DEPTH], sv);
            }
            noff[t % DEPTH] = -1;
            if (has_next && t < 8 && !(act & 0x20)) { noff[t % DEPTH] = nc.load(x, nxt, H, W, nv[t % DEPTH]); nc.advance(W); }   /* interleaved at store time */
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 cur_a = av;
                const int step = 4 * t + q + 1;                            // next step's fragment
                if (step < 36) {
                    const int nt = step / 4, nq = step % 4;
                    av = *reinterpret_cast<const float4*>(patch + a0 + ((nt / 3) * PW + (nt % 3)) * CS + 8 * nq);
                }
                const float* pa = &cur_a.x; const float* pb = &breg[t][q].x;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[j], pb[j], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const float4 sv = interleave_slot(nv[i]);
            if (noff[i] >= 0) store_slot(npatch, noff[i], sv);
        }
        // ---- epilogue: C/D layout row = (r&3) + 8*(r>>2) + 4*lhi (pixel), col = l31 (channel) ----
        float* yout = y + (long)cur.img * HW * CH;
        const int pb0 = cur.p0 + 32 * wave;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = pb0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (pp >= HW || (act & 0x10)) continue;
            float v = acc[r];
            float* d = yout + (long)pp * CH + l31;
            if (beta != 0.f) v += beta * *d;
            if ((act & 15) == 1) v = fmaxf(v, 0.f);
            *d = v;
        }
        if (!has_next) break;
        // next patch complete, current patch no longer read.  LDS-only barrier: __syncthreads() would also wait for this tile's
        // output stores to drain (s_waitcnt vmcnt(0)), ~1-2 us of idle matrix cores per tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        tile = ntile; cur = nxt; buf ^= 1;
    }
}


// ---- split-bf16 variant (data gradient), C = 32 and C = 64 ----------------------------------------------------------------
// Same tiling; the patch is stored as two bf16 planes (hi = bf16(x), lo = bf16(x - hi); pixel stride C + 8 bf16 => conflict-free
// b128 fragment reads), the filter as hi / lo B fragments in registers, and every (pixel block, tap, 16-channel chunk) is
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_32x32x16_bf16 (8 passes each instead of 8 fp32 MFMAs of 16 passes).
//   C = 32: 8 waves, one 32-pixel block each, all 32 output channels (144 filter registers per lane);
//   C = 64 (template path, NOT dispatched): 4 waves, two pixel blocks each, 32 of the 64 output channels per workgroup (288
//           filter registers per lane, a tile is visited twice) -- measured 141 us = no faster than the split implicit GEMM.
// Single-buffered patch (two planes: 80.6 KB at 70-wide maps / C = 32, 117 KB at 35-wide maps / C = 64).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// NP = 3 (round 4, fp32-class): three pieces, six MFMAs per product; the filter's three piece fragments are 216 registers per lane, so the
// C = 32 kernel then runs FOUR waves (one per SIMD, 512 registers each) with two pixel blocks per wave instead of eight waves with one.
template <int C, int NP = 2> struct X3Geo {
    static constexpr int THREADS = (C == 32 && NP < 3) ? 512 : 256;      // 8 waves x 1 block, or 4 waves x 2 blocks of 32 pixels
    static constexpr int MB = (C == 32 && NP < 3) ? 1 : 2;               // pixel blocks per wave
    static constexpr int NH = C / 32;                        // 32-channel output halves
    static constexpr int CHUNKS = C / 16;
    static constexpr int CSX = C + 8;                        // bf16 per pixel in a plane
    static constexpr int QP = C / 4;                         // float4 quads per pixel
    static constexpr int PSTEP = THREADS / QP;               // padded pixels between a thread's consecutive slots
};

template <int C, int NP>
__global__ __launch_bounds__((X3Geo<C, NP>::THREADS), 1) void conv3x3_x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                               float* __restrict__ y, int N, int H, int W, int flip,
                                                                               int act, float beta, int plane_elems) {
    using G = X3Geo<C, NP>;
    extern __shared__ __attribute__((aligned(16))) unsigned short pl[];  // [2 planes][rows_max][W + 2][CSX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = H * W, PW = W + 2;
    const int tpi = (HW + TP - 1) / TP;
    const long tiles = (long)N * tpi;
    const int nh = G::NH == 1 ? 0 : (blockIdx.x % G::NH);              // output-channel half of this workgroup
    const int nout = nh * 32 + l31;

    // filter -> registers: for tap t and 16-channel chunk c, lane (n = nout, half = lhi) holds channels 16c + 8*half .. +7 of row n
    bf16x8_t bq[NP][9][G::CHUNKS];                                       // piece q of the filter fragments
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
#pragma unroll
        for (int c = 0; c < G::CHUNKS; ++c) {
            const float* src = w + ((long)nout * 9 + tt) * C + 16 * c + 8 * lhi;
            const float4 u0 = *reinterpret_cast<const float4*>(src);
            const float4 u1 = *reinterpret_cast<const float4*>(src + 4);
            unsigned e0[NP], e1[NP], e2[NP], e3[NP];
            splitn_bf16<NP>(u0.x, u0.y, e0); splitn_bf16<NP>(u0.z, u0.w, e1); splitn_bf16<NP>(u1.x, u1.y, e2); splitn_bf16<NP>(u1.z, u1.w, e3);
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[q][t][c] = __builtin_bit_cast(bf16x8_t, make_uint4(e0[q], e1[q], e2[q], e3[q]));
        }
    }
    unsigned short* phi = pl;
    for (long tile = blockIdx.x / G::NH; tile < tiles; tile += gridDim.x / G::NH) {
        const Tile cur = tile_of(tile, tpi, HW, W);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // previous tile's fragment reads are done
        {
            // patch fill: slot = (padded pixel, 4-channel quad); a thread's slots are PSTEP padded pixels apart
            const int c4 = tid % G::QP;
            int pp = tid / G::QP;
            int pr = pp / PW, px = pp - pr * PW;
            long g = ((long)cur.img * HW + (long)(cur.r0 - 1 + pr) * W + (px - 1)) * C + 4 * c4;
            while (pr < cur.rows) {
                constexpr int NS = 8;
                float4 v[NS]; int off[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    off[i] = -1;
                    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pr < cur.rows) {
                        const int gy = cur.r0 - 1 + pr;
                        if (gy >= 0 && gy < H && px >= 1 && px <= W) v[i] = *reinterpret_cast<const float4*>(x + g);
                        off[i] = pp * G::CSX + 4 * c4;
                    }
                    pp += G::PSTEP; px += G::PSTEP; g += (long)G::PSTEP * C;
                    while (px >= PW) { px -= PW; ++pr; g -= 2 * C; }
                }
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (off[i] >= 0) {
                        unsigned e0[NP], e1[NP];
                        splitn_bf16<NP>(v[i].x, v[i].y, e0); splitn_bf16<NP>(v[i].z, v[i].w, e1);
#pragma unroll
                        for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(phi + q * plane_elems + off[i]) = make_uint2(e0[q], e1[q]);
                    }
            }
        }
        __syncthreads();
        int a0[G::MB];
#pragma unroll
        for (int b = 0; b < G::MB; ++b) {
            int p = cur.p0 + 32 * (G::MB * wave + b) + l31;
            if (p >= HW) p = HW - 1;                                     // clamp: stays inside the patch, result discarded
            const int py = p / W, px = p - py * W;
            a0[b] = ((py - cur.r0) * PW + px) * G::CSX + 8 * lhi;
        }
        f32x16 acc[G::MB];
#pragma unroll
        for (int b = 0; b < G::MB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * PW + (t % 3)) * G::CSX;
#pragma unroll
            for (int c = 0; c < G::CHUNKS; ++c)
#pragma unroll
                for (int b = 0; b < G::MB; ++b) {
                    bf16x8_t aq[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) aq[q] = *reinterpret_cast<const bf16x8_t*>(phi + q * plane_elems + a0[b] + toff + 16 * c);
                    if constexpr (NP == 3) {                              // six products, smallest first (conv_planes.hip)
                        constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                        for (int u = 0; u < 6; ++u) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[QA[u]], bq[QB[u]][t][c], acc[b], 0, 0, 0);
                    } else {
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[1], bq[0][t][c], acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], bq[1][t][c], acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[0], bq[0][t][c], acc[b], 0, 0, 0);
                    }
                }
        }
        float* yout = y + (long)cur.img * HW * C;
#pragma unroll
        for (int b = 0; b < G::MB; ++b) {
            const int pb0 = cur.p0 + 32 * (G::MB * wave + b);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = pb0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (pp >= HW) continue;
                float v = acc[b][r];
                float* d = yout + (long)pp * C + nout;
                if (beta != 0.f) v += beta * *d;
                if ((act & 15) == 1) v = fmaxf(v, 0.f);
                *d = v;
            }
        }
    }
}


// ---- PREFETCHING form of conv3x3_x3_kernel<32, 3> (round 4).  The first form fills its single patch buffer (global loads -> split -> LDS), then
// runs the tile's 216 MFMAs per wave, then stores, tile after tile: with one wave per SIMD nothing hides the ~2 us of load latency and the ~1 us of
// split + LDS writes of each of a workgroup's 17.5 tiles (262 us per launch for 51 us of matrix work).  Here the NEXT tile's global loads are issued
// into registers (16 float4 per thread: the whole 7-row patch) BEFORE the current tile's MFMAs and stores, and are split / written to LDS after the
// barrier that retires the current patch -- the load latency runs under the matrix phase.  Same values, same MFMA order: bit-identical results. ----
__global__ __launch_bounds__(256, 1) void conv3x3_x3p_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N, int H,
                                                             int W, int flip, int act, float beta, int plane_elems) {
    using G = X3Geo<32, 3>;
    constexpr int NP = 3, C = 32, NSF = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned short pl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int HW = H * W, PW = W + 2;
    const int tpi = (HW + TP - 1) / TP;
    const long tiles = (long)N * tpi;
    const int nout = l31;

    bf16x8_t bq[NP][9][G::CHUNKS];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
#pragma unroll
        for (int c = 0; c < G::CHUNKS; ++c) {
            const float* src = w + ((long)nout * 9 + tt) * C + 16 * c + 8 * lhi;
            const float4 u0 = *reinterpret_cast<const float4*>(src);
            const float4 u1 = *reinterpret_cast<const float4*>(src + 4);
            unsigned e0[NP], e1[NP], e2[NP], e3[NP];
            splitn_bf16<NP>(u0.x, u0.y, e0); splitn_bf16<NP>(u0.z, u0.w, e1); splitn_bf16<NP>(u1.x, u1.y, e2); splitn_bf16<NP>(u1.z, u1.w, e3);
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[q][t][c] = __builtin_bit_cast(bf16x8_t, make_uint4(e0[q], e1[q], e2[q], e3[q]));
        }
    }
    unsigned short* phi = pl;
    float4 pv[NSF]; int poff[NSF];                           // the NEXT tile's patch slots of this thread, in flight / waiting for the barrier
    const int c4 = tid % G::QP, pp0 = tid / G::QP;
    auto issue = [&](const Tile& t) {
        int pp = pp0;
        int pr = pp / PW, px = pp - pr * PW;
        long g = ((long)t.img * HW + (long)(t.r0 - 1 + pr) * W + (px - 1)) * C + 4 * c4;
#pragma unroll
        for (int i = 0; i < NSF; ++i) {
            poff[i] = -1;
            pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr < t.rows) {
                const int gy = t.r0 - 1 + pr;
                if (gy >= 0 && gy < H && px >= 1 && px <= W) pv[i] = *reinterpret_cast<const float4*>(x + g);
                poff[i] = pp * G::CSX + 4 * c4;
            }
            pp += G::PSTEP; px += G::PSTEP; g += (long)G::PSTEP * C;
            if (px >= PW) { px -= PW; ++pr; g -= 2 * C; }        // PSTEP < PW (host check): at most one row wrap per slot
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NSF; ++i)
            if (poff[i] >= 0) {
                unsigned e0[NP], e1[NP];
                splitn_bf16<NP>(pv[i].x, pv[i].y, e0); splitn_bf16<NP>(pv[i].z, pv[i].w, e1);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(phi + q * plane_elems + poff[i]) = make_uint2(e0[q], e1[q]);
            }
    };
    const int dbg = act >> 8;                               // timing ablations (ha2g_conv_c32_prefetch(1 | bits << 1)): 1 no MFMA, 2 no LDS fill, 4 no stores, 8 no loads
    long tile = blockIdx.x;
    if (tile >= tiles) return;
    Tile cur = tile_of(tile, tpi, HW, W);
    issue(cur);
    commit();
    __syncthreads();
    for (;;) {
        const long ntile = tile + gridDim.x;
        const bool has_next = ntile < tiles;
        Tile nxt = cur;
        if (has_next) { nxt = tile_of(ntile, tpi, HW, W); if (!(dbg & 8)) issue(nxt); }
        int a0[G::MB];
#pragma unroll
        for (int b = 0; b < G::MB; ++b) {
            int p = cur.p0 + 32 * (G::MB * wave + b) + l31;
            if (p >= HW) p = HW - 1;                                     // clamp: stays inside the patch, result discarded
            const int py = p / W, px = p - py * W;
            a0[b] = ((py - cur.r0) * PW + px) * G::CSX + 8 * lhi;
        }
        f32x16 acc[G::MB];
#pragma unroll
        for (int b = 0; b < G::MB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        if (!(dbg & 1))
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * PW + (t % 3)) * G::CSX;
#pragma unroll
            for (int c = 0; c < G::CHUNKS; ++c) {
                // the two pixel blocks' chains INTERLEAVED: six back-to-back MFMAs on one accumulator each wait out the previous one's latency
                // (the first form ran them block after block); alternating the blocks keeps two independent chains in the pipe.  Per accumulator
                // the order of the products is unchanged.
                bf16x8_t aq[G::MB][NP];
#pragma unroll
                for (int b = 0; b < G::MB; ++b)
#pragma unroll
                    for (int q = 0; q < NP; ++q) aq[b][q] = *reinterpret_cast<const bf16x8_t*>(phi + q * plane_elems + a0[b] + toff + 16 * c);
                constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int b = 0; b < G::MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[b][QA[u]], bq[QB[u]][t][c], acc[b], 0, 0, 0);
            }
        }
        float* yout = y + (long)cur.img * HW * C;
        if (!(dbg & 4))
#pragma unroll
        for (int b = 0; b < G::MB; ++b) {
            const int pb0 = cur.p0 + 32 * (G::MB * wave + b);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = pb0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (pp >= HW) continue;
                float v = acc[b][r];
                float* d = yout + (long)pp * C + nout;
                if (beta != 0.f) v += beta * *d;
                if ((act & 15) == 1) v = fmaxf(v, 0.f);
                *d = v;
            }
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading the current patch
        if (!(dbg & 2)) commit();                                         // the prefetched tile: split + LDS writes (its loads landed under the MFMAs)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // patch complete (LDS only: the output stores keep draining)
        tile = ntile; cur = nxt;
    }
}


// ---- ANTI-PHASE form (round 4): conv3x3_c32pp_kernel.  tools/c32_ablate.py on the prefetching form: matrix phase 94 us, output stores 36, global loads 30,
// split + LDS fill 26 -- and they ADD UP (182 us): with one wave per SIMD nothing runs beside the matrix pipe.  Here a workgroup is two groups of four
// waves (wave w and w + 4 share a SIMD) that work on their OWN 128-pixel tiles in opposite phases: while one group's 216 MFMAs per wave run, the other
// stores its finished tile, splits / writes its next patch from registers and issues the loads of the tile after that; one workgroup barrier per phase.
// To fit two waves per SIMD a wave owns 64 pixels x 16 output channels on v_mfma_f32_16x16x32_bf16 (108 filter registers instead of 216); the weight
// fragment is the first operand (D = W X^T), so a lane holds four consecutive output channels of one pixel: one 16-byte store per 16 x 16 tile.
// Patch: [piece][padded pixel][32 channels] bf16, 64 bytes per pixel, 16-byte slots XOR-ed with 2 ((pixel >> 2) & 1) and fragment columns permuted
// (pxo below: conflict-free ds_read_b128 for any tap shift).  Products and their order per accumulator = the other three-piece kernels (six per k block, smallest first). ----
constexpr int TPH = 128;                                     // pixels of a group's tile
__device__ __forceinline__ int fast_div_c32(int m, int d, unsigned mg) {
    int q = (int)__umulhi((unsigned)m, mg);                   // mg = ceil(2^32 / d), d >= 2: the quotient or one more
    if (q * d > m) --q;
    return q;
}
__global__ __launch_bounds__(512, 1) void conv3x3_c32pp_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N, int H,
                                                               int W, int flip, int act, float beta, int plane_elems, unsigned mgW,
                                                               const float* __restrict__ rsd, const unsigned* __restrict__ rbits) {
    constexpr int NP = 3, C = 32, NSF = 12;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned short ppl[];          // [group][piece][plane_elems]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // groups by wave >> 2: wave w and w + 4 share a SIMD, so each SIMD holds one wave of EACH group and its matrix pipe alternates between them
    // (A/B bit 14 of act: groups by wave parity = both waves of a SIMD in the same phase: 177 vs 145 us, tools/c32_ablate.py)
    const bool by_half = !((act >> 14) & 1);
    const int grp = by_half ? (wave >> 2) : (wave & 1), w4 = by_half ? (wave & 3) : (wave >> 1), ph = w4 & 1, chh = w4 >> 1;
    const int l15 = lane & 15, kp = lane >> 4;
    // Fragment column l15 -> pixel offset inside its 16-pixel tile.  A ds_read_b128 is served in four groups of 16 lanes, {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32: each group holds columns {0-3, 12-15} of one k slot and columns {4-11} of its NEIGHBOUR slot (k ^ 1).
    // With columns {0-3, 12-15} on pixels 0-7 and columns 4-11 on pixels 8-15 a group reads slot k of eight consecutive pixels and slot k ^ 1 of the
    // next eight; the patch's slot swizzle  s ^ 2 ((pixel >> 2) & 1)  then puts the four pixels of every residue class mod 4 (the ones that share
    // banks: a pixel is 64 bytes) on the four different slots {k, k ^ 2, k ^ 1, k ^ 3} -- conflict-free for ANY tap shift of the base pixel, as
    // long as the 16 pixels are consecutive in the padded patch (a tile that wraps around an image row keeps a 2-way conflict on a few lanes).
    // The round-4 swizzle  s ^ ((pixel >> 2) & 3)  with columns = pixels in order measured 44 % of the LDS-active cycles as conflicts
    // (profiles/r04_pmc_c32.txt): no XOR of the slot by a function of pixel >> 2 alone is conflict-free under that lane grouping.
    const int pxo = l15 < 4 ? l15 : (l15 < 12 ? l15 + 4 : l15 - 8);
    const int gt = w4 * 64 + lane;                           // thread index inside the group
    const int HW = H * W, PW = W + 2;
    const int tpi = (HW + TPH - 1) / TPH;
    const long tiles = (long)N * tpi;
    unsigned short* const pg = ppl + (long)grp * NP * plane_elems;

    // filter: lane (channel chh * 16 + l15, k piece kp) holds input channels 8 kp .. + 7 of every tap -- pieces 0 / 1 in registers (72), piece 2 (one
    // of the six products per tap) in LDS behind the patches: 36 registers less, which is what lets FOUR accumulator chains run interleaved (a
    // dependent v_mfma_f32_16x16x32_bf16 waits ~60 cycles: with two chains the matrix phase ran at 29 cycles per MFMA, tools/c32_ablate.py)
    bf16x8_t bq[2][9];
    unsigned short* const fl2 = ppl + 2L * NP * plane_elems;                        // [channel half][tap][lane][8]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
        const float* src = w + ((long)(chh * 16 + l15) * 9 + tt) * C + 8 * kp;
        const float4 u0 = *reinterpret_cast<const float4*>(src);
        const float4 u1 = *reinterpret_cast<const float4*>(src + 4);
        unsigned e0[NP], e1[NP], e2[NP], e3[NP];
        splitn_bf16<NP>(u0.x, u0.y, e0); splitn_bf16<NP>(u0.z, u0.w, e1); splitn_bf16<NP>(u1.x, u1.y, e2); splitn_bf16<NP>(u1.z, u1.w, e3);
#pragma unroll
        for (int q = 0; q < 2; ++q) bq[q][t] = __builtin_bit_cast(bf16x8_t, make_uint4(e0[q], e1[q], e2[q], e3[q]));
        if (grp == 0 && ph == 0) *reinterpret_cast<uint4*>(fl2 + ((chh * 9 + t) * 64 + lane) * 8) = make_uint4(e0[2], e1[2], e2[2], e3[2]);
    }

    auto tile_at = [&](long t) {
        Tile r;
        r.img = (int)(t / tpi); r.p0 = (int)(t - (long)r.img * tpi) * TPH;
        const int pend = min(r.p0 + TPH, HW);
        r.r0 = fast_div_c32(r.p0, W, mgW);
        r.rows = fast_div_c32(pend - 1, W, mgW) - r.r0 + 3;
        return r;
    };
    // patch slots of this thread: padded pixel (source numbering, rows of PW pixels) pp0 + 32 i, channel quad c4 (4 fp32 = 8 bytes of a piece plane)
    const int c4 = gt & 7, pp0 = gt >> 3;
    const int pr0 = pp0 / PW, px0 = pp0 - pr0 * PW;
    float4 pv[NSF]; unsigned pmask = 0u;                     // the next tile's slots in flight; bit i = slot i lies inside the patch
    auto issue = [&](const Tile& t) {
        int pr = pr0, px = px0;
        long g = ((long)t.img * HW + (long)(t.r0 - 1 + pr) * W + (px - 1)) * C + 4 * c4;
        pmask = 0u;
#pragma unroll
        for (int i = 0; i < NSF; ++i) {
            pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pr < t.rows) {
                const int gy = t.r0 - 1 + pr;
                if (gy >= 0 && gy < H && px >= 1 && px <= W) pv[i] = *reinterpret_cast<const float4*>(x + g);
                pmask |= 1u << i;
            }
            px += 32; g += 32L * C;
            if (px >= PW) { px -= PW; ++pr; g -= 2 * C; }        // 32 < PW (host check): at most one row wrap per slot
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NSF; ++i)
            if (pmask & (1u << i)) {
                const int pp = pp0 + 32 * i;
                const int off = pp * 32 + ((((c4 >> 1) ^ ((pp >> 1) & 2)) << 3) | ((c4 & 1) << 2));      // element offset: swizzled 16-byte slot + half
                unsigned e0[NP], e1[NP];
                splitn_bf16<NP>(pv[i].x, pv[i].y, e0); splitn_bf16<NP>(pv[i].z, pv[i].w, e1);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(pg + q * plane_elems + off) = make_uint2(e0[q], e1[q]);
            }
    };
    f32x4_t acc[4];
    auto matrix = [&](const Tile& t) {
        int ppb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int p = t.p0 + ph * 64 + i * 16 + pxo;
            if (p >= HW) p = HW - 1;                         // clamp: stays inside the patch, result discarded
            const int py = fast_div_c32(p, W, mgW), px = p - py * W;
            ppb[i] = (py - t.r0) * PW + px;
            acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};          // (pixel piece, filter piece), smallest products first
        // 18 blocks (tap, pair of row tiles) of 12 MFMAs over two fragment sets.  Round 4 issued a set's six reads + the tap's filter piece in one
        // burst BETWEEN two blocks: ~25 instructions during which the matrix pipe drained, and the filter-piece read at the head of every tap was
        // waited for with lgkmcnt(0) -- one fully exposed LDS round trip per tap (ISA: profiles/r05_c32pp_isa_notes.txt; 29-31 cycles per MFMA where
        // a wave sustains 16).  Now every read is issued INSIDE a block, two or three instructions behind an MFMA pair, as soon as its destination
        // registers retire: a set's piece 2 is last read by the block's first MFMA pair, piece 1 by the fourth, piece 0 by the sixth; the filter piece
        // by the second pair.  Reads are issued in the order their consumers need them (piece 2, then 0, then 1: LDS returns in order, so every wait
        // is a counted one) one to two blocks ahead of their use.
        bf16x8_t aq[2][2][NP];                               // [set = pair][row tile of the pair][piece]
        int eo[2][2];                                        // element offset of the set's two pixels at the tap being fetched
        bf16x8_t b2;
        auto addr = [&](int tp, int set) {
            const int toff = (tp / 3) * PW + (tp % 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pp = ppb[2 * set + j] + toff;
                eo[set][j] = pp * 32 + ((kp ^ ((pp >> 1) & 2)) << 3);
            }
        };
        auto rd = [&](int set, int q) {
#pragma unroll
            for (int j = 0; j < 2; ++j) aq[set][j][q] = *reinterpret_cast<const bf16x8_t*>(pg + q * plane_elems + eo[set][j]);
        };
        auto rdb2 = [&](int tp) { b2 = *reinterpret_cast<const bf16x8_t*>(fl2 + ((chh * 9 + tp) * 64 + lane) * 8); };
        addr(0, 0); rd(0, 2); rd(0, 0); rd(0, 1);
        addr(0, 1); rd(1, 2); rd(1, 0); rd(1, 1);
        rdb2(0);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
            for (int pr_ = 0; pr_ < 2; ++pr_) {
                const int other = 1 - pr_;
                const int tp_other = pr_ == 0 ? tp : tp + 1;                        // the tap the other set is consumed at next
                auto mm = [&](int u) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[2 * pr_ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QB[u] == 2 ? b2 : bq[QB[u] & 1][tp], aq[pr_][j][QA[u]], acc[2 * pr_ + j], 0, 0, 0);
                };
                __builtin_amdgcn_sched_barrier(0);
                mm(0);                                                                // last readers of this set's piece 2
                if (tp + 1 < 9) addr(tp + 1, pr_);
                if (!(tp == 0 && pr_ == 0) && tp_other < 9) rd(other, 0);           // the other set's piece 0 retired with the previous block
                __builtin_amdgcn_sched_barrier(0);
                mm(1);                                                                // last readers of the filter's piece 2 (second block of the tap)
                if (tp + 1 < 9) rd(pr_, 2);
                __builtin_amdgcn_sched_barrier(0);
                mm(2);
                if (pr_ == 1 && tp + 1 < 9) rdb2(tp + 1);
                __builtin_amdgcn_sched_barrier(0);
                mm(3);                                                                // last readers of this set's piece 1
                if (tp + 1 < 9) rd(pr_, 1);
                __builtin_amdgcn_sched_barrier(0);
                mm(4);
                mm(5);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto store = [&](const Tile& t) {
        float* yout = y + (long)t.img * HW * C + chh * 16 + 4 * kp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = t.p0 + ph * 64 + i * 16 + pxo;
            if (p >= HW) continue;
            f32x4_t v = acc[i];
            f32x4_t* d = reinterpret_cast<f32x4_t*>(yout + (long)p * C);
            if (beta != 0.f) v += beta * *d;
            if (rsd != nullptr) {
                // masked residual (round 6; the block's identity shortcut under autograd, ResNetBlocks.py:34-36): y = conv + (out > 0 ? dout : 0) with the
                // decisions from the forward tail's bit words -- what "beta = 1 onto dres" added, without dres ever being written
                const long e = (yout - y) + (long)p * C;
                const f32x4_t dd = *reinterpret_cast<const f32x4_t*>(rsd + e);
                const long vi = e >> 2;
                const unsigned b = (rbits[vi >> 3] >> (4 * (int)(vi & 7))) & 15u;
                v[0] += (b & 1u) ? dd[0] : 0.f; v[1] += (b & 2u) ? dd[1] : 0.f; v[2] += (b & 4u) ? dd[2] : 0.f; v[3] += (b & 8u) ? dd[3] : 0.f;
            }
            if ((act & 15) == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            *d = v;
        }
    };

    // this group's tiles: 2 blockIdx.x + grp, then every 2 gridDim.x; both groups run the same number of rounds (the barriers are workgroup-wide)
    const long stride = 2L * gridDim.x;
    const long first = 2L * blockIdx.x + grp;
    const long n_rounds = (tiles - 2L * blockIdx.x + stride - 1) / stride;           // rounds of group 0 (>= those of group 1)
    Tile cur = tile_at(first < tiles ? first : 0), nxt = cur;
    bool v_cur = first < tiles, v_nxt = first + stride < tiles;
    if (v_cur) { issue(cur); commit(); }
    if (v_nxt) { nxt = tile_at(first + stride); issue(nxt); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");                            // half a round behind group 0
    const int dbg = act >> 8;                               // timing ablations (ha2g_conv_c32_prefetch bits 1-4): 1 no MFMA, 2 no LDS fill, 4 no stores, 8 no loads
    for (long k = 0; k < n_rounds; ++k) {
        if (v_cur && !(dbg & 1)) matrix(cur);                                         // MATRIX phase
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // OTHER phase: next patch from registers FIRST (its loads are the oldest outstanding memory operations: a counted vmcnt suffices -- behind the
        // stores the wait would be for the stores' write acknowledgements), then this tile's stores, then the loads of the tile after the next
        const Tile done = cur; const bool v_done = v_cur;
        cur = nxt; v_cur = v_nxt;
        if (v_cur && !(dbg & 2)) commit();
        if (v_done && !(dbg & 4)) store(done);
        const long t2 = first + (k + 2) * stride;
        v_nxt = t2 < tiles;
        if (v_nxt) { nxt = tile_at(t2); if (!(dbg & 8)) issue(nxt); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
}


// ---- direct weight gradient, 32 -> 32 channels, 3x3 / stride 1 / pad 1 (split-bf16) ---------------------------------------------------
// dW[co][tap][ci] = sum over pixels of dy[p][co] * x[p + tap][ci].  The implicit GEMM stages the im2col gather of x -- the same patch nine
// times, for only 32 rows of reuse -- and is bound by that staging (237 us at B = 128).  Here a tile is 128 consecutive pixels of one image:
// its x patch (rows + halo, zero padding) and its dy strip are stored ONCE as bf16 hi / lo planes [pixel][32 channels], and every fragment
// is a pair of transpose reads (ds_read_b64_tr_b16, see gemm.hip SPLIT = 3): the dy fragment (A: co x 16 pixels) is shared by the nine
// taps, the x fragment of tap t is the same read at a row offset of (kh * PW + kw) patch pixels.  Each wave owns 32 pixels of the tile and
// nine 32 x 32 accumulators (144 AGPRs) that live across all tiles of its persistent workgroup; at the end the four waves are added in
// wave order through LDS and the workgroup writes ONE partial [32][9][32]; gemm.hip's wide split-K reduce adds the partials in double.
constexpr int WG_TP = 128, WG_NT = 256;

// T = float: fp32 tensors, split here into hi / lo planes (three MFMAs per product); T = b16 (bf16-storage mode): the tensors ARE the hi plane,
// one MFMA per product, the lo planes stay unused.
// NP = 3 (round 4, fp32-class): three piece planes per operand (93 KB at 70-wide maps: one workgroup per CU), six MFMAs per product.
// PF (round 6, fp32 / three pieces only): the NEXT tile's x patch and dy strip are loaded into registers (12 + 4 sixteen-byte slots per thread, branch-free
// buffer loads: padding = an out-of-range offset, which returns zeros) BEFORE this tile's MFMA phase and split / written to LDS behind it -- the first form
// loaded, waited, split, wrote and only then computed: 2.6 us of exposed load latency against 1.4 us of MFMAs per tile.  The side queue is the saturated
// one during the tower's backward (10.7 of 12.1 ms, profiles/r06_*), and beside this kernel bn_bwd_apply ran 18 -> 107 us and the 32-channel data
// gradient 152 -> 273 us: its duration is main-queue time.  Same values, same summation order: bit-identical partials.
// TPX = pixels per tile: 128, or 256 in the prefetching form where the LDS holds it (70-wide maps: 142.5 KB) -- a 256-pixel tile spans 3.7 image rows + 2 halo
// rows (1.6x its pixels) where a 128-pixel tile spans 1.8 + 2 (2.4x): 21 % fewer bytes loaded per pixel
template <typename T, int NP = 2, bool PF = false, int TPX = 128>
__global__ __launch_bounds__(WG_NT, (NP == 3 ? 1 : 2)) void conv3x3_c32_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                                     float* __restrict__ part, int N, int H, int W, int plane_elems) {
    constexpr bool F32 = sizeof(T) == 4;
    static_assert(!PF || (F32 && NP == 3), "the prefetching form serves the fp32 three-piece mode");
    static_assert(TPX == 128 || (PF && TPX == 256), "256-pixel tiles: prefetching form only");
    extern __shared__ __attribute__((aligned(16))) unsigned short wpl[];        // [x pieces][dy pieces]
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5, g4 = lane >> 4, q16 = lane & 15;
    const int krow = 8 * (g4 >> 1) + (q16 >> 2), moff = 16 * (g4 & 1) + 4 * (q16 & 3);
    const int HW = H * W, PW = W + 2;
    const int tpi = (HW + TPX - 1) / TPX;
    const long tiles = (long)N * tpi;
    unsigned short* xh = wpl;                                                   // piece q of x at xh + q * plane_elems
    unsigned short* xl = wpl + plane_elems;
    unsigned short* dh = wpl + NP * plane_elems;                                // piece q of dy at dh + q * TPX * CH
    unsigned short* dl = dh + TPX * CH;
    auto tr = [](const unsigned short* ptr) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(__attribute__((address_space(3))) const unsigned short*)ptr);
    };
    auto frag = [&](const unsigned short* p0, const unsigned short* p1) {      // k rows krow..krow+3 (p0) and krow+4..krow+7 (p1)
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(tr(p0), tr(p1), 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // one tile's MFMA phase out of the LDS planes (both forms)
    auto mfma_tile = [&](const Tile& cur) {
#pragma unroll
        for (int ch = 0; ch < TPX / 64; ++ch) {
            const int pb = (TPX / 4) * wave + 16 * ch + krow;                           // this lane's two tile-local pixels: pb and pb + 4
            const bf16x8_t ah = frag(dh + pb * CH + moff, dh + (pb + 4) * CH + moff);
            bf16x8_t al = ah, a2 = ah;
            if constexpr (F32) al = frag(dl + pb * CH + moff, dl + (pb + 4) * CH + moff);
            if constexpr (F32 && NP == 3) a2 = frag(dl + TPX * CH + pb * CH + moff, dl + TPX * CH + (pb + 4) * CH + moff);
            int p0 = cur.p0 + pb, p1 = p0 + 4;
            if (p0 >= HW) p0 = HW - 1;                                           // clamp: stays inside the patch; dy is zero there
            if (p1 >= HW) p1 = HW - 1;
            const int y0 = p0 / W, y1 = p1 / W;
            const int r0 = ((y0 - cur.r0) * PW + (p0 - y0 * W)) * CH + moff;     // tap (0,0) = the pixel above-left (halo included)
            const int r1 = ((y1 - cur.r0) * PW + (p1 - y1 * W)) * CH + moff;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * PW + (t % 3)) * CH;
                const bf16x8_t bh = frag(xh + r0 + toff, xh + r1 + toff);
                if constexpr (F32 && NP == 3) {                                  // six products, smallest first
                    const bf16x8_t bl = frag(xl + r0 + toff, xl + r1 + toff);
                    const bf16x8_t b2 = frag(xl + plane_elems + r0 + toff, xl + plane_elems + r1 + toff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                } else if constexpr (F32) {
                    const bf16x8_t bl = frag(xl + r0 + toff, xl + r1 + toff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    };
    auto tile_geom = [&](long tile) {
        Tile t;
        t.img = (int)(tile / tpi); t.p0 = (int)(tile % tpi) * TPX;
        const int pend = min(t.p0 + TPX, HW);
        t.r0 = t.p0 / W;
        t.rows = (pend - 1) / W - t.r0 + 3;
        return t;
    };
    if constexpr (PF) {
        constexpr int XS = TPX == 256 ? 16 : 12, DS = TPX / 32;                  // x patch / dy strip slots per thread (host: rows_max * PW <= 32 XS)
        constexpr unsigned OOB = 0xfffffff0u;
        const int c4 = tid & 7;
        // slot i of this thread = padded pixel (tid >> 3) + 32 i of the patch: its patch row / byte offset relative to the patch origin (row r0 - 1,
        // column 0) depend on PW only
        int s_pr[XS], s_rel[XS];
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int pp = (tid >> 3) + 32 * i, pr = pp / PW, px = pp - pr * PW;
            s_pr[i] = pr;
            s_rel[i] = (px >= 1 && px <= W) ? ((pr * W + (px - 1)) * CH + 4 * c4) * 4 : -1;
        }
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long)N * HW * CH * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long)N * HW * CH * 4), 0x00020000);
        typedef float f4_t __attribute__((ext_vector_type(4)));
        f4_t xv[XS], dv[DS];
        auto issue = [&](const Tile& t) {
            const long base = ((long)t.img * HW + (long)(t.r0 - 1) * W) * CH * 4;       // may be negative for the first image's top halo: those slots are masked
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                const int gy = t.r0 - 1 + s_pr[i];
                const bool on = s_pr[i] < t.rows && gy >= 0 && gy < H && s_rel[i] >= 0;
                const unsigned off = on ? (unsigned)(base + s_rel[i]) : OOB;
                xv[i] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                const int p = t.p0 + (tid >> 3) + 32 * i;
                const unsigned off = p < HW ? (unsigned)((((long)t.img * HW + p) * CH + 4 * c4) * 4) : OOB;
                dv[i] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(dr, (int)off, 0, 0));
            }
        };
        auto commit = [&](const Tile& t) {
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                if (s_pr[i] < t.rows) {                                          // rows of the patch: pixels outside the image carry the zeros the load returned
                    const int off = ((tid >> 3) + 32 * i) * CH + 4 * c4;
                    unsigned e0[NP], e1[NP];
                    splitn_bf16<NP>(xv[i][0], xv[i][1], e0); splitn_bf16<NP>(xv[i][2], xv[i][3], e1);
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(xh + q * plane_elems + off) = make_uint2(e0[q], e1[q]);
                }
            }
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                const int o = ((tid >> 3) + 32 * i) * CH + 4 * c4;
                unsigned e0[NP], e1[NP];
                splitn_bf16<NP>(dv[i][0], dv[i][1], e0); splitn_bf16<NP>(dv[i][2], dv[i][3], e1);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dh + q * TPX * CH + o) = make_uint2(e0[q], e1[q]);
            }
        };
        long tile = blockIdx.x;
        Tile cur{}, nxt{};
        if (tile < tiles) { cur = tile_geom(tile); issue(cur); }
        for (; tile < tiles; tile += gridDim.x) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous tile's fragment reads are done
            commit(cur);                                                         // waits for this tile's loads (they flew under the previous MFMA phase)
            __syncthreads();
            const long next = tile + gridDim.x;
            if (next < tiles) { nxt = tile_geom(next); issue(nxt); }
            mfma_tile(cur);
            cur = nxt;
        }
    } else
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        Tile cur;
        cur.img = (int)(tile / tpi); cur.p0 = (int)(tile % tpi) * TPX;
        const int pend = min(cur.p0 + TPX, HW);
        cur.r0 = cur.p0 / W;
        cur.rows = (pend - 1) / W - cur.r0 + 3;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // the previous tile's fragment reads are done
        {   // x patch: slot = (padded pixel, channel quad); a thread's slots are 32 padded pixels apart
            const int c4 = tid & 7;
            int pp = tid >> 3;
            int pr = pp / PW, px = pp - pr * PW;
            long g = ((long)cur.img * HW + (long)(cur.r0 - 1 + pr) * W + (px - 1)) * CH + 4 * c4;
            while (pr < cur.rows) {
                constexpr int NS = 4;
                float4 v[NS]; uint2 hv[NS]; int off[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    off[i] = -1;
                    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    hv[i] = make_uint2(0u, 0u);
                    if (pr < cur.rows) {
                        const int gy = cur.r0 - 1 + pr;
                        if (gy >= 0 && gy < H && px >= 1 && px <= W) {
                            if constexpr (F32) v[i] = *reinterpret_cast<const float4*>(x + g);
                            else hv[i] = *reinterpret_cast<const uint2*>(x + g);
                        }
                        off[i] = pp * CH + 4 * c4;
                    }
                    pp += 32; px += 32; g += 32L * CH;
                    while (px >= PW) { px -= PW; ++pr; g -= 2 * CH; }
                }
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    if (off[i] >= 0) {
                        if constexpr (F32) {
                            unsigned e0[NP], e1[NP];
                            splitn_bf16<NP>(v[i].x, v[i].y, e0); splitn_bf16<NP>(v[i].z, v[i].w, e1);
#pragma unroll
                            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(xh + q * plane_elems + off[i]) = make_uint2(e0[q], e1[q]);
                        } else {
                            *reinterpret_cast<uint2*>(xh + off[i]) = hv[i];
                        }
                    }
            }
            // dy strip: 128 pixels x 8 quads = 4 slots per thread; rows past the image end are zeros (their products vanish)
            float4 d[4]; uint2 dv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pix = (tid >> 3) + 32 * i, p = cur.p0 + pix;
                d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                dv[i] = make_uint2(0u, 0u);
                if (p < HW) {
                    if constexpr (F32) d[i] = *reinterpret_cast<const float4*>(dy + ((long)cur.img * HW + p) * CH + 4 * c4);
                    else dv[i] = *reinterpret_cast<const uint2*>(dy + ((long)cur.img * HW + p) * CH + 4 * c4);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = ((tid >> 3) + 32 * i) * CH + 4 * c4;
                if constexpr (F32) {
                    unsigned e0[NP], e1[NP];
                    splitn_bf16<NP>(d[i].x, d[i].y, e0); splitn_bf16<NP>(d[i].z, d[i].w, e1);
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dh + q * TPX * CH + o) = make_uint2(e0[q], e1[q]);
                } else {
                    *reinterpret_cast<uint2*>(dh + o) = dv[i];
                }
            }
        }
        __syncthreads();
        mfma_tile(cur);
    }
    // ---- the four waves' accumulators, added in wave order through LDS; one partial [co][tap][ci] per workgroup ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(wpl);                                  // 9216 floats
    for (int w = 0; w < WG_NT / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float* d = red + (co * 9 + t) * CH + l31;
                    *d = (w == 0 ? 0.f : *d) + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (long)blockIdx.x * 9 * CH * CH;
    for (int e = tid; e < 9 * CH * CH / 4; e += WG_NT) reinterpret_cast<float4*>(out)[e] = reinterpret_cast<const float4*>(red)[e];
}

}  // namespace

// x [N][H][W][32], w [32][3][3][32] (n, tap, k), y [N][H][W][32] = act(beta*y + conv3x3(x, w)); flip = 1 visits the taps in
// reverse (data gradient).  Returns -100 if the shape does not fit the LDS budget (caller falls back to the implicit GEMM).
int conv3x3_c32_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta,
                              hipStream_t st) {
    const int rows_max = (TP + W - 2) / W + 1 + 2;
    const int patch_floats = rows_max * (W + 2) * CS;
    const size_t lds = (size_t)2 * patch_floats * sizeof(float);
    if (lds > 150 * 1024 || (long)H * W < TP || rows_max * (W + 2) * 8 > 8 * NT || W + 2 <= 64) return -100;   // one workgroup per CU, 8 slots per thread
    static bool attr_set[64] = {false};                  // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return ha2g_set_error(-2, "conv3x3_c32: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    const long tiles = (long)N * (((long)H * W + TP - 1) / TP);
    const int grid = (int)(tiles < 256 ? tiles : 256);
    const unsigned row_slots = (unsigned)(W + 2) * 8;
    const unsigned row_magic = (unsigned)((0x100000000ULL + row_slots - 1) / row_slots);    // s / row_slots = umulhi(s, magic), s < 2^16
    hipLaunchKernelGGL(conv3x3_c32_kernel, dim3(grid), dim3(NT), lds, st, x, w, y, N, H, W, flip, act, beta, patch_floats, row_magic);
    HA2G_CHECK_LAUNCH("conv3x3_c32");
    return 0;
}

// split-bf16 variants (used for the data gradient); C = 32 or 64 channels in and out; same contract
template <int C, int NP = 2>
static int conv3x3_x3_launch_t(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta, hipStream_t st) {
    using G = X3Geo<C, NP>;
    const int rows_max = (TP + W - 2) / W + 1 + 2;
    const int plane_elems = rows_max * (W + 2) * G::CSX;
    const size_t lds = (size_t)NP * plane_elems * sizeof(unsigned short);
    if (lds > 150 * 1024 || (long)H * W < TP || W + 2 <= G::PSTEP) return -100;
    static bool attr_set[64] = {false};                  // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<C, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return ha2g_set_error(-2, "conv3x3_x3: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    const long items = (long)N * (((long)H * W + TP - 1) / TP) * G::NH;
    int grid = (int)(items < 256 ? items : 256);
    grid = grid / G::NH * G::NH;                         // workgroup parity = output-channel half
    if (grid < G::NH) return -100;
    hipLaunchKernelGGL((conv3x3_x3_kernel<C, NP>), dim3(grid), dim3(G::THREADS), lds, st, x, w, y, N, H, W, flip, act, beta, plane_elems);
    HA2G_CHECK_LAUNCH("conv3x3_x3");
    return 0;
}
static int g_x3_prefetch = 1;       // conv3x3_x3p_kernel where its 16 slots per thread hold the patch (ha2g_conv_c32_prefetch(0): the first form, A/B)
static int g_x3p_dbg = 0;
static int g_c32pp = 1;             // the anti-phase kernel conv3x3_c32pp_kernel (ha2g_conv_c32_prefetch bit 5 clears it: A/B)
extern "C" void ha2g_conv_c32_prefetch(int on) { g_x3_prefetch = on & 1; g_x3p_dbg = ((on >> 1) & 15) | (((on >> 6) & 3) << 4) | (((on >> 8) & 1) << 6); g_c32pp = !((on >> 5) & 1); }
static int conv3x3_x3p_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta, hipStream_t st) {
    using G = X3Geo<32, 3>;
    const int rows_max = (TP + W - 2) / W + 1 + 2;
    const int plane_elems = rows_max * (W + 2) * G::CSX;
    const size_t lds = (size_t)3 * plane_elems * sizeof(unsigned short);
    if (lds > 150 * 1024 || (long)H * W < TP || W + 2 <= G::PSTEP) return -100;
    if ((long)rows_max * (W + 2) * G::QP > 16L * G::THREADS) return -100;          // the patch must fit the 16 prefetch slots per thread
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return ha2g_set_error(-2, "conv3x3_x3p: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    const long items = (long)N * (((long)H * W + TP - 1) / TP);
    const int grid = (int)(items < 256 ? items : 256);
    hipLaunchKernelGGL(conv3x3_x3p_kernel, dim3(grid), dim3(G::THREADS), lds, st, x, w, y, N, H, W, flip, (act & 255) | (g_x3p_dbg << 8), beta, plane_elems);
    HA2G_CHECK_LAUNCH("conv3x3_x3p");
    return 0;
}
// does the anti-phase kernel serve an H x W image (in the current mode)?  (gemm.hip: ha2g_conv2d_dgrad_resid_supported)
int conv3x3_c32pp_serves(int H, int W) {
    if (W < 2) return 0;
    const int rows_max = (TPH + W - 2) / W + 1 + 2;
    const int plane_elems = rows_max * (W + 2) * 32;
    const size_t lds = (size_t)2 * 3 * plane_elems * sizeof(unsigned short) + 2 * 9 * 64 * 16;
    if (lds > 158 * 1024 || (long)H * W < TPH || W + 2 <= 32) return 0;
    if ((long)rows_max * (W + 2) * 8 > 12L * 256) return 0;
    return gemm_bwd_pieces() == 3 && g_c32pp;
}
static int conv3x3_c32pp_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta, hipStream_t st,
                                const float* rsd = nullptr, const unsigned* rbits = nullptr) {
    const int rows_max = (TPH + W - 2) / W + 1 + 2;
    const int plane_elems = rows_max * (W + 2) * 32;
    const size_t lds = (size_t)2 * 3 * plane_elems * sizeof(unsigned short) + 2 * 9 * 64 * 16;      // two groups' patches + the filter's third piece
    if (lds > 158 * 1024 || (long)H * W < TPH || W + 2 <= 32 || W < 2) return -100;
    if ((long)rows_max * (W + 2) * 8 > 12L * 256) return -100;                       // the patch must fit the 12 prefetch slots per thread
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
            return ha2g_set_error(-2, "conv3x3_c32pp: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    const long tiles = (long)N * (((long)H * W + TPH - 1) / TPH);
    const long pairs = (tiles + 1) / 2;
    const int grid = (int)(pairs < 256 ? pairs : 256);
    const unsigned mgW = (unsigned)(((1ULL << 32) + (unsigned)W - 1) / (unsigned)W);
    hipLaunchKernelGGL(conv3x3_c32pp_kernel, dim3(grid), dim3(512), lds, st, x, w, y, N, H, W, flip, (act & 255) | (g_x3p_dbg << 8), beta, plane_elems, mgW,
                       rsd, rbits);
    HA2G_CHECK_LAUNCH("conv3x3_c32pp");
    return 0;
}
int conv3x3_c32_x3_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta, hipStream_t st, const float* rsd,
                          const unsigned* rbits) {
    if (gemm_bwd_pieces() == 3 && g_c32pp) {
        const int rc = conv3x3_c32pp_launch(x, w, y, N, H, W, flip, act, beta, st, rsd, rbits);
        if (rc != -100) return rc;
    }
    if (rsd != nullptr) return -100;                         // the masked-residual epilogue exists in the anti-phase kernel only
    if (gemm_bwd_pieces() == 3 && g_x3_prefetch) {
        const int rc = conv3x3_x3p_launch(x, w, y, N, H, W, flip, act, beta, st);
        if (rc != -100) return rc;
    }
    if (gemm_bwd_pieces() == 3) return conv3x3_x3_launch_t<32, 3>(x, w, y, N, H, W, flip, act, beta, st);      // fp32-class: three pieces
    return conv3x3_x3_launch_t<32, 2>(x, w, y, N, H, W, flip, act, beta, st);
}

// dW partials of the direct 32-channel weight gradient: returns the number of partial [32][9][32] blocks written to `part` (the caller
// reduces them), -100 when the shape does not fit (caller falls back to the implicit GEMM), < 0 on error.
int conv3x3_c32_wgrad_blocks(int N, int H, int W) {
    const long tiles = (long)N * (((long)H * W + WG_TP - 1) / WG_TP);
    int cus = hw_cu_count();
    if (cus > g_side_cus) cus = g_side_cus;
    return (int)(tiles < 2L * cus ? tiles : 2L * cus);
}
static int g_c32_wgrad_pf = 1;      // ha2g_conv_c32_wgrad_prefetch: 0 = first form, 1 (default) = register-prefetching form, 2 = + 256-pixel tiles where they fit (no gain in the step: 36.16 vs 36.20 ms)
extern "C" void ha2g_conv_c32_wgrad_prefetch(int on) { g_c32_wgrad_pf = on; }
template <typename T, int NP = 2>
static int c32_wgrad_launch_t(const T* x, const T* dy, float* part, int N, int H, int W, hipStream_t st) {
    const int rows_max = (WG_TP + W - 2) / W + 1 + 2;
    const int plane_elems = rows_max * (W + 2) * CH;
    size_t lds = ((size_t)NP * plane_elems + NP * WG_TP * CH) * sizeof(unsigned short);
    if (lds < 9 * CH * CH * sizeof(float)) lds = 9 * CH * CH * sizeof(float);
    constexpr int LDS_MAX = NP == 3 ? 150 * 1024 : 78 * 1024;                         // NP <= 2: two workgroups per CU; NP = 3: one
    if (lds > LDS_MAX || (long)H * W < WG_TP || W + 2 <= 32) return -100;             // 32 padded pixels between a thread's slots
    static bool attr_set[64] = {false};                                              // per instantiation and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    constexpr bool CAN_PF = sizeof(T) == 4 && NP == 3;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_wgrad_kernel<T, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess)
            return ha2g_set_error(-2, "conv3x3_c32_wgrad: cannot raise the dynamic LDS limit");
        if constexpr (CAN_PF)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_wgrad_kernel<T, NP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess)
                return ha2g_set_error(-2, "conv3x3_c32_wgrad: cannot raise the dynamic LDS limit");
        attr_set[dev] = true;
    }
    int grid = conv3x3_c32_wgrad_blocks(N, H, W);
    if (NP == 3 && grid > 256) grid = (grid + 1) / 2;                                 // one workgroup per CU
    if constexpr (CAN_PF) {
        // the prefetching form: the patch in 12 (16) register slots per thread, tensors inside one 2 GB buffer resource; 256-pixel tiles where they fit the LDS
        const int rows2 = (256 + W - 2) / W + 1 + 2, pe2 = rows2 * (W + 2) * CH;
        const size_t lds2 = ((size_t)NP * pe2 + NP * 256 * CH) * sizeof(unsigned short);
        if (g_c32_wgrad_pf >= 2 && rows2 * (W + 2) <= 16 * 32 && lds2 <= (size_t)LDS_MAX && (long)H * W >= 256 && (long)N * H * W * CH * 4 < (1L << 31)) {
            static bool attr2[64] = {false};
            if (!attr2[dev]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_wgrad_kernel<T, NP, true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess)
                    return ha2g_set_error(-2, "conv3x3_c32_wgrad: cannot raise the dynamic LDS limit");
                attr2[dev] = true;
            }
            const long tiles2 = (long)N * (((long)H * W + 255) / 256);
            int grid2 = grid;
            if (grid2 > tiles2) grid2 = (int)tiles2;
            hipLaunchKernelGGL((conv3x3_c32_wgrad_kernel<T, NP, true, 256>), dim3(grid2), dim3(WG_NT), lds2, st, x, dy, part, N, H, W, pe2);
            HA2G_CHECK_LAUNCH("conv3x3_c32_wgrad (prefetching, 256-pixel tiles)");
            return grid2;
        }
        if (g_c32_wgrad_pf && rows_max * (W + 2) <= 12 * 32 && (long)N * H * W * CH * 4 < (1L << 31)) {
            hipLaunchKernelGGL((conv3x3_c32_wgrad_kernel<T, NP, true>), dim3(grid), dim3(WG_NT), lds, st, x, dy, part, N, H, W, plane_elems);
            HA2G_CHECK_LAUNCH("conv3x3_c32_wgrad (prefetching)");
            return grid;
        }
    }
    hipLaunchKernelGGL((conv3x3_c32_wgrad_kernel<T, NP>), dim3(grid), dim3(WG_NT), lds, st, x, dy, part, N, H, W, plane_elems);
    HA2G_CHECK_LAUNCH("conv3x3_c32_wgrad");
    return grid;
}
int conv3x3_c32_wgrad_launch(const float* x, const float* dy, float* part, int N, int H, int W, hipStream_t st) {
    if (gemm_bwd_pieces() == 3) return c32_wgrad_launch_t<float, 3>(x, dy, part, N, H, W, st);                  // fp32-class: three pieces
    return c32_wgrad_launch_t<float, 2>(x, dy, part, N, H, W, st);
}
// bf16 tensors (bf16-storage mode): single planes, one MFMA per product
int conv3x3_c32_wgrad_b16_launch(const void* x, const void* dy, float* part, int N, int H, int W, hipStream_t st) {
    return c32_wgrad_launch_t<b16>((const b16*)x, (const b16*)dy, part, N, H, W, st);
}
