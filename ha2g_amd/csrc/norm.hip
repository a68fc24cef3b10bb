// Train-mode BatchNorm (2-D and 1-D share one [rows][C] channels-last form) and the squeeze-excite
// pointwise pieces of the SE-ResNet audio encoder.  HBM-bound kernels: float4 lanes over channels,
// per-block partial column sums + a tiny fixed-order final reduction in double (deterministic).
//
// BatchNorm semantics = torch (reference scripts/model/ResNetBlocks.py:24-30, ResNetSE34V2.py:127-129):
// normalise with the biased batch variance, update running_var with the unbiased one, momentum 0.1.
#include "common.h"

namespace {

constexpr int NB_MAX = 1024;   // max row-chunk blocks of a partial reduction

// element access of the activation tensors (fp32 | bf16-storage mode): ld4 / st4 / ld1 of common.h

// A thread moves 16 bytes per access: V = 4 fp32 or 8 bf16 consecutive channels (VW<T>::V); ldv / stv / ldp below.
template <typename T> struct VW { static constexpr int V = 16 / (int)sizeof(T); };
template <int V> struct fvec { float v[V]; };
template <int V> struct dvec { double v[V]; };
template <int V> __device__ __forceinline__ dvec<V> dzero() { dvec<V> r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = 0.0;
    return r; }
__device__ __forceinline__ fvec<4> ldv(const float* p, long i) {
    const float4 w = reinterpret_cast<const float4*>(p)[i];
    fvec<4> r; r.v[0] = w.x; r.v[1] = w.y; r.v[2] = w.z; r.v[3] = w.w; return r;
}
__device__ __forceinline__ fvec<8> ldv(const b16* p, long i) {
    const uint4 w = reinterpret_cast<const uint4*>(p)[i];
    fvec<8> r;
    r.v[0] = __uint_as_float(w.x << 16); r.v[1] = __uint_as_float(w.x & 0xffff0000u); r.v[2] = __uint_as_float(w.y << 16); r.v[3] = __uint_as_float(w.y & 0xffff0000u);
    r.v[4] = __uint_as_float(w.z << 16); r.v[5] = __uint_as_float(w.z & 0xffff0000u); r.v[6] = __uint_as_float(w.w << 16); r.v[7] = __uint_as_float(w.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void stv(float* p, long i, const fvec<4>& r) { reinterpret_cast<float4*>(p)[i] = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]); }
__device__ __forceinline__ void stv(b16* p, long i, const fvec<8>& r) {        // round to nearest even
    uint4 h; unsigned l;
    split2_bf16(r.v[0], r.v[1], h.x, l); split2_bf16(r.v[2], r.v[3], h.y, l); split2_bf16(r.v[4], r.v[5], h.z, l); split2_bf16(r.v[6], r.v[7], h.w, l);
    reinterpret_cast<uint4*>(p)[i] = h;
}
// per-channel fp32 parameters of the V channels of vector lane cv
template <int V> __device__ __forceinline__ fvec<V> ldp(const float* p, int cv) {
    fvec<V> r;
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
        const float4 w = reinterpret_cast<const float4*>(p)[cv * (V / 4) + q];
        r.v[4 * q] = w.x; r.v[4 * q + 1] = w.y; r.v[4 * q + 2] = w.z; r.v[4 * q + 3] = w.w;
    }
    return r;
}
// the two bf16 planes of the split-bf16 product (fp32 tensors only)
// pnp = 3: three pieces (split3_bf16, all 24 mantissa bits), the third plane one plane stride (lo - hi) behind lo: equally spaced planes
__device__ __forceinline__ void st_planes(unsigned short* hi, unsigned short* lo, int pnp, long i, const fvec<4>& r) {
    if (pnp == 3) {
        unsigned a[3], b[3];
        split3_bf16(r.v[0], r.v[1], a[0], a[1], a[2]); split3_bf16(r.v[2], r.v[3], b[0], b[1], b[2]);
        reinterpret_cast<uint2*>(hi)[i] = make_uint2(a[0], b[0]);
        reinterpret_cast<uint2*>(lo)[i] = make_uint2(a[1], b[1]);
        reinterpret_cast<uint2*>(lo + (lo - hi))[i] = make_uint2(a[2], b[2]);
        return;
    }
    uint2 h, l;
    split2_bf16(r.v[0], r.v[1], h.x, l.x); split2_bf16(r.v[2], r.v[3], h.y, l.y);
    reinterpret_cast<uint2*>(hi)[i] = h;
    reinterpret_cast<uint2*>(lo)[i] = l;
}
__device__ __forceinline__ void st_planes(unsigned short*, unsigned short*, int, long, const fvec<8>&) {}

// thread layout shared by the column reductions: CV = C / V vector lanes per row, 256 % CV == 0
struct ColMap {
    int cv, r0, rstep;
    __device__ ColMap(int CV) { cv = threadIdx.x % CV; r0 = threadIdx.x / CV; rstep = 256 / CV; }
};

// combine the per-thread double partials of threads sharing cv, write [which][C][blk] (double); lds: [2][256] dvec<V>
template <int V>
__device__ __forceinline__ void block_col_reduce(const dvec<V>& a, const dvec<V>& b, int CV, double* __restrict__ part, int C, dvec<V>* lds) {
    lds[threadIdx.x] = a;
    lds[256 + threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < CV) {
        dvec<V> sa = dzero<V>(), sb = dzero<V>();
        for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
            for (int k = 0; k < V; ++k) { sa.v[k] += lds[t].v[k]; sb.v[k] += lds[256 + t].v[k]; }
        }
        // layout [which][C][blocks]: the final kernels' waves then read a column's partials CONTIGUOUSLY (the first layout, [block][which][C], made every
        // lane of a final wave touch its own cache line: 7 us per final launch, 122 of them per step on the main queue)
        const long nb = gridDim.x;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            part[((long)(threadIdx.x * V + k)) * nb + blockIdx.x] = sa.v[k];
            part[((long)C + threadIdx.x * V + k) * nb + blockIdx.x] = sb.v[k];
        }
    }
}

// The same for narrow tensors (CV = C / V <= 16 lanes per row, i.e. C <= 64 fp32 channels) with 1 / 8 of the LDS (round 6): the partials of the lanes of a
// wave that share cv (lane % CV) are first added by xor-shuffles over the lane bits above CV (a fixed tree), then one dvec per (wave, cv) goes through LDS.
// Why: beside a persistent weight-gradient workgroup (120-156 KB of the CU's 160 KB of LDS) a 16-24 KB reduction buffer allows ONE workgroup of a
// statistics pass per CU (or none); 2-3 KB allows as many as the registers do.  Deterministic; another summation order than block_col_reduce.
__device__ __forceinline__ double shfl_xor_d(double v, int o) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(b & 0xffffffffLL), o, 64), hi = __shfl_xor((int)(b >> 32), o, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int V>
__device__ __forceinline__ dvec<V> wave_cv_sum(dvec<V> a, int CV) {
    for (int o = CV; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] += shfl_xor_d(a.v[k], o);
    }
    return a;
}
// lds: [2][4][16] dvec<V>
template <int V>
__device__ __forceinline__ void block_col_reduce_small(dvec<V> a, dvec<V> b, int CV, double* __restrict__ part, int C, dvec<V>* lds) {
    a = wave_cv_sum<V>(a, CV); b = wave_cv_sum<V>(b, CV);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < CV) { lds[w * 16 + lane] = a; lds[64 + w * 16 + lane] = b; }
    __syncthreads();
    if (threadIdx.x < CV) {
        dvec<V> sa = dzero<V>(), sb = dzero<V>();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < V; ++k) { sa.v[k] += lds[q * 16 + threadIdx.x].v[k]; sb.v[k] += lds[64 + q * 16 + threadIdx.x].v[k]; }
        }
        const long nb = gridDim.x;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            part[((long)(threadIdx.x * V + k)) * nb + blockIdx.x] = sa.v[k];
            part[((long)C + threadIdx.x * V + k) * nb + blockIdx.x] = sb.v[k];
        }
    }
}

// mode 0: (x - K, (x-K)^2) with K = row 0 (shift against cancellation); mode 1: (dy, dy * xhat).
// Sums are carried in double (torch's CPU batch-norm accumulates in double too; the kernel is HBM-bound, the
// fp64 adds are free) -- nearly-dead post-ReLU channels make sum(dy*xhat) cancel by 1e3..1e4.
template <int MODE, typename T, int RPT = 4, int SMALL = 0>
__global__ __launch_bounds__(256) void col_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          long rows, int C, double* __restrict__ part) {
    constexpr int V = VW<T>::V;
    __shared__ dvec<V> lds[SMALL == 2 ? 1 : (SMALL == 1 ? 128 : 512)];
    const int CV = C / V;
    ColMap m(CV);
    const long rows_per = (rows + gridDim.x - 1) / gridDim.x;
    const long rbeg = (long)blockIdx.x * rows_per, rend = min(rows, rbeg + rows_per);
    dvec<V> a = dzero<V>(), b = dzero<V>();
    fvec<V> k4, is4;
    if (MODE == 0) { k4 = ldv(x, m.cv); is4 = k4; }
    else { k4 = ldp<V>(mean, m.cv); is4 = ldp<V>(invstd, m.cv); }
    // four rows per trip: the loads are issued together (one row per trip is load -> wait -> add, a single 16-byte request in flight per
    // wave: ~2.7 TB/s measured), the adds keep the row order, so the sums are bit-identical to the one-row loop's
    auto accum = [&](const fvec<V>& v, const fvec<V>& d) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (MODE == 0) {
                const double vx = (double)v.v[k] - k4.v[k];
                a.v[k] += vx;
                b.v[k] += vx * vx;
            } else {
                a.v[k] += d.v[k];
                b.v[k] += (double)d.v[k] * (((double)v.v[k] - k4.v[k]) * is4.v[k]);
            }
        }
    };
    long r = rbeg + m.r0;
    for (; r + (RPT - 1) * m.rstep < rend; r += RPT * m.rstep) {
        fvec<V> v[RPT], d[RPT];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            v[j] = ldv(x + (r + (long)j * m.rstep) * C, m.cv);
            d[j] = v[j];
            if (MODE == 1) d[j] = ldv(dy + (r + (long)j * m.rstep) * C, m.cv);
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) accum(v[j], d[j]);
    }
    for (; r < rend; r += m.rstep) {
        fvec<V> v = ldv(x + r * C, m.cv), d = v;
        if (MODE == 1) d = ldv(dy + r * C, m.cv);
        accum(v, d);
    }
    if constexpr (SMALL == 2) {
        // no LDS at all: every WAVE leaves its own partial (block index 4 blockIdx + wave of 4 gridDim blocks) -- a statistics pass then fits beside a
        // persistent plane weight-gradient workgroup (156 of the CU's 160 KB of LDS), where no LDS-using workgroup can be resident
        a = wave_cv_sum<V>(a, CV); b = wave_cv_sum<V>(b, CV);
        const int lane = threadIdx.x & 63;
        if (lane < CV) {
            const long nb = 4L * gridDim.x, blk = 4L * blockIdx.x + (threadIdx.x >> 6);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                part[((long)(lane * V + k)) * nb + blk] = a.v[k];
                part[((long)C + lane * V + k) * nb + blk] = b.v[k];
            }
        }
    } else if constexpr (SMALL == 1) block_col_reduce_small<V>(a, b, CV, part, C, lds);
    else block_col_reduce<V>(a, b, CV, part, C, lds);
}

// Per-IMAGE column sums (x, x^2) of x [N][HW][C] (round 6): grid (nchunk, N), block (k, n) = rows [k per, (k + 1) per) of image n -> part [2][C][N nchunk]
// doubles, block index n nchunk + k: the layout (and the "tiles inside one image" property) of a convolution epilogue's statistics, so that
// bn_stats_final_kernel (x == null: unshifted sums) AND the SE squeeze (se_mlp_fwd_kernel) read them -- layer 1's 32-channel convolutions have no
// statistics epilogue, and its bn2 output was written by one pass and re-read by the next only to be pooled and scaled.  Unshifted squares in double: the
// variance loses log2(mean^2 / var) of 53 bits.
__global__ __launch_bounds__(256) void bn_image_partial_kernel(const float* __restrict__ x, int HW, int C, double* __restrict__ part) {
    __shared__ dvec<4> lds[512];
    const int CV = C / 4;
    ColMap m(CV);
    const int nchunk = gridDim.x, per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const float* xb = x + (long)blockIdx.y * HW * C;
    dvec<4> a = dzero<4>(), b = dzero<4>();
    auto accum = [&](const fvec<4>& v) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double vx = (double)v.v[k]; a.v[k] += vx; b.v[k] += vx * vx; }
    };
    int r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {
        fvec<4> v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ldv(xb + (long)(r + j * m.rstep) * C, m.cv);
#pragma unroll
        for (int j = 0; j < 4; ++j) accum(v[j]);
    }
    for (; r < rend; r += m.rstep) accum(ldv(xb + (long)r * C, m.cv));
    lds[threadIdx.x] = a;
    lds[256 + threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < CV) {
        dvec<4> sa = dzero<4>(), sb = dzero<4>();
        for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { sa.v[k] += lds[t].v[k]; sb.v[k] += lds[256 + t].v[k]; }
        }
        const long nb = (long)gridDim.x * gridDim.y, blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            part[((long)(threadIdx.x * 4 + k)) * nb + blk] = sa.v[k];
            part[((long)C + threadIdx.x * 4 + k) * nb + blk] = sb.v[k];
        }
    }
}

// final reductions: one wave per column; lanes stride the block partials, fixed-order shuffle tree (deterministic)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// one BLOCK per column: wave w adds the partials b = 64 j + lane of its quarter [w nq, (w + 1) nq) of the blocks (four loads per trip in flight), a
// fixed shuffle tree per wave, the four wave sums added in wave order by thread 0: deterministic.  (One wave per column read a convolution
// epilogue's 2 560 tile sums in 8.8 us.)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int nblk, long rows, int C,
                                                             const T* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ invstd, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float momentum, float eps) {
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x;
    const int nq = (nblk + 3) >> 2, b0 = w * nq, b1 = min(nblk, b0 + nq);
    const double* p1 = part + (long)c * nblk;
    const double* p2 = part + ((long)C + c) * nblk;
    double s1 = 0.0, s2 = 0.0;
    {
        int b = b0 + lane;
        for (; b + 192 < b1; b += 256) {
            double q1[4], q2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { q1[j] = p1[b + 64 * j]; q2[j] = p2[b + 64 * j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += q1[j]; s2 += q2[j]; }
        }
        for (; b < b1; b += 64) { s1 += p1[b]; s2 += p2[b]; }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) { red[w] = s1; red[4 + w] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = ((red[0] + red[1]) + red[2]) + red[3];
    s2 = ((red[4] + red[5]) + red[6]) + red[7];
    double n = (double)rows, m1 = s1 / n;
    double var = s2 / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    double mu = (x != nullptr ? (double)ld1(x, c) : 0.0) + m1;      // x == null: unshifted partial sums (a convolution epilogue's)
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
    }
}

__global__ __launch_bounds__(256) void pair_final_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ out0,
                                                         float* __restrict__ out1, float* __restrict__ acc0, float* __restrict__ acc1) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    {   // partials in fixed order, four pairs of loads per trip in flight
        int b = lane;
        for (; b + 192 < nblk; b += 256) {
            double p1[4], p2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { p1[j] = part[(long)c * nblk + b + 64 * j]; p2[j] = part[((long)C + c) * nblk + b + 64 * j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += p1[j]; s2 += p2[j]; }
        }
        for (; b < nblk; b += 64) { s1 += part[(long)c * nblk + b]; s2 += part[((long)C + c) * nblk + b]; }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) {
        out0[c] = (float)s1; out1[c] = (float)s2;
        if (acc0) acc0[c] += (float)s1;                  // optional: accumulate straight into the parameters' .grad buffers
        if (acc1) acc1[c] += (float)s2;
    }
}

// y = (x - mean) * invstd * gamma + beta ; act: 0 none, 2 leaky-relu(0.01)
// PL = 1: y is ALSO written as the two bf16 planes of the split-bf16 product (hi = bf16(y), lo = bf16(y - hi)): it is the x operand of the
// next convolution's plane-based weight gradient (conv_planes.hip), split once here instead of once per consumer tile and tap in the backward
template <int PL, typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, long rows,
                                int C, int act, unsigned short* __restrict__ y_hi, unsigned short* __restrict__ y_lo, int pnp) {
    constexpr int V = VW<T>::V;
    static_assert(!PL || V == 4, "planes are written from fp32 tensors");
    const int CV = C / V;
    const long total = rows * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const fvec<V> v = ldv(x, i);
        const fvec<V> mu = ldp<V>(mean, cv), is = ldp<V>(invstd, cv), g = ldp<V>(gamma, cv), b = ldp<V>(beta, cv);
        fvec<V> r;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            r.v[k] = (v.v[k] - mu.v[k]) * is.v[k] * g.v[k] + b.v[k];
            if (act == 2) r.v[k] = r.v[k] > 0.f ? r.v[k] : 0.01f * r.v[k];
        }
        if (!PL || y != nullptr) stv(y, i, r);             // planes only (y == NULL): every consumer of y reads the piece planes
        if (PL) st_planes(y_hi, y_lo, pnp, i, r);
    }
}


// BatchNorm apply fused with the squeeze of the SE layer that follows it (ResNetBlocks.py:24-36,81-83): writes y = bn(x) and
// per-(image, channel) sums of y in the same pass (the separate hw-mean kernel re-read the whole tensor).  grid (chunks, N):
// a block owns a row chunk of ONE image; per-thread double partials, block combine through LDS, one partial per
// (image, chunk, channel); pool_final adds the chunks in ascending order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, int HW, int C,
                                                            double* __restrict__ part) {
    constexpr int V = VW<T>::V;
    __shared__ dvec<V> lds[256];
    const int CV = C / V;
    ColMap m(CV);
    const int nchunk = gridDim.x;
    const int per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const long base = (long)blockIdx.y * HW * C;
    const fvec<V> mu = ldp<V>(mean, m.cv), is = ldp<V>(invstd, m.cv), g = ldp<V>(gamma, m.cv), b = ldp<V>(beta, m.cv);
    dvec<V> a = dzero<V>();
    auto row = [&](const fvec<V>& v, int r) {
        fvec<V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = (v.v[k] - mu.v[k]) * is.v[k] * g.v[k] + b.v[k];
        stv(y + base + (long)r * C, m.cv, o);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] += o.v[k];          // the squeeze sums the unrounded fp32 values (bf16 storage rounds only the store)
    };
    int r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {           // four loads in flight per wave, rows consumed in order
        fvec<V> v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ldv(x + base + (long)(r + j * m.rstep) * C, m.cv);
#pragma unroll
        for (int j = 0; j < 4; ++j) row(v[j], r + j * m.rstep);
    }
    for (; r < rend; r += m.rstep) row(ldv(x + base + (long)r * C, m.cv), r);
    lds[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < CV) {
        dvec<V> s = dzero<V>();
        for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
            for (int k = 0; k < V; ++k) s.v[k] += lds[t].v[k];
        }
        double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * C + threadIdx.x * V;
#pragma unroll
        for (int k = 0; k < V; ++k) p[k] = s.v[k];
    }
}
// gate (nullable): out = sum * scale * g (1 - g) -- the SE backward's sigmoid' factor, so that no separate pointwise launch follows
__global__ void pool_final_kernel(const double* __restrict__ part, int nchunk, int C, long NC, float scale, float* __restrict__ out,
                                  const float* __restrict__ gate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    const long n = i / C; const int c = (int)(i % C);
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(n * nchunk + k) * C + c];
    float v = (float)(s * scale);
    if (gate) { const float g = gate[i]; v = v * g * (1.f - g); }
    out[i] = v;
}

// SE squeeze of a BatchNorm's output from the per-tile column sums of its INPUT (a convolution epilogue's, tiles inside one image):
// pooled[n][c] = (sum of image n's tiles / HW - mean) invstd gamma + beta -- the mean of an affine map is the affine map of the mean
__global__ void pool_from_partials_kernel(const double* __restrict__ part, int nblk, int gpi, int C, long NC, double inv_hw,
                                          const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ pooled) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    const long n = i / C; const int c = (int)(i - n * C);
    const double* p = part + (long)c * nblk + n * gpi;
    double s = 0.0;
    for (int k = 0; k < gpi; ++k) s += p[k];
    pooled[i] = (float)((s * inv_hw - (double)mean[c]) * (double)invstd[c] * (double)gamma[c] + (double)beta[c]);
}

// dx = gamma * invstd * (dy - sum_dy/N - xhat * sum_dy_xhat/N)
// PL = 1: dx is ALSO (or, with dx == nullptr, ONLY) written as two bf16 planes  hi = bf16(dx), lo = bf16(dx - hi)  -- the operand format of the
// plane-based split-bf16 consumers (conv_planes.hip): this pass is HBM-bound, the split rides in its shadow, and the convolution data /
// weight gradients that read the tensor no longer split it once per consumer tile.
template <int PL, typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat, T* __restrict__ dx,
                                    long rows, int C, int relu_mask, unsigned short* __restrict__ dx_hi, unsigned short* __restrict__ dx_lo, int pnp) {
    constexpr int V = VW<T>::V;
    static_assert(!PL || V == 4, "planes are written from fp32 tensors");
    const int CV = C / V;
    const long total = rows * CV;
    const float invn = 1.f / (float)rows;
    const double dn = (double)invn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const fvec<V> d = ldv(dy, i), v = ldv(x, i);
        const fvec<V> mu = ldp<V>(mean, cv), is = ldp<V>(invstd, cv), g = ldp<V>(gamma, cv), s1 = ldp<V>(sum_dy, cv), s2 = ldp<V>(sum_dy_xhat, cv);
        fvec<V> r;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            r.v[k] = (float)((double)g.v[k] * is.v[k] * ((double)d.v[k] - s1.v[k] * dn - ((double)v.v[k] - mu.v[k]) * is.v[k] * (s2.v[k] * dn)));
            if (relu_mask) r.v[k] = v.v[k] > 0.f ? r.v[k] : 0.f;  // x is a ReLU output (conv -> ReLU -> BN): chain the ReLU derivative, mask = (x > 0)
        }
        if (!PL || dx != nullptr) stv(dx, i, r);
        if (PL) st_planes(dx_hi, dx_lo, pnp, i, r);
    }
}

// ---- squeeze-excite pieces; x is [N][HW][C] -------------------------------------------------------------
// per-image column mean (MODE 0) or per-image sum of a*b (MODE 1: ds[n,c] = sum_hw dpre*b2 with dpre = dout*(out>0))
// AFF (MODE 1): x is the INPUT of the BatchNorm whose output the sum needs -- b2 = (x - mean) invstd gamma + beta is recomputed per element with
// bn_apply_pool_kernel's own expression (the same bits), so that the BatchNorm's output need not exist in memory (block_fwd's statistics path)
struct BnAff { const float* mean; const float* invstd; const float* gamma; const float* beta; };
template <int MODE, typename T, bool AFF = false>
__global__ __launch_bounds__(256) void image_col_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                        const T* __restrict__ outp, int HW, int C, float* __restrict__ res,
                                                        float scale, double* __restrict__ part, const float* __restrict__ gate, BnAff aff = BnAff{}) {
    // grid (nchunk, N): block (k, n) sums rows [k*per, (k+1)*per) of image n; part == nullptr (nchunk = 1): final floats to res, else
    // double partials [n][k][C] for pool_final_kernel (more blocks than images: a 128-image batch alone fills half the CUs)
    constexpr int V = VW<T>::V;
    __shared__ dvec<V> lds[256];
    const int CV = C / V;
    ColMap m(CV);
    const int nchunk = gridDim.x, per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const long base = (long)blockIdx.y * HW * C;
    dvec<V> a = dzero<V>();
    fvec<V> mu, is, g, b;
    if (AFF) { mu = ldp<V>(aff.mean, m.cv); is = ldp<V>(aff.invstd, m.cv); g = ldp<V>(aff.gamma, m.cv); b = ldp<V>(aff.beta, m.cv); }
    // one block per image: four rows per trip so that 4 (MODE 0) / 12 (MODE 1) 16-byte loads are in flight per wave; the adds keep the row
    // order of the one-row loop (bit-identical sums)
    auto accum = [&](const fvec<V>& v, const fvec<V>& d, const fvec<V>& o) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float vk = v.v[k];
            if (AFF) vk = (vk - mu.v[k]) * is.v[k] * g.v[k] + b.v[k];
            if (MODE == 1) a.v[k] += o.v[k] > 0.f ? (double)vk * d.v[k] : 0.0;
            else a.v[k] += vk;
        }
    };
    int r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {
        fvec<V> v[4], d[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long off = base + (long)(r + j * m.rstep) * C;
            v[j] = ldv(x + off, m.cv);
            d[j] = v[j]; o[j] = v[j];
            if (MODE == 1) { d[j] = ldv(dout + off, m.cv); o[j] = ldv(outp + off, m.cv); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) accum(v[j], d[j], o[j]);
    }
    for (; r < rend; r += m.rstep) {
        const long off = base + (long)r * C;
        fvec<V> v = ldv(x + off, m.cv), d = v, o = v;
        if (MODE == 1) { d = ldv(dout + off, m.cv); o = ldv(outp + off, m.cv); }
        accum(v, d, o);
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < CV) {
        dvec<V> s = dzero<V>();
        for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
            for (int k = 0; k < V; ++k) s.v[k] += lds[t].v[k];
        }
        if (part) {
            double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * C + threadIdx.x * V;
#pragma unroll
            for (int k = 0; k < V; ++k) p[k] = s.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float o = (float)(s.v[k] * scale);
                if (gate) { const float g = gate[(long)blockIdx.y * C + threadIdx.x * V + k]; o = o * g * (1.f - g); }
                res[(long)blockIdx.y * C + threadIdx.x * V + k] = o;
            }
        }
    }
}

// out = relu(x * s[n,c] + res)
template <int PL, typename T, bool AFF = false>
__global__ void se_scale_add_relu_kernel(const T* __restrict__ x, const float* __restrict__ s, const T* __restrict__ res,
                                         T* __restrict__ out, long N, int HW, int C, unsigned short* __restrict__ o_hi,
                                         unsigned short* __restrict__ o_lo, int pnp, BnAff aff = BnAff{}, unsigned* __restrict__ mbits = nullptr) {
    // mbits (fp32, C % 32 == 0; nullable): the ReLU decisions (out > 0) as four bits per 16-byte vector, eight vectors per 32-bit word (word i >> 3, nibble
    // i & 7 of vector index i) -- the block's backward reads 1 / 32 of the bytes of `out` for its mask (round 6)
    constexpr int V = VW<T>::V;
    static_assert(!PL || V == 4, "planes are written from fp32 tensors");
    const int CV = C / V;
    const long per = (long)HW * CV, total = N * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / per; const int cv = (int)(i % CV);
        fvec<V> v = ldv(x, i);
        const fvec<V> r = ldv(res, i);
        const fvec<V> sc = ldp<V>(s + n * C, cv);
        if (AFF) {                                          // x is the BatchNorm's INPUT: its output with bn_apply_pool_kernel's expression (the same bits)
            const fvec<V> mu = ldp<V>(aff.mean, cv), is = ldp<V>(aff.invstd, cv), g = ldp<V>(aff.gamma, cv), b = ldp<V>(aff.beta, cv);
#pragma unroll
            for (int k = 0; k < V; ++k) v.v[k] = (v.v[k] - mu.v[k]) * is.v[k] * g.v[k] + b.v[k];
        }
        fvec<V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = fmaxf(v.v[k] * sc.v[k] + r.v[k], 0.f);
        stv(out, i, o);
        if (PL) st_planes(o_hi, o_lo, pnp, i, o);
        if constexpr (V == 4) {
            if (mbits != nullptr) {                         // total % 8 == 0 and a wave's vectors are consecutive: groups of eight lanes are whole
                unsigned nib = (o.v[0] > 0.f ? 1u : 0u) | (o.v[1] > 0.f ? 2u : 0u) | (o.v[2] > 0.f ? 4u : 0u) | (o.v[3] > 0.f ? 8u : 0u);
                unsigned w = nib << (4 * (threadIdx.x & 7));
                w |= __shfl_xor(w, 1, 64); w |= __shfl_xor(w, 2, 64); w |= __shfl_xor(w, 4, 64);
                if ((threadIdx.x & 7) == 0) mbits[i >> 3] = w;
            }
        }
    }
}
// the four ReLU decisions of vector i from se_scale_add_relu_kernel's bit words
__device__ __forceinline__ unsigned relu_bits(const unsigned* __restrict__ mbits, long i) { return (mbits[i >> 3] >> (4 * (int)(i & 7))) & 15u; }
// dpre = dout * (out > 0); dres = dpre; dx = dpre * s[n,c] + dpool[n,c]   (dpool already divided by HW)
template <typename T>
__global__ void se_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ outp, const float* __restrict__ s,
                                    const float* __restrict__ dpool, T* __restrict__ dres, T* __restrict__ dx, long N,
                                    int HW, int C) {
    constexpr int V = VW<T>::V;
    const int CV = C / V;
    const long per = (long)HW * CV, total = N * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / per; const int cv = (int)(i % CV);
        const fvec<V> d = ldv(dout, i), o = ldv(outp, i);
        const fvec<V> sc = ldp<V>(s + n * C, cv), dp = ldp<V>(dpool + n * C, cv);
        fvec<V> p, q;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            p.v[k] = o.v[k] > 0.f ? d.v[k] : 0.f;
            q.v[k] = p.v[k] * sc.v[k] + dp.v[k];
        }
        stv(dres, i, p);
        stv(dx, i, q);
    }
}

// Data path of the SE excitation MLP's backward in one launch (ResNetBlocks.py:84-89 under autograd): per image n
//   dh1[n][j]   = (h1[n][j] > 0) * sum_c dsc[n][c] * w2[c][j]          (fc.2: Linear(R -> C), weight [C][R]; ReLU')
//   dpool[n][c] = inv_hw * sum_j dh1[n][j] * w0[j][c]                   (fc.0: Linear(C -> R), weight [R][C]; the squeeze's 1 / HW)
// C <= 256, R <= 32 (reduction 8).  Three dependent launches (GEMM, eltwise, GEMM) of a few microseconds each sat on the backward's critical path
// per block; fixed summation order (eight strided partials per j, then ascending), no atomics.
// dpart (round 6, nullable): dsc is not given but still the chunk sums of the SE-backward reduction pass (image_col_kernel<1>: [n][nchunk][C] doubles); this
// kernel finishes them as pool_final_kernel did -- chunks added in order, cast, times the gate's sigmoid' -- and writes dsc out (the SE weight gradient reads
// it): one launch less per block, the same bits.
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ dsc, const float* __restrict__ h1, const float* __restrict__ w2,
                                                         const float* __restrict__ w0, float* __restrict__ dh1, float* __restrict__ dpool, int C,
                                                         int R, float inv_hw, const double* __restrict__ dpart = nullptr, int nchunk = 0,
                                                         const float* __restrict__ gate = nullptr, float* __restrict__ dsc_out = nullptr) {
    __shared__ float sd[256];
    __shared__ float part[8][32];
    __shared__ float sh[32];
    const int n = blockIdx.x, t = threadIdx.x;
    if (t < C) {
        float v;
        if (dpart != nullptr) {
            double sm = 0.0;
            for (int k = 0; k < nchunk; ++k) sm += dpart[((long)n * nchunk + k) * C + t];
            v = (float)(sm * 1.f);
            if (gate) { const float g = gate[(long)n * C + t]; v = v * g * (1.f - g); }
            dsc_out[(long)n * C + t] = v;
        } else {
            v = dsc[(long)n * C + t];
        }
        sd[t] = v;
    }
    __syncthreads();
    const int j = t & 31, g = t >> 5;
    float acc = 0.f;
    if (j < R) {
        // eight weight loads in flight, added in the same ascending order (one load -> wait -> fma per term before: 32 L2 round trips = 16 of the kernel's 21 us)
        int c = g;
        for (; c + 56 < C; c += 64) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w2[(long)(c + 8 * u) * R + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += sd[c + 8 * u] * wv[u];
        }
        for (; c < C; c += 8) acc += sd[c] * w2[(long)c * R + j];
    }
    part[g][j] = acc;
    __syncthreads();
    if (t < R) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += part[q][t];
        v = h1[(long)n * R + t] > 0.f ? v : 0.f;
        dh1[(long)n * R + t] = v;
        sh[t] = v;
    }
    __syncthreads();
    if (t < C) {
        float v = 0.f;
        int q = 0;
        for (; q + 7 < R; q += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w0[(long)(q + u) * C + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += sh[q + u] * wv[u];
        }
        for (; q < R; ++q) v += sh[q] * w0[(long)q * C + t];
        dpool[(long)n * C + t] = v * inv_hw;
    }
}

// ---- SE backward + bn2 backward WITHOUT the gradient tensor between them (round 6, second half) ------------------------------------------------
// Block tail (ResNetBlocks.py:30-37): z = bn2(c2); out = relu(z * s[n,c] + residual), s = SE gate of mean_hw z.  With dpre = dout * (out > 0):
//   dz[n,hw,c] = dpre * s[n,c] + dpool[n,c]                 (dpool = the squeeze's gradient, already / HW)
// and bn2's backward needs sum dz and sum dz * xhat per channel.  Both follow from PER-IMAGE sums over hw that the SE reduction pass can take
// while it reads (dout, out, c2) anyway:  A1 = sum dpre, A2 = sum dpre * xhat, A3 = sum xhat  (xhat = (c2 - mean) invstd):
//   SE:  sum_hw dpre * z  = gamma A2 + beta A1           (z = gamma xhat + beta)
//   bn2: sum_hw dz        = s A1 + HW dpool,   sum_hw dz * xhat = s A2 + dpool A3
// so the pass that WROTE dz (se_bwd_apply) and the column pass that READ it back with c2 (col_partial<1>) disappear: two passes over three tensors
// (reduce, apply) instead of four over nine tensor reads.  All sums in double, fixed order (chunks ascending, images ascending): deterministic.
template <bool BITS, int RPT = 4, int SMALL = 0>
__global__ __launch_bounds__(256) void se_bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ outp,
                                                           int HW, int C, double* __restrict__ part, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const unsigned* __restrict__ mbits) {
    // grid (nchunk, N): block (k, n) sums rows [k * per, (k + 1) * per) of image n -> part [n][k][3][C]
    __shared__ dvec<4> lds[3][SMALL == 2 ? 1 : (SMALL == 1 ? 64 : 256)];
    const int CV = C / 4;
    ColMap m(CV);
    const int nchunk = gridDim.x, per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const long base = (long)blockIdx.y * HW * C;
    dvec<4> a1 = dzero<4>(), a2 = dzero<4>(), a3 = dzero<4>();
    const fvec<4> mu = ldp<4>(mean, m.cv), is = ldp<4>(invstd, m.cv);
    // the ReLU decisions: (out > 0) read from the block's output, or BITS: from se_scale_add_relu_kernel's bit words (a 32nd of the bytes)
    auto accum = [&](const fvec<4>& v, const fvec<4>& d, unsigned bits) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double xh = ((double)v.v[k] - mu.v[k]) * is.v[k];          // col_partial_kernel<1>'s xhat
            const double dp = ((bits >> k) & 1u) ? (double)d.v[k] : 0.0;
            a1.v[k] += dp; a2.v[k] += dp * xh; a3.v[k] += xh;
        }
    };
    auto decisions = [&](long off) -> unsigned {
        if (BITS) return relu_bits(mbits, (off >> 2) + m.cv);
        const fvec<4> o = ldv(outp + off, m.cv);
        return (o.v[0] > 0.f ? 1u : 0u) | (o.v[1] > 0.f ? 2u : 0u) | (o.v[2] > 0.f ? 4u : 0u) | (o.v[3] > 0.f ? 8u : 0u);
    };
    int r = rbeg + m.r0;
    for (; r + (RPT - 1) * m.rstep < rend; r += RPT * m.rstep) {              // RPT = 4: twelve (BITS: eight + four 4-byte) loads in flight per thread
        fvec<4> v[RPT], d[RPT];
        unsigned bt[RPT];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const long off = base + (long)(r + j * m.rstep) * C;
            v[j] = ldv(x + off, m.cv); d[j] = ldv(dout + off, m.cv); bt[j] = decisions(off);
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) accum(v[j], d[j], bt[j]);
    }
    for (; r < rend; r += m.rstep) {
        const long off = base + (long)r * C;
        accum(ldv(x + off, m.cv), ldv(dout + off, m.cv), decisions(off));
    }
    if constexpr (SMALL == 2) {                              // no LDS: one partial per WAVE, chunk index 4 k + wave of 4 nchunk (see col_partial_kernel)
        a1 = wave_cv_sum<4>(a1, CV); a2 = wave_cv_sum<4>(a2, CV); a3 = wave_cv_sum<4>(a3, CV);
        const int lane = threadIdx.x & 63;
        if (lane < CV) {
            double* p = part + (((long)blockIdx.y * nchunk + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 3 * C + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { p[k] = a1.v[k]; p[(long)C + k] = a2.v[k]; p[2L * C + k] = a3.v[k]; }
        }
        return;
    }
    if constexpr (SMALL == 1) {                              // CV <= 16: shuffle pre-reduction, one dvec per (wave, cv) through LDS (block_col_reduce_small)
        a1 = wave_cv_sum<4>(a1, CV); a2 = wave_cv_sum<4>(a2, CV); a3 = wave_cv_sum<4>(a3, CV);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane < CV) { lds[0][w * 16 + lane] = a1; lds[1][w * 16 + lane] = a2; lds[2][w * 16 + lane] = a3; }
        __syncthreads();
        if (threadIdx.x < CV) {
            double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * 3 * C + threadIdx.x * 4;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                dvec<4> sacc = dzero<4>();
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) sacc.v[k] += lds[q][wv * 16 + threadIdx.x].v[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) p[(long)q * C + k] = sacc.v[k];
            }
        }
        return;
    }
    lds[0][threadIdx.x] = a1; lds[1][threadIdx.x] = a2; lds[2][threadIdx.x] = a3;
    __syncthreads();
    if (threadIdx.x < CV) {
        double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * 3 * C + threadIdx.x * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dvec<4> sacc = dzero<4>();
            for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sacc.v[k] += lds[q][t].v[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) p[(long)q * C + k] = sacc.v[k];
        }
    }
}
// se_mlp_bwd_kernel fed by se_bn_reduce_kernel's chunk sums; also leaves bn2's per-image statistics behind: stat [2][C][N] doubles (pair_final_kernel's
// layout with "blocks" = images): stat[c][n] = sum_hw dz, stat[C + c][n] = sum_hw dz * xhat of image n.
__global__ __launch_bounds__(256) void se_bn_mlp_bwd_kernel(const float* __restrict__ h1, const float* __restrict__ w2, const float* __restrict__ w0,
                                                            float* __restrict__ dh1, float* __restrict__ dpool, int C, int R, int HW, int N,
                                                            const double* __restrict__ dpart, int nchunk, const float* __restrict__ gate,
                                                            float* __restrict__ dsc_out, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            double* __restrict__ stat) {
    __shared__ float sd[256];
    __shared__ float part[8][32];
    __shared__ float sh[32];
    const int n = blockIdx.x, t = threadIdx.x;
    double A1 = 0.0, A2 = 0.0, A3 = 0.0;
    float gt = 0.f;
    if (t < C) {
        for (int k = 0; k < nchunk; ++k) {
            const double* p = dpart + ((long)n * nchunk + k) * 3 * C + t;
            A1 += p[0]; A2 += p[C]; A3 += p[2 * C];
        }
        float v = (float)((double)gamma[t] * A2 + (double)beta[t] * A1);      // sum_hw dpre * z
        gt = gate[(long)n * C + t];
        v = v * gt * (1.f - gt);
        dsc_out[(long)n * C + t] = v;
        sd[t] = v;
    }
    __syncthreads();
    const int j = t & 31, g = t >> 5;
    float acc = 0.f;
    if (j < R) {
        int c = g;
        for (; c + 56 < C; c += 64) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w2[(long)(c + 8 * u) * R + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += sd[c + 8 * u] * wv[u];
        }
        for (; c < C; c += 8) acc += sd[c] * w2[(long)c * R + j];
    }
    part[g][j] = acc;
    __syncthreads();
    if (t < R) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += part[q][t];
        v = h1[(long)n * R + t] > 0.f ? v : 0.f;
        dh1[(long)n * R + t] = v;
        sh[t] = v;
    }
    __syncthreads();
    if (t < C) {
        float v = 0.f;
        int q = 0;
        for (; q + 7 < R; q += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w0[(long)(q + u) * C + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += sh[q + u] * wv[u];
        }
        for (; q < R; ++q) v += sh[q] * w0[(long)q * C + t];
        const float dp = v * (1.f / (float)HW);
        dpool[(long)n * C + t] = dp;
        stat[(long)t * N + n] = (double)gt * A1 + (double)HW * (double)dp;
        stat[((long)C + t) * N + n] = (double)gt * A2 + (double)dp * A3;
    }
}
// dpre = dout * (out > 0); dres = dpre; dz = dpre * s[n,c] + dpool[n,c] (se_bwd_apply_kernel's expression, never stored);
// dx = gamma invstd (dz - sum_dz / M - xhat sum_dz_xhat / M) (bn_bwd_apply_kernel's expression), as fp32 and / or piece planes
template <int PL, bool BITS>
__global__ void se_bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ outp, const float* __restrict__ x,
                                       const float* __restrict__ s, const float* __restrict__ dpool, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                       const float* __restrict__ sum_dy_xhat, float* __restrict__ dres, float* __restrict__ dx, long N, int HW, int C,
                                       unsigned short* __restrict__ dx_hi, unsigned short* __restrict__ dx_lo, int pnp, const unsigned* __restrict__ mbits) {
    const int CV = C / 4;
    const long per = (long)HW * CV, total = N * per;
    const float invn = 1.f / (float)(N * HW);
    const double dn = (double)invn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / per; const int cv = (int)(i % CV);
        const fvec<4> d = ldv(dout, i), v = ldv(x, i);
        unsigned bits;
        if (BITS) bits = relu_bits(mbits, i);
        else { const fvec<4> o = ldv(outp, i); bits = (o.v[0] > 0.f ? 1u : 0u) | (o.v[1] > 0.f ? 2u : 0u) | (o.v[2] > 0.f ? 4u : 0u) | (o.v[3] > 0.f ? 8u : 0u); }
        const fvec<4> sc = ldp<4>(s + n * C, cv), dp = ldp<4>(dpool + n * C, cv);
        const fvec<4> mu = ldp<4>(mean, cv), is = ldp<4>(invstd, cv), g = ldp<4>(gamma, cv), s1 = ldp<4>(sum_dy, cv), s2 = ldp<4>(sum_dy_xhat, cv);
        fvec<4> p, r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p.v[k] = ((bits >> k) & 1u) ? d.v[k] : 0.f;
            const float q = p.v[k] * sc.v[k] + dp.v[k];
            r.v[k] = (float)((double)g.v[k] * is.v[k] * ((double)q - s1.v[k] * dn - ((double)v.v[k] - mu.v[k]) * is.v[k] * (s2.v[k] * dn)));
        }
        if (dres != nullptr) stv(dres, i, p);               // null: the consumer adds the masked residual itself (ha2g_conv2d_dgrad_*resid*)
        if (!PL || dx != nullptr) stv(dx, i, r);
        if (PL) st_planes(dx_hi, dx_lo, pnp, i, r);
    }
}

// Weight and bias gradients of the SE excitation MLP in ONE launch (round 6; ResNetBlocks.py:84-89 under autograd):
//   dW2[c][j] += sum_n dsc[n][c] h1[n][j],  db2[c] += sum_n dsc[n][c]          (fc.2: Linear(R -> C), weight [C][R])
//   dW0[j][c] += sum_n dh1[n][j] pooled[n][c],  db0[j] += sum_n dh1[n][j]      (fc.0: Linear(C -> R), weight [R][C])
// Before: two generic weight-gradient GEMM launches per block (M, N <= 256, K = batch: < 1 MFLOP each, 33-94 us apiece in the step = 1.7 ms per
// step over the 16 blocks, profiles/r05_gemm_census.txt).  Block = 8 channels x 32 hidden units, thread (c, j) walks the batch in ascending order
// (fixed order, fp32: 128 terms); no atomics.
__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const float* __restrict__ dsc, const float* __restrict__ h1, const float* __restrict__ dh1,
                                                           const float* __restrict__ pooled, float* __restrict__ dw2, float* __restrict__ db2,
                                                           float* __restrict__ dw0, float* __restrict__ db0, int N, int C, int R) {
    // operands of 64 images at a time through LDS (coalesced loads; the first form walked the batch with four dependent global loads per image: 110 us)
    constexpr int NB = 64;
    __shared__ float s_d[NB][8], s_p[NB][8], s_h[NB][32], s_g[NB][32];
    const int t = threadIdx.x, j = t & 31, cl = t >> 5, c0 = blockIdx.x * 8, c = c0 + cl;
    const bool on = c < C && j < R;
    float a2 = 0.f, a0 = 0.f, b2 = 0.f, b0 = 0.f;
    for (int n0 = 0; n0 < N; n0 += NB) {
        const int nb = N - n0 < NB ? N - n0 : NB;
        for (int i = t; i < NB * 8; i += 256) {
            const int n = i >> 3, cc = i & 7;
            const bool ok = n < nb && c0 + cc < C;
            s_d[n][cc] = ok ? dsc[(long)(n0 + n) * C + c0 + cc] : 0.f;
            s_p[n][cc] = ok ? pooled[(long)(n0 + n) * C + c0 + cc] : 0.f;
        }
        for (int i = t; i < NB * 32; i += 256) {
            const int n = i >> 5, jj = i & 31;
            const bool ok = n < nb && jj < R;
            s_h[n][jj] = ok ? h1[(long)(n0 + n) * R + jj] : 0.f;
            s_g[n][jj] = ok ? dh1[(long)(n0 + n) * R + jj] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < NB; ++n) {                   // ascending batch order (rows past nb are zeros)
            const float d = s_d[n][cl], pl = s_p[n][cl], h = s_h[n][j], g = s_g[n][j];
            a2 = fmaf(d, h, a2); a0 = fmaf(g, pl, a0);
            b2 += d; b0 += g;
        }
        __syncthreads();
    }
    if (on) { dw2[(long)c * R + j] += a2; dw0[(long)j * C + c] += a0; }
    if (c < C && j == 0) db2[c] += b2;
    if (blockIdx.x == 0 && t < 32 && j < R) db0[j] += b0;          // channel 0's threads: b0 does not depend on c
}

// se_mlp_wgrad_kernel for up to 16 blocks of the tower in ONE launch (round 6): sixteen launches of 4-32 workgroups sat on the side queue of the tower's
// backward at 52 us each (alone: 17) between the persistent weight-gradient kernels; their operands ([N][C], [N][R]) are small enough to keep until the
// trunk's backward is enqueued.  Same arithmetic and order per block.
struct SeWgradJobs {
    const float* dsc[16]; const float* h1[16]; const float* dh1[16]; const float* pooled[16];
    float* dw2[16]; float* db2[16]; float* dw0[16]; float* db0[16];
    int C[16], R[16], start[17];
};
__global__ __launch_bounds__(256) void se_mlp_wgrad_multi_kernel(SeWgradJobs jb, int njobs, int N) {
    constexpr int NB = 64;
    __shared__ float s_d[NB][8], s_p[NB][8], s_h[NB][32], s_g[NB][32];
    int q = 0;
    while (q + 1 < njobs && (int)blockIdx.x >= jb.start[q + 1]) ++q;
    const float* __restrict__ dsc = jb.dsc[q]; const float* __restrict__ h1 = jb.h1[q]; const float* __restrict__ dh1 = jb.dh1[q];
    const float* __restrict__ pooled = jb.pooled[q];
    const int C = jb.C[q], R = jb.R[q];
    const int t = threadIdx.x, j = t & 31, cl = t >> 5, c0 = ((int)blockIdx.x - jb.start[q]) * 8, c = c0 + cl;
    const bool on = c < C && j < R;
    float a2 = 0.f, a0 = 0.f, b2 = 0.f, b0 = 0.f;
    for (int n0 = 0; n0 < N; n0 += NB) {
        const int nb = N - n0 < NB ? N - n0 : NB;
        for (int i = t; i < NB * 8; i += 256) {
            const int n = i >> 3, cc = i & 7;
            const bool ok = n < nb && c0 + cc < C;
            s_d[n][cc] = ok ? dsc[(long)(n0 + n) * C + c0 + cc] : 0.f;
            s_p[n][cc] = ok ? pooled[(long)(n0 + n) * C + c0 + cc] : 0.f;
        }
        for (int i = t; i < NB * 32; i += 256) {
            const int n = i >> 5, jj = i & 31;
            const bool ok = n < nb && jj < R;
            s_h[n][jj] = ok ? h1[(long)(n0 + n) * R + jj] : 0.f;
            s_g[n][jj] = ok ? dh1[(long)(n0 + n) * R + jj] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < NB; ++n) {
            const float d = s_d[n][cl], pl = s_p[n][cl], h = s_h[n][j], g = s_g[n][j];
            a2 = fmaf(d, h, a2); a0 = fmaf(g, pl, a0);
            b2 += d; b0 += g;
        }
        __syncthreads();
    }
    if (on) { jb.dw2[q][(long)c * R + j] += a2; jb.dw0[q][(long)j * C + c] += a0; }
    if (c < C && j == 0) jb.db2[q][c] += b2;
    if ((int)blockIdx.x == jb.start[q] && t < 32 && j < R) jb.db0[q][j] += b0;
}

static int g_small_lds = 2;         // narrow tensors (C <= 64): the backward statistics passes with shuffle pre-reduction and 2-3 KB of LDS (ha2g_bn_debug_small_lds: A/B)
static int g_colp_rpt = 4;          // rows per trip of the BatchNorm-backward statistics pass (ha2g_bn_debug_rows_per_trip: A/B of the loads in flight per wave)
inline int chunk_blocks(long rows) {
    long b = rows / 512;
    if (b < 1) b = 1;
    if (b > NB_MAX) b = NB_MAX;
    return (int)b;
}
inline int flat_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
template <typename T> inline bool okCv(int C) { constexpr int V = VW<T>::V; return C % V == 0 && C / V <= 256 && 256 % (C / V) == 0; }
inline bool okC(int C) { return okCv<float>(C); }


// ---- host side, one template per pass over the element type (float | b16) ------------------------------------------------------------------
static int pool_chunks(int N, int HW) {
    int c = 2048 / (N < 1 ? 1 : N);                      // ~2k blocks per launch
    int cap = HW / 64;                                   // >= 64 rows per block
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}
template <typename T>
int bn_stats_t(const T* x, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var, float momentum, float eps,
               float* ws, void* stream) {
    HA2G_REQUIRE(okCv<T>(C), "bn: unsupported channel count %d", C);
    HA2G_REQUIRE(rows > 0, "bn: empty batch");
    hipStream_t st = (hipStream_t)stream;
    int nb = chunk_blocks(rows);
    hipLaunchKernelGGL((col_partial_kernel<0, T>), dim3(nb), dim3(256), 0, st, x, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, rows, C,
                       (double*)ws);
    hipLaunchKernelGGL(bn_stats_final_kernel<T>, dim3(C), dim3(256), 0, st, (const double*)ws, nb, rows, C, x, mean, invstd, running_mean,
                       running_var, momentum, eps);
    HA2G_CHECK_LAUNCH("bn_stats");
    return 0;
}
template <int PL, typename T>
int bn_apply_t(const T* x, const float* mean, const float* invstd, const float* gamma, const float* beta, T* y, void* y_hi, void* y_lo, int pnp, long rows,
               int C, int act, void* stream) {
    HA2G_REQUIRE(C % VW<T>::V == 0, "bn: C %% 4");
    hipLaunchKernelGGL((bn_apply_kernel<PL, T>), dim3(flat_grid(rows * (C / VW<T>::V))), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, y,
                       rows, C, act, (unsigned short*)y_hi, (unsigned short*)y_lo, pnp);
    HA2G_CHECK_LAUNCH("bn_apply");
    return 0;
}
template <typename T>
int bn_apply_pool_t(const T* x, const float* mean, const float* invstd, const float* gamma, const float* beta, T* y, int N, int HW, int C,
                    float* pooled, float* ws, void* stream) {
    HA2G_REQUIRE(okCv<T>(C), "bn_apply_pool: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = pool_chunks(N, HW);
    hipLaunchKernelGGL(bn_apply_pool_kernel<T>, dim3(nchunk, N), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, HW, C, (double*)ws);
    hipLaunchKernelGGL(pool_final_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, (long)N * C,
                       1.f / (float)HW, pooled, (const float*)nullptr);
    HA2G_CHECK_LAUNCH("bn_apply_pool");
    return 0;
}
template <int PL, typename T>
int bn_bwd_t(const T* dy, const T* x, const float* mean, const float* invstd, const float* gamma, T* dx, void* dx_hi, void* dx_lo, int pnp, float* dgamma,
             float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta, float* ws, void* stream) {
    HA2G_REQUIRE(okCv<T>(C), "bn: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    int nb = chunk_blocks(rows);
    int nfin = nb;
    if (g_small_lds && sizeof(T) == 4 && C / VW<T>::V <= (g_small_lds == 3 ? 16 : 8))       // C = 32 (mode 3: C <= 64): the small-LDS form (see block_col_reduce_small)
        hipLaunchKernelGGL((col_partial_kernel<1, T, 2, 1>), dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    else if (g_small_lds == 1 && sizeof(T) == 4 && C / VW<T>::V <= 64) {      // wider (cv == lane needs C / 4 <= 64): one partial per WAVE, no LDS
        hipLaunchKernelGGL((col_partial_kernel<1, T, 2, 2>), dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
        nfin = 4 * nb;
    } else if (g_colp_rpt == 8) hipLaunchKernelGGL((col_partial_kernel<1, T, 8>), dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    else if (g_colp_rpt == 2) hipLaunchKernelGGL((col_partial_kernel<1, T, 2>), dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    else hipLaunchKernelGGL((col_partial_kernel<1, T>), dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    hipLaunchKernelGGL(pair_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)ws, nfin, C, dbeta, dgamma, acc_dbeta, acc_dgamma);
    if (PL || dx)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<PL, T>), dim3(flat_grid(rows * (C / VW<T>::V))), dim3(256), 0, st, dy, x, mean, invstd, gamma,
                           (const float*)dbeta, (const float*)dgamma, dx, rows, C, relu_mask, (unsigned short*)dx_hi, (unsigned short*)dx_lo, pnp);
    HA2G_CHECK_LAUNCH("bn_bwd");
    return 0;
}
template <int PL, typename T>
int se_scale_add_relu_t(const T* x, const float* s, const T* res, T* out, void* o_hi, void* o_lo, int pnp, int N, int HW, int C, void* stream,
                        const BnAff* aff = nullptr, unsigned* mbits = nullptr) {
    HA2G_REQUIRE(C % VW<T>::V == 0, "se: C %% 4");
    HA2G_REQUIRE(mbits == nullptr || (aff != nullptr && sizeof(T) == 4 && C % 32 == 0), "se: the decision bits need fp32 storage, C %% 32 == 0 and the on-the-fly bn2 form");
    if (aff)
        hipLaunchKernelGGL((se_scale_add_relu_kernel<PL, T, true>), dim3(flat_grid((long)N * HW * (C / VW<T>::V))), dim3(256), 0, (hipStream_t)stream, x, s, res,
                           out, (long)N, HW, C, (unsigned short*)o_hi, (unsigned short*)o_lo, pnp, *aff, mbits);
    else
    hipLaunchKernelGGL((se_scale_add_relu_kernel<PL, T>), dim3(flat_grid((long)N * HW * (C / VW<T>::V))), dim3(256), 0, (hipStream_t)stream, x, s, res, out,
                       (long)N, HW, C, (unsigned short*)o_hi, (unsigned short*)o_lo, pnp, BnAff{});
    HA2G_CHECK_LAUNCH("se_scale_add_relu");
    return 0;
}
template <typename T>
int se_bwd_scale_t(const T* dout, const T* out, const T* x, float* ds, int N, int HW, int C, const float* gate, float* ws, void* stream,
                   const BnAff* aff = nullptr) {
    HA2G_REQUIRE(okCv<T>(C), "se: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ws ? pool_chunks(N, HW) : 1;          // ws: ha2g_bn_apply_pool_workspace_floats(N, HW, C) floats, or null
    if (aff) {
        if (nchunk > 1) {
            hipLaunchKernelGGL((image_col_kernel<1, T, true>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)ws, (const float*)nullptr, *aff);
            hipLaunchKernelGGL(pool_final_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, (long)N * C, 1.f, ds, gate);
        } else {
            hipLaunchKernelGGL((image_col_kernel<1, T, true>), dim3(1, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)nullptr, gate, *aff);
        }
        HA2G_CHECK_LAUNCH("se_bwd_scale_bn");
        return 0;
    }
    if (nchunk > 1) {
        hipLaunchKernelGGL((image_col_kernel<1, T>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)ws, (const float*)nullptr);
        hipLaunchKernelGGL(pool_final_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, (long)N * C, 1.f, ds,
                           gate);
    } else {
        hipLaunchKernelGGL((image_col_kernel<1, T>), dim3(1, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)nullptr, gate);
    }
    HA2G_CHECK_LAUNCH("se_bwd_scale");
    return 0;
}
template <typename T>
int se_bwd_apply_t(const T* dout, const T* out, const float* s, const float* dpool, T* dres, T* dx, int N, int HW, int C, void* stream) {
    HA2G_REQUIRE(C % VW<T>::V == 0, "se: C %% 4");
    hipLaunchKernelGGL(se_bwd_apply_kernel<T>, dim3(flat_grid((long)N * HW * (C / VW<T>::V))), dim3(256), 0, (hipStream_t)stream, dout, out, s, dpool, dres, dx,
                       (long)N, HW, C);
    HA2G_CHECK_LAUNCH("se_bwd_apply");
    return 0;
}

// bf16 <-> fp32 streams of the bf16-storage mode's boundaries (taps, tap gradients)
__global__ void f32_to_b16_kernel(const float* __restrict__ x, b16* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st4(y, i, ld4(x, i));
}
__global__ void b16_to_f32_kernel(const b16* __restrict__ x, float* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st4(y, i, ld4(x, i));
}
// out = bf16(a + b): a bf16 (nullable: out = bf16(b)), b fp32
__global__ void add_f32_to_b16_kernel(const b16* __restrict__ a, const float* __restrict__ b, b16* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = ld4(b, i);
        if (a) { const float4 w = ld4(a, i); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        st4(out, i, v);
    }
}

}  // namespace

extern "C" {

// floats of scratch needed by the column reductions below
long ha2g_bn_workspace_floats(int C) { return (long)NB_MAX * 2 * C * 2; }   /* partials are doubles */

// mean/invstd [C] out; running_mean/var updated in place when non-null.  x is [rows][C], C in {4,8,...,1024} with 256 % (C/4) == 0.
// mean / invstd / running statistics from per-block partial sums [2][C][nblk] (sum, sum of squares; doubles) that a producer's epilogue left
// behind (ha2g_conv2d_fwd_planes_np_stats_f32): the second half of ha2g_bn_stats_f32 without its pass over the tensor.  nn.BatchNorm2d in training
// mode (ResNetBlocks.py:27-28,34): biased variance for the normalisation, unbiased for running_var, momentum 0.1.
int ha2g_bn_stats_finalize_f32(const void* part, int nblk, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                               float momentum, float eps, void* stream) {
    HA2G_REQUIRE(part != nullptr && nblk > 0 && rows > 0 && C > 0, "bn_stats_finalize: empty input");
    hipLaunchKernelGGL(bn_stats_final_kernel<float>, dim3(C), dim3(256), 0, (hipStream_t)stream, (const double*)part, nblk, rows, C,
                       (const float*)nullptr, mean, invstd, running_mean, running_var, momentum, eps);
    HA2G_CHECK_LAUNCH("bn_stats_finalize");
    return 0;
}
int ha2g_bn_stats_f32(const float* x, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                      float momentum, float eps, float* ws, void* stream) {
    return bn_stats_t<float>(x, rows, C, mean, invstd, running_mean, running_var, momentum, eps, ws, stream);
}
int ha2g_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                      long rows, int C, int act, void* stream) {
    return bn_apply_t<0, float>(x, mean, invstd, gamma, beta, y, nullptr, nullptr, 0, rows, C, act, stream);
}
// ha2g_bn_apply_f32 that also writes y as bf16 planes y_hi / y_lo [rows][C]
int ha2g_bn_apply_planes_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y, void* y_hi,
                             void* y_lo, long rows, int C, int act, void* stream) {
    HA2G_REQUIRE(y_hi != nullptr && y_lo != nullptr, "bn_apply_planes: null plane");
    return bn_apply_t<1, float>(x, mean, invstd, gamma, beta, y, y_hi, y_lo, 2, rows, C, act, stream);
}
// np = 2 or 3 equally spaced piece planes: piece q at planes + q * ps (elements)
int ha2g_bn_apply_planes_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y, void* planes,
                                long ps, int np, long rows, int C, int act, void* stream) {
    HA2G_REQUIRE(planes != nullptr && (np == 2 || np == 3), "bn_apply_planes_np: null plane / np = %d", np);
    return bn_apply_t<1, float>(x, mean, invstd, gamma, beta, y, planes, (unsigned short*)planes + ps, np, rows, C, act, stream);
}
// y = bn(x) for x [N][HW][C] AND pooled[n][c] = mean over HW of y (the SE squeeze) in one pass over the tensor.
// ws: >= ha2g_bn_apply_pool_workspace_floats(N, HW, C) floats.
long ha2g_bn_apply_pool_workspace_floats(int N, int HW, int C) { return (long)N * pool_chunks(N, HW) * C * 2; }
int ha2g_bn_apply_pool_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                           int N, int HW, int C, float* pooled, float* ws, void* stream) {
    return bn_apply_pool_t<float>(x, mean, invstd, gamma, beta, y, N, HW, C, pooled, ws, stream);
}
// dgamma = sum dy*xhat, dbeta = sum dy, dx as torch's batch-norm backward (train mode)
// relu_mask = 1: x is the output of a ReLU that precedes the BatchNorm; dx then is the gradient w.r.t. the ReLU's INPUT
int ha2g_bn_bwd_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx,
                    float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta, float* ws,
                    void* stream) {
    return bn_bwd_t<0, float>(dy, x, mean, invstd, gamma, dx, nullptr, nullptr, 0, dgamma, dbeta, rows, C, relu_mask, acc_dgamma, acc_dbeta, ws, stream);
}
// ha2g_bn_bwd_f32 whose dx goes out as bf16 planes dx_hi / dx_lo [rows][C] (and, when dx != NULL, in fp32 as well)
int ha2g_bn_bwd_planes_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* dx_hi,
                           void* dx_lo, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                           float* ws, void* stream) {
    HA2G_REQUIRE(dx_hi != nullptr && dx_lo != nullptr, "bn_bwd_planes: null plane");
    return bn_bwd_t<1, float>(dy, x, mean, invstd, gamma, dx, dx_hi, dx_lo, 2, dgamma, dbeta, rows, C, relu_mask, acc_dgamma, acc_dbeta, ws, stream);
}
int ha2g_bn_bwd_planes_np_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* planes,
                              long ps, int np, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                              float* ws, void* stream) {
    HA2G_REQUIRE(planes != nullptr && (np == 2 || np == 3), "bn_bwd_planes_np: null plane / np = %d", np);
    return bn_bwd_t<1, float>(dy, x, mean, invstd, gamma, dx, planes, (unsigned short*)planes + ps, np, dgamma, dbeta, rows, C, relu_mask, acc_dgamma,
                              acc_dbeta, ws, stream);
}
// BatchNorm backward whose statistics pass already happened: stat_part [2][C][stat_nblk] doubles = tile sums of dy and of dy * xhat left behind by the
// producer of dy (ha2g_conv2d_dgrad_planes_np_bnstats_f32).  pair_final adds the tiles in order (-> dbeta, dgamma, accumulated into acc_* when given), then
// the apply pass as in ha2g_bn_bwd_planes_np_f32 (planes may be NULL: fp32 dx only).
int ha2g_bn_bwd_planes_np_partials_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* planes,
                                       long ps, int np, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                                       const void* stat_part, int stat_nblk, void* stream) {
    HA2G_REQUIRE(okCv<float>(C), "bn: unsupported channel count %d", C);
    HA2G_REQUIRE(stat_part != nullptr && stat_nblk > 0, "bn_bwd_partials: no partial sums");
    HA2G_REQUIRE(planes == nullptr || np == 2 || np == 3, "bn_bwd_partials: np = %d", np);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)stat_part, stat_nblk, C, dbeta, dgamma, acc_dbeta, acc_dgamma);
    if (planes != nullptr)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1, float>), dim3(flat_grid(rows * (C / 4))), dim3(256), 0, st, dy, x, mean, invstd, gamma, (const float*)dbeta,
                           (const float*)dgamma, dx, rows, C, relu_mask, (unsigned short*)planes, (unsigned short*)planes + ps, np);
    else if (dx != nullptr)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<0, float>), dim3(flat_grid(rows * (C / 4))), dim3(256), 0, st, dy, x, mean, invstd, gamma, (const float*)dbeta,
                           (const float*)dgamma, dx, rows, C, relu_mask, (unsigned short*)nullptr, (unsigned short*)nullptr, 0);
    HA2G_CHECK_LAUNCH("bn_bwd_partials");
    return 0;
}
// out[n][c] = mean over HW of x[n][hw][c]
int ha2g_hw_mean_f32(const float* x, float* out, int N, int HW, int C, void* stream) {
    HA2G_REQUIRE(okC(C), "hw_mean: unsupported channel count %d", C);
    hipLaunchKernelGGL((image_col_kernel<0, float>), dim3(1, N), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, (const float*)nullptr, HW, C,
                       out, 1.f / (float)HW, (double*)nullptr, (const float*)nullptr);
    HA2G_CHECK_LAUNCH("hw_mean");
    return 0;
}
int ha2g_se_scale_add_relu_f32(const float* x, const float* s, const float* res, float* out, int N, int HW, int C, void* stream) {
    return se_scale_add_relu_t<0, float>(x, s, res, out, nullptr, nullptr, 0, N, HW, C, stream);
}
// the same, out also as bf16 planes (the next block's conv1 reads them in its weight gradient)
int ha2g_se_scale_add_relu_planes_f32(const float* x, const float* s, const float* res, float* out, void* o_hi, void* o_lo, int N, int HW, int C,
                                      void* stream) {
    HA2G_REQUIRE(o_hi != nullptr && o_lo != nullptr, "se_scale_add_relu_planes: null plane");
    return se_scale_add_relu_t<1, float>(x, s, res, out, o_hi, o_lo, 2, N, HW, C, stream);
}
int ha2g_se_scale_add_relu_planes_np_f32(const float* x, const float* s, const float* res, float* out, void* planes, long ps, int np, int N, int HW,
                                         int C, void* stream) {
    HA2G_REQUIRE(planes != nullptr && (np == 2 || np == 3), "se_scale_add_relu_planes_np: null plane / np = %d", np);
    return se_scale_add_relu_t<1, float>(x, s, res, out, planes, (unsigned short*)planes + ps, np, N, HW, C, stream);
}
// The SE tail of a block whose bn2 statistics came out of conv2's epilogue (ResNetBlocks.py:29-36,81-95): bn2's output b2 is never materialised.
//   ha2g_bn_pool_from_partials_f32: the squeeze mean_hw b2 from the per-tile column sums of c2 (stat_part of ha2g_conv2d_fwd_planes_np_stats_f32 when its
//     tiles lie inside one image: nblk = N * tiles per image, ha2g_conv2d_fwd_planes_stat_tiles_per_image > 0);
//   ha2g_se_bn_scale_add_relu_np_f32: out = relu(bn2(c2) * s + res) (+ np piece planes of out, np = 0: none) -- bn2(c2) with ha2g_bn_apply_pool_f32's expression;
//   ha2g_se_bwd_scale_bn_f32: ds[n][c] = sum_hw dout (out > 0) bn2(c2), the gate's sigmoid' folded in as in ha2g_se_bwd_scale_f32.
int ha2g_bn_pool_from_partials_f32(const void* part, int nblk, int N, int HW, int C, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* pooled, void* stream) {
    HA2G_REQUIRE(part != nullptr && N > 0 && nblk > 0 && nblk % N == 0 && HW > 0, "bn_pool_from_partials: %d blocks do not split over %d images", nblk, N);
    hipLaunchKernelGGL(pool_from_partials_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)part, nblk, nblk / N, C,
                       (long)N * C, 1.0 / (double)HW, mean, invstd, gamma, beta, pooled);
    HA2G_CHECK_LAUNCH("bn_pool_from_partials");
    return 0;
}
int ha2g_se_bn_scale_add_relu_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* s,
                                     const float* res, float* out, void* planes, long ps, int np, int N, int HW, int C, void* stream) {
    HA2G_REQUIRE(np == 0 || (planes != nullptr && (np == 2 || np == 3)), "se_bn_scale_add_relu_np: null plane / np = %d", np);
    const BnAff aff{mean, invstd, gamma, beta};
    if (np == 0) return se_scale_add_relu_t<0, float>(x, s, res, out, nullptr, nullptr, 0, N, HW, C, stream, &aff);
    return se_scale_add_relu_t<1, float>(x, s, res, out, planes, (unsigned short*)planes + ps, np, N, HW, C, stream, &aff);
}
// ... and leaves the ReLU decisions (out > 0) behind as bits: mask_bits = N * HW * C / 32 32-bit words (C % 32 == 0), for ha2g_se_bn_bwd_*'s mask_bits
int ha2g_se_bn_scale_add_relu_mask_np_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* s,
                                          const float* res, float* out, void* planes, long ps, int np, int N, int HW, int C, void* mask_bits, void* stream) {
    HA2G_REQUIRE(np == 0 || (planes != nullptr && (np == 2 || np == 3)), "se_bn_scale_add_relu_mask_np: null plane / np = %d", np);
    HA2G_REQUIRE(mask_bits != nullptr, "se_bn_scale_add_relu_mask_np: null bit buffer");
    const BnAff aff{mean, invstd, gamma, beta};
    if (np == 0) return se_scale_add_relu_t<0, float>(x, s, res, out, nullptr, nullptr, 0, N, HW, C, stream, &aff, (unsigned*)mask_bits);
    return se_scale_add_relu_t<1, float>(x, s, res, out, planes, (unsigned short*)planes + ps, np, N, HW, C, stream, &aff, (unsigned*)mask_bits);
}
int ha2g_se_bwd_scale_bn_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, void* stream) {
    const BnAff aff{mean, invstd, gamma, beta};
    return se_bwd_scale_t<float>(dout, out, x, ds, N, HW, C, gate, ws, stream, &aff);
}
// ds[n][c] = sum_hw dout*(out>0)*x
int ha2g_se_bwd_scale_f32(const float* dout, const float* out, const float* x, float* ds, int N, int HW, int C, const float* gate, float* ws,
                          void* stream) {
    return se_bwd_scale_t<float>(dout, out, x, ds, N, HW, C, gate, ws, stream);
}
// ha2g_se_bwd_scale[_bn]_f32 + ha2g_se_mlp_bwd_f32 with the reduction's final pass folded into the MLP launch (round 6): ds [N][C] is still written (the
// excitation MLP's weight gradient reads it).  mean == NULL: x is bn2's materialised output; else bn2 is applied on the fly to its input x.
int ha2g_se_bwd_scale_mlp_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, const float* h1, const float* w2,
                              const float* w0, float* dh1, float* dpool, int R, void* stream) {
    HA2G_REQUIRE(okCv<float>(C), "se: unsupported channel count %d", C);
    HA2G_REQUIRE(C >= 1 && C <= 256 && R >= 1 && R <= 32, "se_bwd_scale_mlp: unsupported widths C = %d, R = %d", C, R);
    HA2G_REQUIRE(ws != nullptr && gate != nullptr, "se_bwd_scale_mlp: null workspace / gate");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = pool_chunks(N, HW);
    if (mean != nullptr) {
        const BnAff aff{mean, invstd, gamma, beta};
        hipLaunchKernelGGL((image_col_kernel<1, float, true>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)ws, (const float*)nullptr, aff);
    } else {
        hipLaunchKernelGGL((image_col_kernel<1, float>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)ws, (const float*)nullptr);
    }
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(N), dim3(256), 0, st, (const float*)nullptr, h1, w2, w0, dh1, dpool, C, R, 1.f / (float)HW, (const double*)ws, nchunk,
                       gate, ds);
    HA2G_CHECK_LAUNCH("se_bwd_scale_mlp");
    return 0;
}
int ha2g_se_bwd_apply_f32(const float* dout, const float* out, const float* s, const float* dpool, float* dres, float* dx, int N,
                          int HW, int C, void* stream) {
    return se_bwd_apply_t<float>(dout, out, s, dpool, dres, dx, N, HW, C, stream);
}

// per-image statistics partials of x [N][HW][C] for ha2g_bn_stats_finalize_f32 + ha2g_se_mlp_fwd_f32 (see bn_image_partial_kernel): part [2][C][nblk] doubles
// with nblk = N * ha2g_bn_image_partial_chunks(N, HW)
int ha2g_bn_image_partial_chunks(int N, int HW) { return pool_chunks(N, HW); }
int ha2g_bn_image_partials_f32(const float* x, int N, int HW, int C, double* part, void* stream) {
    HA2G_REQUIRE(okCv<float>(C), "bn_image_partials: unsupported channel count %d", C);
    HA2G_REQUIRE(part != nullptr && x != nullptr && N >= 0 && HW > 0, "bn_image_partials: null buffer / empty image");
    if (N == 0) return 0;
    hipLaunchKernelGGL(bn_image_partial_kernel, dim3(pool_chunks(N, HW), N), dim3(256), 0, (hipStream_t)stream, x, HW, C, part);
    HA2G_CHECK_LAUNCH("bn_image_partials");
    return 0;
}
// ha2g_se_mlp_wgrad_f32 for n <= 16 blocks in one launch: HOST arrays of n device pointers / widths; every job's batch is N images
int ha2g_se_mlp_wgrad_multi_f32(int n, const void* const* dsc, const void* const* h1, const void* const* dh1, const void* const* pooled, void* const* dw2,
                                void* const* db2, void* const* dw0, void* const* db0, const int* C, const int* R, int N, void* stream) {
    HA2G_REQUIRE(n >= 0 && n <= 16, "se_mlp_wgrad_multi: %d jobs (max 16)", n);
    if (n == 0 || N == 0) return 0;
    SeWgradJobs jb{};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        HA2G_REQUIRE(C[i] >= 1 && C[i] <= 256 && R[i] >= 1 && R[i] <= 32, "se_mlp_wgrad_multi: unsupported widths C = %d, R = %d", C[i], R[i]);
        jb.dsc[i] = (const float*)dsc[i]; jb.h1[i] = (const float*)h1[i]; jb.dh1[i] = (const float*)dh1[i]; jb.pooled[i] = (const float*)pooled[i];
        jb.dw2[i] = (float*)dw2[i]; jb.db2[i] = (float*)db2[i]; jb.dw0[i] = (float*)dw0[i]; jb.db0[i] = (float*)db0[i];
        jb.C[i] = C[i]; jb.R[i] = R[i]; jb.start[i] = total;
        total += ceil_div(C[i], 8);
    }
    jb.start[n] = total;
    hipLaunchKernelGGL(se_mlp_wgrad_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, jb, n, N);
    HA2G_CHECK_LAUNCH("se_mlp_wgrad_multi");
    return 0;
}
// ---- SE backward + bn2 backward in two passes (round 6; see se_bn_reduce_kernel) ----
static int g_sebn_rpt = 4, g_sebn_chunk_shift = 0;
// A/B of the reduction pass: rows per trip (4 | 2: 126 | fewer registers per wave) and chunks per image halved `chunk_shift` times (longer loops per block)
void ha2g_bn_debug_small_lds(int on) { g_small_lds = on < 0 ? 0 : (on > 3 ? 3 : on); }      // 0 classic; 1 = C = 32 small LDS, wider: LDS-free; 2 (default) = C = 32 only; 3 = C <= 64
void ha2g_se_bn_debug(int rpt, int chunk_shift) { g_sebn_rpt = rpt == 2 ? 2 : 4; g_sebn_chunk_shift = chunk_shift < 0 ? 0 : (chunk_shift > 4 ? 4 : chunk_shift); }
// floats of workspace ha2g_se_bn_bwd_reduce_mlp_f32 needs
long ha2g_se_bn_bwd_workspace_floats(int N, int HW, int C) { return (long)N * pool_chunks(N, HW) * 4 * 3 * C * 2; }      // (one partial per wave in the LDS-free form)
// Reduction pass + excitation MLP backward.  x = bn2's INPUT (conv2's output) [N][HW][C], mean / invstd / gamma / beta = bn2's; gate = the SE gate [N][C];
// writes ds [N][C] (the MLP's weight gradient reads it), dh1 [N][R], dpool [N][C] (already / HW) and stat [2][C][N] doubles: bn2's per-image backward sums.
int ha2g_se_bn_bwd_reduce_mlp_f32(const float* dout, const float* out, const float* x, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* ds, int N, int HW, int C, const float* gate, float* ws, const float* h1, const float* w2,
                                  const float* w0, float* dh1, float* dpool, int R, double* stat, const void* mask_bits, void* stream) {
    HA2G_REQUIRE(okCv<float>(C), "se_bn_bwd: unsupported channel count %d", C);
    HA2G_REQUIRE(mask_bits == nullptr ? out != nullptr : C % 32 == 0, "se_bn_bwd: the ReLU decisions come from `out` or from mask_bits (C %% 32 == 0)");
    HA2G_REQUIRE(C >= 1 && C <= 256 && R >= 1 && R <= 32, "se_bn_bwd: unsupported widths C = %d, R = %d", C, R);
    HA2G_REQUIRE(ws != nullptr && gate != nullptr && stat != nullptr && mean != nullptr && invstd != nullptr && gamma != nullptr && beta != nullptr,
                 "se_bn_bwd: null workspace / gate / statistics / BatchNorm parameter");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nchunk = pool_chunks(N, HW) >> g_sebn_chunk_shift;
    if (nchunk < 1) nchunk = 1;
    int nfin = nchunk;
    if (mask_bits != nullptr && g_small_lds && C / 4 <= (g_small_lds == 3 ? 16 : 8))
        hipLaunchKernelGGL((se_bn_reduce_kernel<true, 2, 1>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, (double*)ws, mean, invstd, (const unsigned*)mask_bits);
    else if (mask_bits != nullptr && g_small_lds == 1 && C / 4 <= 64) {
        hipLaunchKernelGGL((se_bn_reduce_kernel<true, 2, 2>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, (double*)ws, mean, invstd, (const unsigned*)mask_bits);
        nfin = 4 * nchunk;
    } else if (mask_bits != nullptr && g_sebn_rpt == 2)
        hipLaunchKernelGGL((se_bn_reduce_kernel<true, 2>), dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, (double*)ws, mean, invstd, (const unsigned*)mask_bits);
    else if (mask_bits != nullptr)
        hipLaunchKernelGGL(se_bn_reduce_kernel<true>, dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, (double*)ws, mean, invstd, (const unsigned*)mask_bits);
    else
        hipLaunchKernelGGL(se_bn_reduce_kernel<false>, dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, (double*)ws, mean, invstd, (const unsigned*)nullptr);
    hipLaunchKernelGGL(se_bn_mlp_bwd_kernel, dim3(N), dim3(256), 0, st, h1, w2, w0, dh1, dpool, C, R, HW, N, (const double*)ws, nfin, gate, ds, gamma, beta, stat);
    HA2G_CHECK_LAUNCH("se_bn_bwd_reduce_mlp");
    return 0;
}
// Apply pass: dres = dout * (out > 0) (the residual branch's gradient) and bn2's data gradient dx (fp32, nullable when planes are given) and / or its
// np piece planes (planes, piece stride ps elements; null: fp32 only); dgamma / dbeta [C] = bn2's parameter gradients (fresh sums; acc_*: also added there).
int ha2g_se_bn_bwd_apply_np_f32(const float* dout, const float* out, const float* x, const float* s, const float* dpool, const float* mean,
                                const float* invstd, const float* gamma, float* dres, float* dx, void* planes, long ps, int np, float* dgamma, float* dbeta,
                                float* acc_dgamma, float* acc_dbeta, const double* stat, int N, int HW, int C, const void* mask_bits, void* stream) {
    HA2G_REQUIRE(okCv<float>(C), "se_bn_bwd: unsupported channel count %d", C);
    HA2G_REQUIRE(mask_bits == nullptr ? out != nullptr : C % 32 == 0, "se_bn_bwd_apply: the ReLU decisions come from `out` or from mask_bits (C %% 32 == 0)");
    HA2G_REQUIRE(planes == nullptr || np == 2 || np == 3, "se_bn_bwd_apply: np = %d", np);
    HA2G_REQUIRE(planes != nullptr || dx != nullptr, "se_bn_bwd_apply: no output");
    HA2G_REQUIRE(stat != nullptr && dgamma != nullptr && dbeta != nullptr, "se_bn_bwd_apply: null buffer");
    HA2G_REQUIRE(dres != nullptr || mask_bits != nullptr, "se_bn_bwd_apply: dres may be omitted only where the consumer holds the decision bits");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, stat, N, C, dbeta, dgamma, acc_dbeta, acc_dgamma);
    const dim3 grid(flat_grid((long)N * HW * (C / 4)));
    const unsigned* mb = (const unsigned*)mask_bits;
    unsigned short* ph = (unsigned short*)planes;
#define HA2G_SEBN_APPLY(PL, BITS) hipLaunchKernelGGL((se_bn_bwd_apply_kernel<PL, BITS>), grid, dim3(256), 0, st, dout, out, x, s, dpool, mean, invstd, gamma, \
        (const float*)dbeta, (const float*)dgamma, dres, dx, (long)N, HW, C, ph, ph ? ph + ps : ph, planes ? np : 0, mb)
    if (planes != nullptr) { if (mb) HA2G_SEBN_APPLY(1, true); else HA2G_SEBN_APPLY(1, false); }
    else { if (mb) HA2G_SEBN_APPLY(0, true); else HA2G_SEBN_APPLY(0, false); }
#undef HA2G_SEBN_APPLY
    HA2G_CHECK_LAUNCH("se_bn_bwd_apply");
    return 0;
}

// dh1 [N][R], dpool [N][C] from dsc [N][C], h1 [N][R], fc.2.weight w2 [C][R], fc.0.weight w0 [R][C]; C <= 256, R <= 32
// FORWARD of the SE excitation MLP in one launch (ResNetBlocks.py:84-89): per image n
//   pooled[n][c] = the squeeze -- given, or taken from the per-tile column sums of bn2's INPUT as in pool_from_partials_kernel
//   h1[n][j]     = relu(b0[j] + sum_c pooled[n][c] w0[j][c])            (fc.0: Linear(C -> R), weight [R][C])
//   sc[n][c]     = sigmoid(b2[c] + sum_j h1[n][j] w2[c][j])             (fc.2: Linear(R -> C), weight [C][R]; the gate)
// Two GEMM launches of a few microseconds (+ the squeeze kernel) sat between conv2 and the block's tail on the forward's critical path, per block.
// Fixed summation order (eight strided partials per j, then ascending; ascending over j), no atomics.
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ pooled_in, const double* __restrict__ part, int nblk, int gpi,
                                                         double inv_hw, BnAff aff, const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ pooled_out,
                                                         float* __restrict__ h1, float* __restrict__ sc, int C, int R) {
    __shared__ float sp[256];
    __shared__ float part8[8][32];
    __shared__ float sh[32];
    const int n = blockIdx.x, t = threadIdx.x;
    if (t < C) {
        float v;
        if (part != nullptr) {
            const double* p = part + (long)t * nblk + (long)n * gpi;
            double s = 0.0;
            for (int k = 0; k < gpi; ++k) s += p[k];
            v = (float)((s * inv_hw - (double)aff.mean[t]) * (double)aff.invstd[t] * (double)aff.gamma[t] + (double)aff.beta[t]);
            pooled_out[(long)n * C + t] = v;
        } else {
            v = pooled_in[(long)n * C + t];
        }
        sp[t] = v;
    }
    __syncthreads();
    const int j = t & 31, g = t >> 5;
    float acc = 0.f;
    if (j < R)
        for (int c = g; c < C; c += 8) acc += sp[c] * w0[(long)j * C + c];
    part8[g][j] = acc;
    __syncthreads();
    if (t < R) {
        float v = b0[t];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += part8[q][t];
        v = fmaxf(v, 0.f);
        h1[(long)n * R + t] = v;
        sh[t] = v;
    }
    __syncthreads();
    if (t < C) {
        float v = b2[t];
        for (int q = 0; q < R; ++q) v += sh[q] * w2[(long)t * R + q];
        sc[(long)n * C + t] = 1.0f / (1.0f + expf(-v));
    }
}

void ha2g_bn_debug_rows_per_trip(int n) { g_colp_rpt = n; }
int ha2g_se_mlp_bwd_supported(int C, int R) { return C >= 1 && C <= 256 && R >= 1 && R <= 32; }
int ha2g_se_mlp_bwd_f32(const float* dsc, const float* h1, const float* w2, const float* w0, float* dh1, float* dpool, int N, int C, int R,
                        float inv_hw, void* stream) {
    HA2G_REQUIRE(ha2g_se_mlp_bwd_supported(C, R), "se_mlp_bwd: unsupported widths C = %d, R = %d", C, R);
    if (N == 0) return 0;
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dsc, h1, w2, w0, dh1, dpool, C, R, inv_hw);
    HA2G_CHECK_LAUNCH("se_mlp_bwd");
    return 0;
}

int ha2g_se_mlp_wgrad_f32(const float* dsc, const float* h1, const float* dh1, const float* pooled, float* dw2, float* db2, float* dw0, float* db0,
                          int N, int C, int R, void* stream) {
    HA2G_REQUIRE(ha2g_se_mlp_bwd_supported(C, R), "se_mlp_wgrad: unsupported widths C = %d, R = %d", C, R);
    if (N == 0) return 0;
    hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, dsc, h1, dh1, pooled, dw2, db2, dw0, db0, N, C, R);
    HA2G_CHECK_LAUNCH("se_mlp_wgrad");
    return 0;
}

// Exactly one of pooled_in [N][C] / stat_part (with nblk = N * tiles per image, HW and bn2's mean / invstd / gamma / beta; pooled_out receives the
// squeeze) is given.  h1 [N][R] and sc [N][C] are outputs.
int ha2g_se_mlp_fwd_f32(const float* pooled_in, const void* stat_part, int nblk, int HW, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* w0, const float* b0, const float* w2, const float* b2, float* pooled_out, float* h1,
                        float* sc, int N, int C, int R, void* stream) {
    HA2G_REQUIRE(ha2g_se_mlp_bwd_supported(C, R), "se_mlp_fwd: unsupported widths C = %d, R = %d", C, R);
    HA2G_REQUIRE((pooled_in != nullptr) != (stat_part != nullptr), "se_mlp_fwd: exactly one of pooled_in / stat_part");
    HA2G_REQUIRE(stat_part == nullptr || (N > 0 && nblk > 0 && nblk % N == 0 && HW > 0 && pooled_out != nullptr && mean && invstd && gamma && beta),
                 "se_mlp_fwd: %d statistics blocks do not split over %d images / null BatchNorm parameter", nblk, N);
    if (N == 0) return 0;
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pooled_in, (const double*)stat_part, nblk, stat_part ? nblk / N : 0,
                       HW > 0 ? 1.0 / (double)HW : 0.0, BnAff{mean, invstd, gamma, beta}, w0, b0, w2, b2, pooled_out, h1, sc, C, R);
    HA2G_CHECK_LAUNCH("se_mlp_fwd");
    return 0;
}

// ---- bf16-storage mode (BASELINE config 5): the same passes over bf16 tensors.  Statistics, scales, gradients of gamma / beta and all
// arithmetic are fp32 / double as above; only the activation / activation-gradient streams are 2 bytes per element. ------------------------------
int ha2g_bn_stats_b16(const void* x, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var, float momentum,
                      float eps, float* ws, void* stream) {
    return bn_stats_t<b16>((const b16*)x, rows, C, mean, invstd, running_mean, running_var, momentum, eps, ws, stream);
}
int ha2g_bn_apply_b16(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y, long rows, int C,
                      int act, void* stream) {
    return bn_apply_t<0, b16>((const b16*)x, mean, invstd, gamma, beta, (b16*)y, nullptr, nullptr, 0, rows, C, act, stream);
}
int ha2g_bn_apply_pool_b16(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y, int N, int HW,
                           int C, float* pooled, float* ws, void* stream) {
    return bn_apply_pool_t<b16>((const b16*)x, mean, invstd, gamma, beta, (b16*)y, N, HW, C, pooled, ws, stream);
}
int ha2g_bn_bwd_b16(const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma, void* dx, float* dgamma,
                    float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta, float* ws, void* stream) {
    return bn_bwd_t<0, b16>((const b16*)dy, (const b16*)x, mean, invstd, gamma, (b16*)dx, nullptr, nullptr, 0, dgamma, dbeta, rows, C, relu_mask,
                            acc_dgamma, acc_dbeta, ws, stream);
}
int ha2g_se_scale_add_relu_b16(const void* x, const float* s, const void* res, void* out, int N, int HW, int C, void* stream) {
    return se_scale_add_relu_t<0, b16>((const b16*)x, s, (const b16*)res, (b16*)out, nullptr, nullptr, 0, N, HW, C, stream);
}
int ha2g_se_bwd_scale_b16(const void* dout, const void* out, const void* x, float* ds, int N, int HW, int C, const float* gate, float* ws,
                          void* stream) {
    return se_bwd_scale_t<b16>((const b16*)dout, (const b16*)out, (const b16*)x, ds, N, HW, C, gate, ws, stream);
}
int ha2g_se_bwd_apply_b16(const void* dout, const void* out, const float* s, const float* dpool, void* dres, void* dx, int N, int HW, int C,
                          void* stream) {
    return se_bwd_apply_t<b16>((const b16*)dout, (const b16*)out, s, dpool, (b16*)dres, (b16*)dx, N, HW, C, stream);
}
// y (bf16) = round-to-nearest-even(x); n % 4 == 0
int ha2g_f32_to_b16(const float* x, void* y, long n, void* stream) {
    HA2G_REQUIRE(n % 4 == 0, "f32_to_b16: n %% 4");
    if (n == 0) return 0;
    hipLaunchKernelGGL(f32_to_b16_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, (b16*)y, n / 4);
    HA2G_CHECK_LAUNCH("f32_to_b16");
    return 0;
}
int ha2g_b16_to_f32(const void* x, float* y, long n, void* stream) {
    HA2G_REQUIRE(n % 4 == 0, "b16_to_f32: n %% 4");
    if (n == 0) return 0;
    hipLaunchKernelGGL(b16_to_f32_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const b16*)x, y, n / 4);
    HA2G_CHECK_LAUNCH("b16_to_f32");
    return 0;
}
// out (bf16) = bf16(a (bf16, or NULL = 0) + b (fp32)): a tap's fp32 gradient joins the trunk's bf16 gradient stream
int ha2g_add_f32_to_b16(const void* a, const float* b, void* out, long n, void* stream) {
    HA2G_REQUIRE(n % 4 == 0, "add_f32_to_b16: n %% 4");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_f32_to_b16_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const b16*)a, b, (b16*)out, n / 4);
    HA2G_CHECK_LAUNCH("add_f32_to_b16");
    return 0;
}

}  // extern "C"
