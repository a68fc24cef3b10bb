// Train-mode BatchNorm (2-D and 1-D share one [rows][C] channels-last form) and the squeeze-excite
// pointwise pieces of the SE-ResNet audio encoder.  HBM-bound kernels: float4 lanes over channels,
// per-block partial column sums + a tiny fixed-order final reduction in double (deterministic).
//
// BatchNorm semantics = torch (reference scripts/model/ResNetBlocks.py:24-30, ResNetSE34V2.py:127-129):
// normalise with the biased batch variance, update running_var with the unbiased one, momentum 0.1.
#include "common.h"

namespace {

constexpr int NB_MAX = 1024;   // max row-chunk blocks of a partial reduction

// thread layout shared by the column reductions: C4 = C/4 float4 lanes per row, 256 % C4 == 0
struct ColMap {
    int c4, r0, rstep;
    __device__ ColMap(int C4) { c4 = threadIdx.x % C4; r0 = threadIdx.x / C4; rstep = 256 / C4; }
};

struct d4 { double x, y, z, w; };
__device__ __forceinline__ d4 d4zero() { d4 r; r.x = r.y = r.z = r.w = 0.0; return r; }

// combine the per-thread double partials of threads sharing c4, write [blk][which][C] (double)
__device__ __forceinline__ void block_col_reduce(d4 a, d4 b, int C4, double* __restrict__ part, int C, d4* lds) {
    // lds: [2][256] d4
    lds[threadIdx.x] = a;
    lds[256 + threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < C4) {
        d4 sa = d4zero(), sb = d4zero();
        for (int t = threadIdx.x; t < 256; t += C4) {
            d4 x = lds[t], y = lds[256 + t];
            sa.x += x.x; sa.y += x.y; sa.z += x.z; sa.w += x.w;
            sb.x += y.x; sb.y += y.y; sb.z += y.z; sb.w += y.w;
        }
        double* p0 = part + ((long)blockIdx.x * 2 + 0) * C + threadIdx.x * 4;
        double* p1 = part + ((long)blockIdx.x * 2 + 1) * C + threadIdx.x * 4;
        p0[0] = sa.x; p0[1] = sa.y; p0[2] = sa.z; p0[3] = sa.w;
        p1[0] = sb.x; p1[1] = sb.y; p1[2] = sb.z; p1[3] = sb.w;
    }
}

// mode 0: (x - K, (x-K)^2) with K = row 0 (shift against cancellation); mode 1: (dy, dy * xhat).
// Sums are carried in double (torch's CPU batch-norm accumulates in double too; the kernel is HBM-bound, the
// fp64 adds are free) -- nearly-dead post-ReLU channels make sum(dy*xhat) cancel by 1e3..1e4.
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          long rows, int C, double* __restrict__ part) {
    __shared__ d4 lds[512];
    const int C4 = C >> 2;
    ColMap m(C4);
    const long rows_per = (rows + gridDim.x - 1) / gridDim.x;
    const long rbeg = (long)blockIdx.x * rows_per, rend = min(rows, rbeg + rows_per);
    d4 a = d4zero(), b = d4zero();
    float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), is4 = k4;
    if (MODE == 0) k4 = reinterpret_cast<const float4*>(x)[m.c4];
    else { k4 = reinterpret_cast<const float4*>(mean)[m.c4]; is4 = reinterpret_cast<const float4*>(invstd)[m.c4]; }
    // four rows per trip: the loads are issued together (one row per trip is load -> wait -> add, a single 16-byte request in flight per
    // wave: ~2.7 TB/s measured), the adds keep the row order, so the sums are bit-identical to the one-row loop's
    auto accum = [&](const float4& v, const float4& d) {
        if (MODE == 0) {
            double vx = (double)v.x - k4.x, vy = (double)v.y - k4.y, vz = (double)v.z - k4.z, vw = (double)v.w - k4.w;
            a.x += vx; a.y += vy; a.z += vz; a.w += vw;
            b.x += vx * vx; b.y += vy * vy; b.z += vz * vz; b.w += vw * vw;
        } else {
            a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
            b.x += (double)d.x * (((double)v.x - k4.x) * is4.x); b.y += (double)d.y * (((double)v.y - k4.y) * is4.y);
            b.z += (double)d.z * (((double)v.z - k4.z) * is4.z); b.w += (double)d.w * (((double)v.w - k4.w) * is4.w);
        }
    };
    long r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {
        float4 v[4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = reinterpret_cast<const float4*>(x + (r + (long)j * m.rstep) * C)[m.c4];
            d[j] = v[j];
            if (MODE == 1) d[j] = reinterpret_cast<const float4*>(dy + (r + (long)j * m.rstep) * C)[m.c4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) accum(v[j], d[j]);
    }
    for (; r < rend; r += m.rstep) {
        float4 v = reinterpret_cast<const float4*>(x + r * C)[m.c4], d = v;
        if (MODE == 1) d = reinterpret_cast<const float4*>(dy + r * C)[m.c4];
        accum(v, d);
    }
    block_col_reduce(a, b, C4, part, C, lds);
}

// final reductions: one wave per column; lanes stride the block partials, fixed-order shuffle tree (deterministic)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int nblk, long rows, int C,
                                                             const float* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ invstd, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float momentum, float eps) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    {   // partials in fixed order, four pairs of loads per trip in flight
        int b = lane;
        for (; b + 192 < nblk; b += 256) {
            double p1[4], p2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { p1[j] = part[((long)(b + 64 * j) * 2) * C + c]; p2[j] = part[((long)(b + 64 * j) * 2 + 1) * C + c]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += p1[j]; s2 += p2[j]; }
        }
        for (; b < nblk; b += 64) { s1 += part[((long)b * 2) * C + c]; s2 += part[((long)b * 2 + 1) * C + c]; }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane != 0) return;
    double n = (double)rows, m1 = s1 / n;
    double var = s2 / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    double mu = (double)x[c] + m1;
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
            for (int j = 0; j < 4; ++j) { p1[j] = part[((long)(b + 64 * j) * 2) * C + c]; p2[j] = part[((long)(b + 64 * j) * 2 + 1) * C + c]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1 += p1[j]; s2 += p2[j]; }
        }
        for (; b < nblk; b += 64) { s1 += part[((long)b * 2) * C + c]; s2 += part[((long)b * 2 + 1) * C + c]; }
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
template <int PL>
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, long rows,
                                int C, int act, unsigned short* __restrict__ y_hi, unsigned short* __restrict__ y_lo) {
    const int C4 = C >> 2;
    const long total = rows * C4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 mu = reinterpret_cast<const float4*>(mean)[c4], is = reinterpret_cast<const float4*>(invstd)[c4];
        float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
        float4 r;
        r.x = (v.x - mu.x) * is.x * g.x + b.x; r.y = (v.y - mu.y) * is.y * g.y + b.y;
        r.z = (v.z - mu.z) * is.z * g.z + b.z; r.w = (v.w - mu.w) * is.w * g.w + b.w;
        if (act == 2) {
            r.x = r.x > 0.f ? r.x : 0.01f * r.x; r.y = r.y > 0.f ? r.y : 0.01f * r.y;
            r.z = r.z > 0.f ? r.z : 0.01f * r.z; r.w = r.w > 0.f ? r.w : 0.01f * r.w;
        }
        reinterpret_cast<float4*>(y)[i] = r;
        if (PL) {
            uint2 h, l;
            split2_bf16(r.x, r.y, h.x, l.x); split2_bf16(r.z, r.w, h.y, l.y);
            reinterpret_cast<uint2*>(y_hi)[i] = h;
            reinterpret_cast<uint2*>(y_lo)[i] = l;
        }
    }
}


// BatchNorm apply fused with the squeeze of the SE layer that follows it (ResNetBlocks.py:24-36,81-83): writes y = bn(x) and
// per-(image, channel) sums of y in the same pass (the separate hw-mean kernel re-read the whole tensor).  grid (chunks, N):
// a block owns a row chunk of ONE image; per-thread double partials, block combine through LDS, one partial per
// (image, chunk, channel); pool_final adds the chunks in ascending order (deterministic).
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int HW, int C,
                                                            double* __restrict__ part) {
    __shared__ d4 lds[256];
    const int C4 = C >> 2;
    ColMap m(C4);
    const int nchunk = gridDim.x;
    const int per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const long base = (long)blockIdx.y * HW * C;
    const float4 mu = reinterpret_cast<const float4*>(mean)[m.c4], is = reinterpret_cast<const float4*>(invstd)[m.c4];
    const float4 g = reinterpret_cast<const float4*>(gamma)[m.c4], b = reinterpret_cast<const float4*>(beta)[m.c4];
    d4 a = d4zero();
    auto row = [&](const float4& v, int r) {
        float4 o;
        o.x = (v.x - mu.x) * is.x * g.x + b.x; o.y = (v.y - mu.y) * is.y * g.y + b.y;
        o.z = (v.z - mu.z) * is.z * g.z + b.z; o.w = (v.w - mu.w) * is.w * g.w + b.w;
        reinterpret_cast<float4*>(y + base + (long)r * C)[m.c4] = o;
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    };
    int r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {           // four loads in flight per wave, rows consumed in order
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const float4*>(x + base + (long)(r + j * m.rstep) * C)[m.c4];
#pragma unroll
        for (int j = 0; j < 4; ++j) row(v[j], r + j * m.rstep);
    }
    for (; r < rend; r += m.rstep) row(reinterpret_cast<const float4*>(x + base + (long)r * C)[m.c4], r);
    lds[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C4) {
        d4 s = d4zero();
        for (int t = threadIdx.x; t < 256; t += C4) { d4 v = lds[t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * C + threadIdx.x * 4;
        p[0] = s.x; p[1] = s.y; p[2] = s.z; p[3] = s.w;
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

// dx = gamma * invstd * (dy - sum_dy/N - xhat * sum_dy_xhat/N)
// PL = 1: dx is ALSO (or, with dx == nullptr, ONLY) written as two bf16 planes  hi = bf16(dx), lo = bf16(dx - hi)  -- the operand format of the
// plane-based split-bf16 consumers (conv_planes.hip): this pass is HBM-bound, the split rides in its shadow, and the convolution data /
// weight gradients that read the tensor no longer split it once per consumer tile.
template <int PL>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat, float* __restrict__ dx,
                                    long rows, int C, int relu_mask, unsigned short* __restrict__ dx_hi, unsigned short* __restrict__ dx_lo) {
    const int C4 = C >> 2;
    const long total = rows * C4;
    const float invn = 1.f / (float)rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % C4);
        float4 d = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(x)[i];
        float4 mu = reinterpret_cast<const float4*>(mean)[c4], is = reinterpret_cast<const float4*>(invstd)[c4];
        float4 g = reinterpret_cast<const float4*>(gamma)[c4];
        float4 s1 = reinterpret_cast<const float4*>(sum_dy)[c4], s2 = reinterpret_cast<const float4*>(sum_dy_xhat)[c4];
        float4 r;
        const double dn = (double)invn;
        r.x = (float)((double)g.x * is.x * ((double)d.x - s1.x * dn - ((double)v.x - mu.x) * is.x * (s2.x * dn)));
        r.y = (float)((double)g.y * is.y * ((double)d.y - s1.y * dn - ((double)v.y - mu.y) * is.y * (s2.y * dn)));
        r.z = (float)((double)g.z * is.z * ((double)d.z - s1.z * dn - ((double)v.z - mu.z) * is.z * (s2.z * dn)));
        r.w = (float)((double)g.w * is.w * ((double)d.w - s1.w * dn - ((double)v.w - mu.w) * is.w * (s2.w * dn)));
        if (relu_mask) {          // x is a ReLU output (conv -> ReLU -> BN): chain the ReLU derivative, mask = (x > 0)
            r.x = v.x > 0.f ? r.x : 0.f; r.y = v.y > 0.f ? r.y : 0.f; r.z = v.z > 0.f ? r.z : 0.f; r.w = v.w > 0.f ? r.w : 0.f;
        }
        if (!PL || dx != nullptr) reinterpret_cast<float4*>(dx)[i] = r;
        if (PL) {
            uint2 h, l;
            split2_bf16(r.x, r.y, h.x, l.x); split2_bf16(r.z, r.w, h.y, l.y);
            reinterpret_cast<uint2*>(dx_hi)[i] = h;
            reinterpret_cast<uint2*>(dx_lo)[i] = l;
        }
    }
}

// ---- squeeze-excite pieces; x is [N][HW][C] -------------------------------------------------------------
// per-image column mean (MODE 0) or per-image sum of a*b (MODE 1: ds[n,c] = sum_hw dpre*b2 with dpre = dout*(out>0))
template <int MODE>
__global__ __launch_bounds__(256) void image_col_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                        const float* __restrict__ outp, int HW, int C, float* __restrict__ res,
                                                        float scale, double* __restrict__ part, const float* __restrict__ gate) {
    // grid (nchunk, N): block (k, n) sums rows [k*per, (k+1)*per) of image n; part == nullptr (nchunk = 1): final floats to res, else
    // double partials [n][k][C] for pool_final_kernel (more blocks than images: a 128-image batch alone fills half the CUs)
    __shared__ d4 lds[256];
    const int C4 = C >> 2;
    ColMap m(C4);
    const int nchunk = gridDim.x, per = (HW + nchunk - 1) / nchunk;
    const int rbeg = blockIdx.x * per, rend = min(HW, rbeg + per);
    const long base = (long)blockIdx.y * HW * C;
    d4 a = d4zero();
    // one block per image: four rows per trip so that 4 (MODE 0) / 12 (MODE 1) 16-byte loads are in flight per wave; the adds keep the row
    // order of the one-row loop (bit-identical sums)
    auto accum = [&](const float4& v, const float4& d, const float4& o) {
        if (MODE == 1) {
            a.x += o.x > 0.f ? (double)v.x * d.x : 0.0; a.y += o.y > 0.f ? (double)v.y * d.y : 0.0;
            a.z += o.z > 0.f ? (double)v.z * d.z : 0.0; a.w += o.w > 0.f ? (double)v.w * d.w : 0.0;
        } else {
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    };
    int r = rbeg + m.r0;
    for (; r + 3 * m.rstep < rend; r += 4 * m.rstep) {
        float4 v[4], d[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long off = base + (long)(r + j * m.rstep) * C;
            v[j] = reinterpret_cast<const float4*>(x + off)[m.c4];
            d[j] = v[j]; o[j] = v[j];
            if (MODE == 1) { d[j] = reinterpret_cast<const float4*>(dout + off)[m.c4]; o[j] = reinterpret_cast<const float4*>(outp + off)[m.c4]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) accum(v[j], d[j], o[j]);
    }
    for (; r < rend; r += m.rstep) {
        const long off = base + (long)r * C;
        float4 v = reinterpret_cast<const float4*>(x + off)[m.c4], d = v, o = v;
        if (MODE == 1) { d = reinterpret_cast<const float4*>(dout + off)[m.c4]; o = reinterpret_cast<const float4*>(outp + off)[m.c4]; }
        accum(v, d, o);
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C4) {
        d4 s = d4zero();
        for (int t = threadIdx.x; t < 256; t += C4) { d4 v = lds[t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        if (part) {
            double* p = part + ((long)blockIdx.y * nchunk + blockIdx.x) * C + threadIdx.x * 4;
            p[0] = s.x; p[1] = s.y; p[2] = s.z; p[3] = s.w;
        } else {
            float4 o = make_float4((float)(s.x * scale), (float)(s.y * scale), (float)(s.z * scale), (float)(s.w * scale));
            if (gate) {
                const float4 g = reinterpret_cast<const float4*>(gate + (long)blockIdx.y * C)[threadIdx.x];
                o.x = o.x * g.x * (1.f - g.x); o.y = o.y * g.y * (1.f - g.y); o.z = o.z * g.z * (1.f - g.z); o.w = o.w * g.w * (1.f - g.w);
            }
            reinterpret_cast<float4*>(res + (long)blockIdx.y * C)[threadIdx.x] = o;
        }
    }
}

// out = relu(x * s[n,c] + res)
template <int PL>
__global__ void se_scale_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ res,
                                         float* __restrict__ out, long N, int HW, int C, unsigned short* __restrict__ o_hi,
                                         unsigned short* __restrict__ o_lo) {
    const int C4 = C >> 2;
    const long per = (long)HW * C4, total = N * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long n = i / per; int c4 = (int)(i % C4);
        float4 v = reinterpret_cast<const float4*>(x)[i], r = reinterpret_cast<const float4*>(res)[i];
        float4 sc = reinterpret_cast<const float4*>(s + n * C)[c4];
        float4 o;
        o.x = fmaxf(v.x * sc.x + r.x, 0.f); o.y = fmaxf(v.y * sc.y + r.y, 0.f);
        o.z = fmaxf(v.z * sc.z + r.z, 0.f); o.w = fmaxf(v.w * sc.w + r.w, 0.f);
        reinterpret_cast<float4*>(out)[i] = o;
        if (PL) {
            uint2 h, l;
            split2_bf16(o.x, o.y, h.x, l.x); split2_bf16(o.z, o.w, h.y, l.y);
            reinterpret_cast<uint2*>(o_hi)[i] = h;
            reinterpret_cast<uint2*>(o_lo)[i] = l;
        }
    }
}
// dpre = dout * (out > 0); dres = dpre; dx = dpre * s[n,c] + dpool[n,c]   (dpool already divided by HW)
__global__ void se_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ outp, const float* __restrict__ s,
                                    const float* __restrict__ dpool, float* __restrict__ dres, float* __restrict__ dx, long N,
                                    int HW, int C) {
    const int C4 = C >> 2;
    const long per = (long)HW * C4, total = N * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long n = i / per; int c4 = (int)(i % C4);
        float4 d = reinterpret_cast<const float4*>(dout)[i], o = reinterpret_cast<const float4*>(outp)[i];
        float4 sc = reinterpret_cast<const float4*>(s + n * C)[c4], dp = reinterpret_cast<const float4*>(dpool + n * C)[c4];
        float4 p;
        p.x = o.x > 0.f ? d.x : 0.f; p.y = o.y > 0.f ? d.y : 0.f; p.z = o.z > 0.f ? d.z : 0.f; p.w = o.w > 0.f ? d.w : 0.f;
        reinterpret_cast<float4*>(dres)[i] = p;
        float4 q;
        q.x = p.x * sc.x + dp.x; q.y = p.y * sc.y + dp.y; q.z = p.z * sc.z + dp.z; q.w = p.w * sc.w + dp.w;
        reinterpret_cast<float4*>(dx)[i] = q;
    }
}

inline int chunk_blocks(long rows) {
    long b = rows / 512;
    if (b < 1) b = 1;
    if (b > NB_MAX) b = NB_MAX;
    return (int)b;
}
inline int flat_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
inline bool okC(int C) { return C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0; }

}  // namespace

extern "C" {

// floats of scratch needed by the column reductions below
long ha2g_bn_workspace_floats(int C) { return (long)NB_MAX * 2 * C * 2; }   /* partials are doubles */

// mean/invstd [C] out; running_mean/var updated in place when non-null.  x is [rows][C], C in {4,8,...,1024} with 256 % (C/4) == 0.
int ha2g_bn_stats_f32(const float* x, long rows, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                      float momentum, float eps, float* ws, void* stream) {
    HA2G_REQUIRE(okC(C), "bn: unsupported channel count %d", C);
    HA2G_REQUIRE(rows > 0, "bn: empty batch");
    hipStream_t st = (hipStream_t)stream;
    int nb = chunk_blocks(rows);
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3(nb), dim3(256), 0, st, x, nullptr, nullptr, nullptr, rows, C, (double*)ws);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)ws, nb, rows, C, x, mean, invstd, running_mean,
                       running_var, momentum, eps);
    HA2G_CHECK_LAUNCH("bn_stats");
    return 0;
}
int ha2g_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                      long rows, int C, int act, void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "bn: C %% 4");
    hipLaunchKernelGGL(bn_apply_kernel<0>, dim3(flat_grid(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       y, rows, C, act, (unsigned short*)nullptr, (unsigned short*)nullptr);
    HA2G_CHECK_LAUNCH("bn_apply");
    return 0;
}
// ha2g_bn_apply_f32 that also writes y as bf16 planes y_hi / y_lo [rows][C]
int ha2g_bn_apply_planes_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y, void* y_hi,
                             void* y_lo, long rows, int C, int act, void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "bn: C %% 4");
    HA2G_REQUIRE(y_hi != nullptr && y_lo != nullptr, "bn_apply_planes: null plane");
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(flat_grid(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       y, rows, C, act, (unsigned short*)y_hi, (unsigned short*)y_lo);
    HA2G_CHECK_LAUNCH("bn_apply_planes");
    return 0;
}
// y = bn(x) for x [N][HW][C] AND pooled[n][c] = mean over HW of y (the SE squeeze) in one pass over the tensor.
// ws: >= ha2g_bn_apply_pool_workspace_floats(N, HW, C) floats.
static int pool_chunks(int N, int HW) {
    int c = 2048 / (N < 1 ? 1 : N);                      // ~2k blocks per launch
    int cap = HW / 64;                                   // >= 64 rows per block
    if (c > cap) c = cap;
    return c < 1 ? 1 : c;
}
long ha2g_bn_apply_pool_workspace_floats(int N, int HW, int C) { return (long)N * pool_chunks(N, HW) * C * 2; }
int ha2g_bn_apply_pool_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                           int N, int HW, int C, float* pooled, float* ws, void* stream) {
    HA2G_REQUIRE(okC(C), "bn_apply_pool: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = pool_chunks(N, HW);
    hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(nchunk, N), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, HW, C, (double*)ws);
    hipLaunchKernelGGL(pool_final_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, (long)N * C,
                       1.f / (float)HW, pooled, nullptr);
    HA2G_CHECK_LAUNCH("bn_apply_pool");
    return 0;
}
// dgamma = sum dy*xhat, dbeta = sum dy, dx as torch's batch-norm backward (train mode)
// relu_mask = 1: x is the output of a ReLU that precedes the BatchNorm; dx then is the gradient w.r.t. the ReLU's INPUT
int ha2g_bn_bwd_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx,
                    float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta, float* ws,
                    void* stream) {
    HA2G_REQUIRE(okC(C), "bn: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    int nb = chunk_blocks(rows);
    hipLaunchKernelGGL(col_partial_kernel<1>, dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    hipLaunchKernelGGL(pair_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)ws, nb, C, dbeta, dgamma, acc_dbeta, acc_dgamma);
    if (dx)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(flat_grid(rows * (C / 4))), dim3(256), 0, st, dy, x, mean, invstd, gamma, dbeta,
                           dgamma, dx, rows, C, relu_mask, (unsigned short*)nullptr, (unsigned short*)nullptr);
    HA2G_CHECK_LAUNCH("bn_bwd");
    return 0;
}
// ha2g_bn_bwd_f32 whose dx goes out as bf16 planes dx_hi / dx_lo [rows][C] (and, when dx != NULL, in fp32 as well)
int ha2g_bn_bwd_planes_f32(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma, float* dx, void* dx_hi,
                           void* dx_lo, float* dgamma, float* dbeta, long rows, int C, int relu_mask, float* acc_dgamma, float* acc_dbeta,
                           float* ws, void* stream) {
    HA2G_REQUIRE(okC(C), "bn: unsupported channel count %d", C);
    HA2G_REQUIRE(dx_hi != nullptr && dx_lo != nullptr, "bn_bwd_planes: null plane");
    hipStream_t st = (hipStream_t)stream;
    int nb = chunk_blocks(rows);
    hipLaunchKernelGGL(col_partial_kernel<1>, dim3(nb), dim3(256), 0, st, x, dy, mean, invstd, rows, C, (double*)ws);
    hipLaunchKernelGGL(pair_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)ws, nb, C, dbeta, dgamma, acc_dbeta, acc_dgamma);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(flat_grid(rows * (C / 4))), dim3(256), 0, st, dy, x, mean, invstd, gamma, dbeta, dgamma, dx,
                       rows, C, relu_mask, (unsigned short*)dx_hi, (unsigned short*)dx_lo);
    HA2G_CHECK_LAUNCH("bn_bwd_planes");
    return 0;
}
// out[n][c] = mean over HW of x[n][hw][c]
int ha2g_hw_mean_f32(const float* x, float* out, int N, int HW, int C, void* stream) {
    HA2G_REQUIRE(okC(C), "hw_mean: unsupported channel count %d", C);
    hipLaunchKernelGGL(image_col_kernel<0>, dim3(1, N), dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr, HW, C, out, 1.f / (float)HW, nullptr, nullptr);
    HA2G_CHECK_LAUNCH("hw_mean");
    return 0;
}
int ha2g_se_scale_add_relu_f32(const float* x, const float* s, const float* res, float* out, int N, int HW, int C, void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "se: C %% 4");
    hipLaunchKernelGGL(se_scale_add_relu_kernel<0>, dim3(flat_grid((long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, s, res,
                       out, (long)N, HW, C, (unsigned short*)nullptr, (unsigned short*)nullptr);
    HA2G_CHECK_LAUNCH("se_scale_add_relu");
    return 0;
}
// the same, out also as bf16 planes (the next block's conv1 reads them in its weight gradient)
int ha2g_se_scale_add_relu_planes_f32(const float* x, const float* s, const float* res, float* out, void* o_hi, void* o_lo, int N, int HW, int C,
                                      void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "se: C %% 4");
    HA2G_REQUIRE(o_hi != nullptr && o_lo != nullptr, "se_scale_add_relu_planes: null plane");
    hipLaunchKernelGGL(se_scale_add_relu_kernel<1>, dim3(flat_grid((long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, s, res,
                       out, (long)N, HW, C, (unsigned short*)o_hi, (unsigned short*)o_lo);
    HA2G_CHECK_LAUNCH("se_scale_add_relu_planes");
    return 0;
}
// ds[n][c] = sum_hw dout*(out>0)*x
int ha2g_se_bwd_scale_f32(const float* dout, const float* out, const float* x, float* ds, int N, int HW, int C, const float* gate, float* ws,
                          void* stream) {
    HA2G_REQUIRE(okC(C), "se: unsupported channel count %d", C);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ws ? pool_chunks(N, HW) : 1;          // ws: ha2g_bn_apply_pool_workspace_floats(N, HW, C) floats, or null
    if (nchunk > 1) {
        hipLaunchKernelGGL(image_col_kernel<1>, dim3(nchunk, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)ws, (const float*)nullptr);
        hipLaunchKernelGGL(pool_final_kernel, dim3(ceil_div((long)N * C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, (long)N * C, 1.f, ds,
                           gate);
    } else {
        hipLaunchKernelGGL(image_col_kernel<1>, dim3(1, N), dim3(256), 0, st, x, dout, out, HW, C, ds, 1.f, (double*)nullptr, gate);
    }
    HA2G_CHECK_LAUNCH("se_bwd_scale");
    return 0;
}
int ha2g_se_bwd_apply_f32(const float* dout, const float* out, const float* s, const float* dpool, float* dres, float* dx, int N,
                          int HW, int C, void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "se: C %% 4");
    hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(flat_grid((long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, dout, out, s,
                       dpool, dres, dx, (long)N, HW, C);
    HA2G_CHECK_LAUNCH("se_bwd_apply");
    return 0;
}

}  // extern "C"
