// Stem convolution of the audio encoder: Conv2d(1 -> 32, 3x3, pad 1) + bias + ReLU on the (B,128,70)
// log-mel input (reference scripts/model/ResNetSE34V2.py:27,127-128).  One input channel means K = 9:
// not GEMM-shaped, so it is a direct HBM-bound kernel (reads 4 B, writes 128 B per pixel).
// Also the layout permutation of conv weights for the data-gradient GEMM.
#include "common.h"

namespace {

constexpr int SC = 32;   // stem output channels

// x [N][H][W], w [SC][3][3], y [N][H][W][SC] = relu(conv + bias)
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, int N, int H, int W) {
    __shared__ float ws[SC * 9 + SC];
    for (int i = threadIdx.x; i < SC * 9; i += 256) ws[i] = w[i];
    if (threadIdx.x < SC) ws[SC * 9 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const long total = (long)N * H * W * (SC / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % (SC / 4)); long pix = i / (SC / 4);
        const int ox = (int)(pix % W); long t = pix / W; const int oy = (int)(t % H); const long n = t / H;
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int iy = oy - 1 + kh, ix = ox - 1 + kw;
                xv[kh * 3 + kw] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(n * H + iy) * W + ix] : 0.f;
            }
        float4 r;
        float* rp = &r.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = c4 * 4 + k;
            float s = ws[SC * 9 + co];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) s += xv[tp] * ws[co * 9 + tp];
            rp[k] = fmaxf(s, 0.f);
        }
        st4(y, i, r);
    }
}

// partial[blk][SC][10]: taps 0..8 = dW, 9 = dbias; dy [N*H*W][SC] is the gradient w.r.t. the pre-ReLU conv output
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_partial_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                                 float* __restrict__ part, int N, int H, int W) {
    __shared__ float red[8][SC][10];
    const int co = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const long npix = (long)N * H * W;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long pbeg = (long)blockIdx.x * per, pend = min(npix, pbeg + per);
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    // pixel coordinates advance incrementally (the 64-bit divisions of pix were most of this kernel's instructions); unrolled so that the
    // next pixels' dy / x loads are issued before this pixel's FMAs retire.  The accumulation order over pixels is unchanged.
    int ox, oy; long n;
    { const long p0 = pbeg + pl; ox = (int)(p0 % W); const long t = p0 / W; oy = (int)(t % H); n = t / H; }
#pragma unroll 4
    for (long pix = pbeg + pl; pix < pend; pix += 8) {
        const float d = ld1(dy, pix * SC + co);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int iy = oy - 1 + kh, ix = ox - 1 + kw;
                float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(n * H + iy) * W + ix] : 0.f;
                acc[kh * 3 + kw] += d * xv;
            }
        acc[9] += d;
        ox += 8;
        while (ox >= W) { ox -= W; if (++oy == H) { oy = 0; ++n; } }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) red[pl][co][k] = acc[k];
    __syncthreads();
    for (int e = threadIdx.x; e < SC * 10; e += 256) {
        int c = e / 10, k = e % 10;
        float s = 0.f;
        for (int q = 0; q < 8; ++q) s += red[q][c][k];
        part[(long)e * gridDim.x + blockIdx.x] = s;       // [SC * 10][blocks]: the final kernel's waves read an element's partials contiguously
    }
}
// one wave per element: lanes stride the block partials (four loads in flight), double sums, a fixed xor tree (deterministic).  Round 6: the partial pass
// ran 908 blocks of 2 048 pixels -- 3.5 waves per SIMD walking 64 dependent batches of loads: 197 us for 147 MB on the main queue at the end of the tower's
// backward; 4 096 blocks of ~450 pixels stream it.
__global__ __launch_bounds__(256) void stem_wgrad_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw, float* __restrict__ db,
                                                               float beta) {
    const int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= SC * 10) return;
    const float* p = part + (long)e * nblk;
    double s = 0.0;
    int b = lane;
    for (; b + 192 < nblk; b += 256) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p[b + 64 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (double)v[j];
    }
    for (; b < nblk; b += 64) s += (double)p[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane != 0) return;
    int c = e / 10, k = e % 10;
    if (k < 9) dw[c * 9 + k] = (beta != 0.f ? beta * dw[c * 9 + k] : 0.f) + (float)s;
    else db[c] = (beta != 0.f ? beta * db[c] : 0.f) + (float)s;
}

// w [Cout][KH*KW][Cin] -> wt [Cin][KH*KW][Cout]
__global__ void ohwi_to_ihwo_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int KK, int Cin) {
    long total = (long)Cout * KK * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int co = (int)(i % Cout); long t = i / Cout; int kk = (int)(t % KK); int ci = (int)(t / KK);
        wt[i] = w[((long)co * KK + kk) * Cin + ci];
    }
}

}  // namespace

extern "C" {

int ha2g_stem_conv_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, void* stream) {
    long total = (long)N * H * W * (SC / 4);
    int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, N, H, W);
    HA2G_CHECK_LAUNCH("stem_conv_fwd");
    return 0;
}
// ws >= 4096*320 floats
int ha2g_stem_conv_wgrad_f32(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, float beta, float* ws,
                             void* stream) {
    hipStream_t st = (hipStream_t)stream;
    long npix = (long)N * H * W;
    int nb = (int)(npix / 448 < 1 ? 1 : (npix / 448 > 4096 ? 4096 : npix / 448));
    hipLaunchKernelGGL(stem_wgrad_partial_kernel<float>, dim3(nb), dim3(256), 0, st, x, dy, ws, N, H, W);
    hipLaunchKernelGGL(stem_wgrad_final_kernel, dim3(SC * 10 / 4), dim3(256), 0, st, ws, nb, dw, db, beta);
    HA2G_CHECK_LAUNCH("stem_conv_wgrad");
    return 0;
}
// bf16-storage mode: the stem's output / its gradient as bf16 tensors (the spectrogram and the 32x1x3x3 weight stay fp32)
int ha2g_stem_conv_fwd_b16(const float* x, const float* w, const float* bias, void* y, int N, int H, int W, void* stream) {
    long total = (long)N * H * W * (SC / 4);
    int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(stem_fwd_kernel<b16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, (b16*)y, N, H, W);
    HA2G_CHECK_LAUNCH("stem_conv_fwd_b16");
    return 0;
}
int ha2g_stem_conv_wgrad_b16(const float* x, const void* dy, float* dw, float* db, int N, int H, int W, float beta, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    long npix = (long)N * H * W;
    int nb = (int)(npix / 448 < 1 ? 1 : (npix / 448 > 4096 ? 4096 : npix / 448));
    hipLaunchKernelGGL(stem_wgrad_partial_kernel<b16>, dim3(nb), dim3(256), 0, st, x, (const b16*)dy, ws, N, H, W);
    hipLaunchKernelGGL(stem_wgrad_final_kernel, dim3(SC * 10 / 4), dim3(256), 0, st, ws, nb, dw, db, beta);
    HA2G_CHECK_LAUNCH("stem_conv_wgrad_b16");
    return 0;
}
int ha2g_conv2d_weight_ohwi_to_ihwo_f32(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    long total = (long)Cout * KH * KW * Cin;
    int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(ohwi_to_ihwo_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, KH * KW, Cin);
    HA2G_CHECK_LAUNCH("ohwi_to_ihwo");
    return 0;
}

}  // extern "C"
