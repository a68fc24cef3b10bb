// Shared device/host helpers for the HA2G gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HA2G_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error reporting (C-ABI: functions return 0 or a negative code; text via ha2g_last_error) ----
extern "C" const char* ha2g_last_error(void);
int ha2g_set_error(int code, const char* fmt, ...);

#define HA2G_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) return ha2g_set_error(-2, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define HA2G_REQUIRE(cond, ...)                                    \
    do {                                                           \
        if (!(cond)) return ha2g_set_error(-1, __VA_ARGS__);       \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers ----
// Gate non-linearities: the accurate libm forms (gate math is <1% of the recurrent kernels' time).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blocks of up to 1024 threads; `red` is >= 16 floats of LDS. All threads get the sum.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
