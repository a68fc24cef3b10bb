// How many cycles does ONE wave need per v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 when NCH independent accumulator chains are interleaved?
// (the plane kernel's and the 32-channel kernel's matrix phases measure 23-31 cycles per 16x16x32 MFMA; the pipe needs 16).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_chain_probe.hip -o mfma_chain_probe ; one wave per SIMD (grid = CUs, 256 threads) and one wave per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int NCH, int ITER> __global__ void k16(float* out, unsigned long long* cyc, int seed) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i + seed) & 7); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f32x4_t acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NCH, int ITER> __global__ void k32(float* out, unsigned long long* cyc, int seed) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x + i + seed) & 7); b[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3); }
    f32x16_t acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename F> static void run(const char* name, F launch, int nch, int iter, float* out, unsigned long long* cyc) {
    for (int threads : {64, 256, 512}) {
        launch(threads); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(threads); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-12s chains %d  waves/CU %d: %6.1f shader-clock ticks per MFMA per wave (s_memtime; 100 MHz ticks if constant clock), %7.1f us, %5.1f ns per MFMA per wave\n", name, nch, threads / 64,
               (double)c / ((double)iter * nch), ms * 1e3, ms * 1e6 / ((double)iter * nch));
    }
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
#define RUN16(N) run("16x16x32", [&](int th) { hipLaunchKernelGGL((k16<N, 4096>), dim3(256), dim3(th), 0, 0, out, cyc, 1); }, N, 4096, out, cyc)
#define RUN32(N) run("32x32x16", [&](int th) { hipLaunchKernelGGL((k32<N, 4096>), dim3(256), dim3(th), 0, 0, out, cyc, 1); }, N, 4096, out, cyc)
    RUN16(1); RUN16(2); RUN16(4); RUN16(8);
    RUN32(1); RUN32(2); RUN32(4);
    return 0;
}
