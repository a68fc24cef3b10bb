// Fused Adam over a flat parameter buffer (torch.optim.Adam semantics, no weight decay / amsgrad;
// reference scripts/train.py:155-170: lr 5e-4, betas (0.5, 0.999), eps 1e-8).  Pure HBM stream:
// 16 B read x4 + 16 B write x3 per 4 parameters.  The step counter lives in device memory so that a
// captured hipGraph advances the bias correction on every replay.
#include "common.h"

namespace {

// The scalar prologue runs in double, as torch.optim.Adam's Python scalars do (1 - beta, bias corrections, lr / bc1): computed
// in fp32, 1.f - 0.999f is off by 1.3e-5 relative and that error would sit in every second moment.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                            double lr_d, double b1_d, double b2_d, double eps_d, const int* __restrict__ step, const int* __restrict__ guard) {
    if (guard != nullptr && *guard != 0) return;      // a flagged step (cluster-GRU hand-off time-out) must not touch parameters or moments
    const double t = (double)*step;
    const double bc1 = 1.0 - pow(b1_d, t), bc2 = 1.0 - pow(b2_d, t);
    const float step_size = (float)(lr_d / bc1), rs2 = (float)(1.0 / sqrt(bc2));
    const float b1 = (float)b1_d, b2 = (float)b2_d, omb1 = (float)(1.0 - b1_d), omb2 = (float)(1.0 - b2_d), eps = (float)eps_d;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            M[k] = b1 * M[k] + omb1 * G[k];
            V[k] = b2 * V[k] + omb2 * G[k] * G[k];
            P[k] -= step_size * M[k] / (sqrtf(V[k]) * rs2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float mm = b1 * m[i] + omb1 * g[i], vv = b2 * v[i] + omb2 * g[i] * g[i];
        m[i] = mm; v[i] = vv;
        p[i] -= step_size * mm / (sqrtf(vv) * rs2 + eps);
    }
}
__global__ void step_inc_kernel(int* step, const int* guard) { if (guard == nullptr || *guard == 0) *step += 1; }

}  // namespace

extern "C" {
// guard: optional device int32 word (the cluster-GRU error word, gru_cluster.hip); while it is non-zero the counter, the parameters and the
// moments stay untouched, so a step whose gradients are invalid is a no-op on the optimizer state ("never trains on garbage")
int ha2g_adam_step_inc_guarded(int* step, const int* guard, void* stream) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, guard);
    HA2G_CHECK_LAUNCH("adam_step_inc");
    return 0;
}
int ha2g_adam_step_inc(int* step, void* stream) { return ha2g_adam_step_inc_guarded(step, nullptr, stream); }
// p, g, m, v: 16-byte aligned flat buffers of n floats; step: device int32 holding the (already incremented) step number
int ha2g_adam_guarded_f32(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps, const int* step,
                          const int* guard, void* stream) {
    if (n == 0) return 0;
    long gsz = (n / 4 + 255) / 256;
    int grid = (int)(gsz < 1 ? 1 : (gsz > 8192 ? 8192 : gsz));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, step, guard);
    HA2G_CHECK_LAUNCH("adam");
    return 0;
}
int ha2g_adam_f32(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps, const int* step,
                  void* stream) {
    return ha2g_adam_guarded_f32(p, g, m, v, n, lr, b1, b2, eps, step, nullptr, stream);
}
}
