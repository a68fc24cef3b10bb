// Error channel of the C-ABI + small reductions shared by several ops.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" const char* ha2g_last_error(void) { return g_err; }

int ha2g_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int ha2g_abi_version(void) { return 1; }

namespace {

// out[c] = beta*out[c] + sum_r X[r*ld + c]; one block per 64 columns, 4 waves stride the rows, fixed-order
// LDS combine (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, long rows, int cols,
                                                     float* __restrict__ out, float beta) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // HBM-bound: double accumulation is free and keeps bias grads exact
    if (c < cols) {
        long r = w;
        for (; r + 12 < rows; r += 16) {
            s0 += X[r * ld + c]; s1 += X[(r + 4) * ld + c]; s2 += X[(r + 8) * ld + c]; s3 += X[(r + 12) * ld + c];
        }
        for (; r < rows; r += 4) s0 += X[r * ld + c];
    }
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < cols) {
        double s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        out[c] = (beta != 0.f ? beta * out[c] : 0.f) + (float)s;
    }
}

}  // namespace

extern "C" int ha2g_colsum_f32(const float* X, long ld, long rows, int cols, float* out, float beta, void* stream) {
    if (cols <= 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, (hipStream_t)stream, X, ld, rows, cols, out, beta);
    HA2G_CHECK_LAUNCH("colsum");
    return 0;
}
