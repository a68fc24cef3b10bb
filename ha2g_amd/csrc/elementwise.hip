// Memory-bound glue kernels of the HA2G step: embedding gather / deterministic scatter-add, 1-D im2col /
// col2im for the TCN and discriminator convolutions, weight-norm, pointwise ops, Philox dropout, pixel
// shuffle, speaker-softmax blending.  All are HBM-bound: float4 accesses, grid-stride loops, no atomics
// on floats (reductions are fixed-order => bitwise reproducible).
#include "common.h"

namespace {

constexpr int EB = 256;
inline int grid_for(long n, int per = 1) { long g = (n + (long)EB * per - 1) / ((long)EB * per); return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// ---------------------------------------------------------------- embedding ---------------------------
__global__ void embedding_fwd_kernel(const long* __restrict__ tok, const float* __restrict__ W, float* __restrict__ out,
                                     long n, int C4) {
    long total = n * C4;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        long r = i / C4; int c = (int)(i % C4);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(W + tok[r] * (long)C4 * 4)[c];
    }
}

// dW[tok] += sum over positions carrying tok, deterministic and atomic-free:
//  * the padding id (token `heavy`, ~80 % of all positions: in_text_padded is zero except at word onsets, reference
//    data_loader/lmdb_data_loader.py:116-141) is a masked column sum done in two fixed-order levels;
//  * every other token: the block of its FIRST occurrence sums all its occurrences in ascending position order
//    (4 waves take matches round-robin, fixed-order LDS combine).
__global__ __launch_bounds__(256) void embedding_bwd_heavy_partial(const long* __restrict__ tok, const float* __restrict__ dY,
                                                                   double* __restrict__ part, int n, int C, long heavy) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int pbeg = blockIdx.y * per, pend = min(n, pbeg + per);
    double s = 0.0;
    for (int p = pbeg + w; p < pend; p += 4)
        if (tok[p] == heavy && c < C) s += dY[(long)p * C + c];
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) part[(long)blockIdx.y * C + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
__global__ void embedding_bwd_heavy_final(const double* __restrict__ part, int nchunk, int C, float* __restrict__ dW, long heavy) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nchunk; ++k) s += part[(long)k * C + c];
    dW[heavy * C + c] += (float)s;
}
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long* __restrict__ tok, const float* __restrict__ dY,
                                                            float* __restrict__ dW, int n, int C, long heavy) {
    __shared__ float part[4][64];
    __shared__ int first;
    const int p = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long t = tok[p];
    if (t == heavy) return;
    if (threadIdx.x == 0) first = 1;
    __syncthreads();
    for (int q = threadIdx.x; q < p; q += 256)
        if (tok[q] == t) first = 0;
    __syncthreads();
    if (!first) return;
    const int c = blockIdx.y * 64 + lane;
    float s = 0.f;
    int k = 0;                                            // running index among matches
    for (int q0 = p; q0 < n; q0 += 64) {
        int q = q0 + lane;
        unsigned long long m = __ballot(q < n && tok[q] == t);
        while (m) {
            int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if ((k & 3) == w && c < C) s += dY[(long)(q0 + b) * C + c];
            ++k;
        }
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) dW[t * C + c] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ---------------------------------------------------------------- 1-D im2col --------------------------
// x [B][T][C] -> col [B][To][C*k], col[b][t][c*k + kk] = x[b][t - pad_left + kk*dil][c] (0 outside).
__global__ void im2col1d_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int T, int C, int k, int dil,
                                int pad_left, int To) {
    long total = (long)B * To * C * k;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        int kk = (int)(i % k); long r = i / k; int c = (int)(r % C); r /= C; int t = (int)(r % To); int b = (int)(r / To);
        int ts = t - pad_left + kk * dil;
        col[i] = (ts >= 0 && ts < T) ? x[((long)b * T + ts) * C + c] : 0.f;
    }
}
// k = 2, C % 4 == 0 (the TCN's convolutions, model/tcn.py:16-45): a thread moves four channels of one output row -- two 16-byte loads (tap 0 = row
// t - pad_left, tap 1 = row t - pad_left + dil), two 16-byte stores of the interleaved eight columns (c k + kk); no per-element 64-bit division
// (the generic kernel above: 38 us per 94 MB, 2.4 TB/s)
__global__ void im2col1d_k2_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int T, int C4, int dil, int pad_left, int To) {
    const long total = (long)B * To * C4;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int c4 = (int)(i % C4); const long r = i / C4; const int t = (int)(r % To); const long b = r / To;
        const int t0 = t - pad_left, t1 = t0 + dil;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), v = a;
        if (t0 >= 0 && t0 < T) a = reinterpret_cast<const float4*>(x)[(b * T + t0) * C4 + c4];
        if (t1 >= 0 && t1 < T) v = reinterpret_cast<const float4*>(x)[(b * T + t1) * C4 + c4];
        float4* d = reinterpret_cast<float4*>(col) + (r * C4 + c4) * 2;
        d[0] = make_float4(a.x, v.x, a.y, v.y);
        d[1] = make_float4(a.z, v.z, a.w, v.w);
    }
}
// ... and its transpose: dx[b][ts][c] = dcol[b][ts + pad_left][2 c] + dcol[b][ts + pad_left - dil][2 c + 1] (the generic kernel's order of the two terms)
__global__ void col2im1d_k2_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int T, int C4, int dil, int pad_left, int To) {
    const long total = (long)B * T * C4;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int c4 = (int)(i % C4); const long r = i / C4; const int ts = (int)(r % T); const long b = r / T;
        const int t0 = ts + pad_left, t1 = t0 - dil;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 >= 0 && t0 < To) {
            const float4* p = reinterpret_cast<const float4*>(dcol) + ((b * To + t0) * C4 + c4) * 2;
            const float4 u = p[0], w = p[1];
            s = make_float4(u.x, u.z, w.x, w.z);
        }
        if (t1 >= 0 && t1 < To) {
            const float4* p = reinterpret_cast<const float4*>(dcol) + ((b * To + t1) * C4 + c4) * 2;
            const float4 u = p[0], w = p[1];
            s.x += u.y; s.y += u.w; s.z += w.y; s.w += w.w;
        }
        reinterpret_cast<float4*>(dx)[i] = s;
    }
}
// dx[b][ts][c] = sum_kk dcol[b][ts + pad_left - kk*dil][c*k + kk]
__global__ void col2im1d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int T, int C, int k, int dil,
                                int pad_left, int To) {
    long total = (long)B * T * C;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        int c = (int)(i % C); long r = i / C; int ts = (int)(r % T); int b = (int)(r / T);
        float s = 0.f;
        for (int kk = 0; kk < k; ++kk) {
            int t = ts + pad_left - kk * dil;
            if (t >= 0 && t < To) s += dcol[(((long)b * To + t) * C + c) * k + kk];
        }
        dx[i] = s;
    }
}

// ---------------------------------------------------------------- weight norm --------------------------
// w[o][:] = g[o] * v[o][:] / ||v[o]||   (one block per output channel; n = in*k elements per row)
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                              float* __restrict__ w, float* __restrict__ norm, int n) {
    __shared__ float red[16];
    const int o = blockIdx.x;
    const float* vr = v + (long)o * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += vr[i] * vr[i];
    s = block_sum(s, red);
    const float nr = sqrtf(s);
    if (threadIdx.x == 0) norm[o] = nr;
    const float sc = g[o] / nr;
    for (int i = threadIdx.x; i < n; i += 256) w[(long)o * n + i] = vr[i] * sc;
}
// dg[o] = <dw, v>/||v||;  dv = g/||v|| * (dw - v <dw,v>/||v||^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ g,
                                                              const float* __restrict__ v, const float* __restrict__ norm,
                                                              float* __restrict__ dg, float* __restrict__ dv, int n, float beta) {
    __shared__ float red[16];
    const int o = blockIdx.x;
    const float* vr = v + (long)o * n;
    const float* dr = dw + (long)o * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += dr[i] * vr[i];
    s = block_sum(s, red);
    const float nr = norm[o];
    if (threadIdx.x == 0) dg[o] = (beta != 0.f ? beta * dg[o] : 0.f) + s / nr;
    const float a = g[o] / nr, bq = s / (nr * nr);
    for (int i = threadIdx.x; i < n; i += 256) {
        float* d = dv + (long)o * n + i;
        *d = (beta != 0.f ? beta * *d : 0.f) + a * (dr[i] - vr[i] * bq);
    }
}

// Many weight-norm layers of ONE row length in one launch (grid.y = layer): the 24 (48) convolutions of the generators' text encoders are
// parameters-only work that otherwise costs one tiny launch each at the head of every step.  Same per-row arithmetic as the single kernels.
struct WnMulti { const float* g[32]; const float* v[32]; float* w[32]; float* norm[32]; const float* dw[32]; float* dg[32]; float* dv[32]; };
__global__ __launch_bounds__(256) void weight_norm_multi_fwd_kernel(WnMulti m, int n) {
    __shared__ float red[16];
    const int o = blockIdx.x, t = blockIdx.y;
    const float* vr = m.v[t] + (long)o * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += vr[i] * vr[i];
    s = block_sum(s, red);
    const float nr = sqrtf(s);
    if (threadIdx.x == 0) m.norm[t][o] = nr;
    const float sc = m.g[t][o] / nr;
    float* wr = m.w[t] + (long)o * n;
    for (int i = threadIdx.x; i < n; i += 256) wr[i] = vr[i] * sc;
}
__global__ __launch_bounds__(256) void weight_norm_multi_bwd_kernel(WnMulti m, int n, float beta) {
    __shared__ float red[16];
    const int o = blockIdx.x, t = blockIdx.y;
    const float* vr = m.v[t] + (long)o * n;
    const float* dr = m.dw[t] + (long)o * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += dr[i] * vr[i];
    s = block_sum(s, red);
    const float nr = m.norm[t][o];
    if (threadIdx.x == 0) m.dg[t][o] = (beta != 0.f ? beta * m.dg[t][o] : 0.f) + s / nr;
    const float a = m.g[t][o] / nr, bq = s / (nr * nr);
    float* dvr = m.dv[t] + (long)o * n;
    for (int i = threadIdx.x; i < n; i += 256) dvr[i] = (beta != 0.f ? beta * dvr[i] : 0.f) + a * (dr[i] - vr[i] * bq);
}

// ---------------------------------------------------------------- pointwise ---------------------------
enum EltOp {
    OP_ADD = 0, OP_MUL = 1, OP_ADD_RELU = 2, OP_RELU_BWD = 3, OP_LEAKY_BWD = 4, OP_SIGMOID_BWD = 5, OP_ELU = 6,
    OP_ELU_BWD = 7, OP_REPARAM = 8, OP_REPARAM_BWD_LOGVAR = 9, OP_AXPBY = 10, OP_LEAKY = 11, OP_RELU = 12, OP_SCALE = 13, OP_MUL_SCALAR = 14, OP_SIGMOID = 15, OP_SIGMOID_BWD_PRE = 16, OP_RSQRT_EPS = 17, OP_LEAKY_A = 18,
};

__device__ __forceinline__ float elt(int op, float a, float b, float c, float alpha, float beta) {
    switch (op) {
        case OP_ADD: return a + b;
        case OP_MUL: return a * b;
        case OP_ADD_RELU: return fmaxf(a + b, 0.f);
        case OP_RELU_BWD: return b > 0.f ? a : 0.f;                       // a = dy, b = y
        case OP_LEAKY_BWD: return b > 0.f ? a : 0.01f * a;
        case OP_SIGMOID_BWD: return a * b * (1.f - b);                     // b = sigmoid output
        case OP_ELU: return a > 0.f ? a : expm1f(a);
        case OP_ELU_BWD: return b > 0.f ? a : a * (b + 1.f);              // b = elu output
        case OP_REPARAM: return a + c * expf(0.5f * b);                    // a = mu, b = logvar, c = eps
        case OP_REPARAM_BWD_LOGVAR: return a * c * 0.5f * expf(0.5f * b);  // a = dz
        case OP_AXPBY: return alpha * a + beta * b;
        case OP_LEAKY: return a > 0.f ? a : 0.01f * a;
        case OP_RELU: return fmaxf(a, 0.f);
        case OP_SCALE: return alpha * a;
        case OP_SIGMOID: return 1.f / (1.f + expf(-a));
        // a = dy, b = pre-activation: s(1-s) as sigma(b)*sigma(-b) -- no cancellation when the gate saturates
        case OP_RSQRT_EPS: return 1.f / sqrtf(a + alpha);           // inference BatchNorm: invstd from running_var
        case OP_LEAKY_A: return a > 0.f ? a : alpha * a;             // LeakyReLU(alpha): 0.2 in the FGD auto-encoder
        case OP_SIGMOID_BWD_PRE: return a * (1.f / (1.f + expf(-b))) * (1.f / (1.f + expf(b)));
    }
    return 0.f;
}

__global__ void eltwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                               float* __restrict__ out, long n, float alpha, float beta, int vec) {
    const long n4 = vec ? (n >> 2) : 0;
    if (op == OP_MUL_SCALAR) {          // out = a * b[0], b = device scalar (upstream gradient of a loss term)
        const float sc = b[0] * alpha;
        for (long i = (long)blockIdx.x * EB + threadIdx.x; i < n; i += (long)gridDim.x * EB) out[i] = a[i] * sc;
        return;
    }
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < n4; i += (long)gridDim.x * EB) {
        float4 va = reinterpret_cast<const float4*>(a)[i];
        float4 vb = b ? reinterpret_cast<const float4*>(b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 vc = c ? reinterpret_cast<const float4*>(c)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r;
        r.x = elt(op, va.x, vb.x, vc.x, alpha, beta); r.y = elt(op, va.y, vb.y, vc.y, alpha, beta);
        r.z = elt(op, va.z, vb.z, vc.z, alpha, beta); r.w = elt(op, va.w, vb.w, vc.w, alpha, beta);
        reinterpret_cast<float4*>(out)[i] = r;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * EB + threadIdx.x; i < n; i += (long)gridDim.x * EB)
        out[i] = elt(op, a[i], b ? b[i] : 0.f, c ? c[i] : 0.f, alpha, beta);
}

// ---------------------------------------------------------------- bidirectional sum --------------------
// out[r][h] = y[r][h] + y[r][H+h]  (model/hierarchy_net.py:145);  inverse: dy[r][h] -> dout[r][h], dout[r][H+h]
__global__ void dirsum_kernel(const float* __restrict__ y, float* __restrict__ out, long rows, int H4, int inverse) {
    const long total = rows * H4;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        long r = i / H4; int h = (int)(i % H4);
        if (!inverse) {
            float4 a = reinterpret_cast<const float4*>(y)[r * 2 * H4 + h], b = reinterpret_cast<const float4*>(y)[r * 2 * H4 + H4 + h];
            reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        } else {
            float4 d = reinterpret_cast<const float4*>(y)[i];
            reinterpret_cast<float4*>(out)[r * 2 * H4 + h] = d;
            reinterpret_cast<float4*>(out)[r * 2 * H4 + H4 + h] = d;
        }
    }
}

// ---------------------------------------------------------------- dropout (Philox4x32-10) -------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t h0 = __umulhi(M0, c0), l0 = M0 * c0, h1 = __umulhi(M1, c2), l1 = M1 * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// mask[i] = keep ? 1/(1-p) : 0 ; out = x * mask.  state = {seed, step} lives in device memory so a captured
// hipGraph draws fresh masks on every replay; `stream_id` separates call sites within a step.
// mode (round 6, the TCN blocks' dropouts fused with their neighbours; same mask for the same (state, stream_id, element)):
//   0: out = x * mask                          1: out = relu(x * mask + b)   (dropout + residual add + ReLU, model/tcn.py:44-46 after :29)
//   2: out = b > 0 ? x * mask : 0              (backward: the dropout's backward and the ReLU' of the convolution in front of it, b = its output)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ mask, long n, float p,
                               const unsigned long long* __restrict__ state, unsigned stream_id, int vec, const float* __restrict__ b = nullptr,
                               int mode = 0, long off4 = 0) {
    const unsigned long long seed = state[0], step = state[1];
    const float scale = 1.f / (1.f - p);
    const long n4 = (n + 3) >> 2;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < n4; i += (long)gridDim.x * EB) {
        const long ic = i + off4;                          // off4: this launch covers elements [4 off4, 4 off4 + n) of the tensor the mask is defined on (a row slice)
        uint32_t c0 = (uint32_t)ic, c1 = (uint32_t)(ic >> 32) ^ stream_id, c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        uint32_t rr[4] = {c0, c1, c2, c3};
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = ((rr[j] >> 8) * (1.0f / 16777216.0f)) >= p ? scale : 0.f;
        if (vec && i * 4 + 3 < n) {                        // 16-byte accesses when the buffers allow it (the common case)
            if (mask) reinterpret_cast<float4*>(mask)[i] = make_float4(m[0], m[1], m[2], m[3]);
            if (out) {
                const float4 v = reinterpret_cast<const float4*>(x)[i];
                float4 o = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
                if (mode != 0) {
                    const float4 w = reinterpret_cast<const float4*>(b)[i];
                    if (mode == 1) o = make_float4(fmaxf(o.x + w.x, 0.f), fmaxf(o.y + w.y, 0.f), fmaxf(o.z + w.z, 0.f), fmaxf(o.w + w.w, 0.f));
                    else o = make_float4(w.x > 0.f ? o.x : 0.f, w.y > 0.f ? o.y : 0.f, w.z > 0.f ? o.z : 0.f, w.w > 0.f ? o.w : 0.f);
                }
                reinterpret_cast<float4*>(out)[i] = o;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long e = i * 4 + j;
                if (e < n) {
                    if (mask) mask[e] = m[j];
                    if (out) {
                        float o = x[e] * m[j];
                        if (mode == 1) o = fmaxf(o + b[e], 0.f);
                        else if (mode == 2) o = b[e] > 0.f ? o : 0.f;
                        out[e] = o;
                    }
                }
            }
        }
    }
}
// dropout_kernel's keep factors of elements [4 ic, 4 ic + 4) (the same mask for the same (state, stream_id, element))
__device__ __forceinline__ void philox_mask4(long ic, unsigned long long seed, unsigned long long step, unsigned stream_id, float p, float scale, float* m) {
    uint32_t c0 = (uint32_t)ic, c1 = (uint32_t)(ic >> 32) ^ stream_id, c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint32_t rr[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = ((rr[j] >> 8) * (1.0f / 16777216.0f)) >= p ? scale : 0.f;
}
// im2col1d_k2_kernel of dropout(x) without the dropped tensor (round 6; model/tcn.py:21-31: conv1 -> ReLU -> dropout -> conv2): the mask of ha2g_dropout_f32
// for the same (state, stream_id) over x's elements is re-drawn for both taps of an output row (a launch and 2 x |x| bytes less per convolution)
__global__ void im2col1d_k2_drop_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int T, int C4, int dil, int pad_left, int To, float p,
                                        const unsigned long long* __restrict__ state, unsigned stream_id) {
    const unsigned long long seed = state[0], step = state[1];
    const float scale = 1.f / (1.f - p);
    const long total = (long)B * To * C4;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int c4 = (int)(i % C4); const long r = i / C4; const int t = (int)(r % To); const long b = r / To;
        const int t0 = t - pad_left, t1 = t0 + dil;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), v = a;
        if (t0 >= 0 && t0 < T) {
            const long ic = (b * T + t0) * C4 + c4;
            float m[4];
            philox_mask4(ic, seed, step, stream_id, p, scale, m);
            a = reinterpret_cast<const float4*>(x)[ic];
            a = make_float4(a.x * m[0], a.y * m[1], a.z * m[2], a.w * m[3]);
        }
        if (t1 >= 0 && t1 < T) {
            const long ic = (b * T + t1) * C4 + c4;
            float m[4];
            philox_mask4(ic, seed, step, stream_id, p, scale, m);
            v = reinterpret_cast<const float4*>(x)[ic];
            v = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
        }
        float4* d = reinterpret_cast<float4*>(col) + (r * C4 + c4) * 2;
        d[0] = make_float4(a.x, v.x, a.y, v.y);
        d[1] = make_float4(a.z, v.z, a.w, v.w);
    }
}
__global__ void rng_advance_kernel(unsigned long long* state) { state[1] += 1ull; }

// ---------------------------------------------------------------- pixel shuffle (NHWC) ----------------
// torch.nn.PixelShuffle(r) on logical NCHW: out[n, c, h*r+i, w*r+j] = in[n, c*r*r + i*r + j, h, w].
// Physical layout here is NHWC on both sides.
__global__ void pixel_shuffle_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int Cout, int r,
                                     int inverse) {
    long total = (long)N * H * r * W * r * Cout;
    const int Cin = Cout * r * r;
    for (long o = (long)blockIdx.x * EB + threadIdx.x; o < total; o += (long)gridDim.x * EB) {
        int c = (int)(o % Cout); long t = o / Cout; int ox = (int)(t % (W * r)); t /= (W * r); int oy = (int)(t % (H * r)); int n = (int)(t / (H * r));
        int h = oy / r, i = oy % r, w = ox / r, j = ox % r;
        long src = (((long)n * H + h) * W + w) * Cin + c * r * r + i * r + j;
        if (!inverse) out[o] = in[src]; else out[src] = in[o];     // inverse: `in` is the shuffled-layout gradient
    }
}

// ---------------------------------------------------------------- tap flatten --------------------------
// [N][H][W][C] -> [N][W][C][H]: row (n,w) of the tap FC input with K index c*H + h, exactly the reference's
// reshape(B, C*H, W).transpose(1,2) of an NCHW tensor (model/ResNetSE34V2.py:160-162).  inverse=1 maps back.
// Per image this is the transpose of an H x (W*C) matrix (inverse: of a (W*C) x H one): 32 x 32 tiles through LDS, both sides coalesced
// (the element-wise form read with a stride of W*C floats: 64 us per tap at B = 128).
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float tile[32][33];
    const long base = (long)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + 8 * j][tx] = in[base + (long)r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (r < R && c < Cc) out[base + (long)c * R + r] = tile[tx][ty + 8 * j];
    }
}

// ---------------------------------------------------------------- speaker-softmax blending -------------
// logits [B][3][L] -> w = softmax over the 3 granularities; blend_i[b,t,:] = sum_g feat_g[b,t,:] * w[b,g,i]
// (reference: scripts/model/ResNetSE34V2.py:202-212).  feats are [B][T*F] rows.
__global__ void blend_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ f0, const float* __restrict__ f1,
                                 const float* __restrict__ f2, float* __restrict__ wout, float* __restrict__ blend, int B,
                                 int L, int TF) {
    const int b = blockIdx.x;
    __shared__ float w[3 * 8];
    if (threadIdx.x < L) {
        int i = threadIdx.x;
        float a0 = logits[(b * 3 + 0) * L + i], a1 = logits[(b * 3 + 1) * L + i], a2 = logits[(b * 3 + 2) * L + i];
        float m = fmaxf(a0, fmaxf(a1, a2));
        float e0 = expf(a0 - m), e1 = expf(a1 - m), e2 = expf(a2 - m), s = e0 + e1 + e2;
        w[0 * L + i] = e0 / s; w[1 * L + i] = e1 / s; w[2 * L + i] = e2 / s;
        wout[(b * 3 + 0) * L + i] = e0 / s; wout[(b * 3 + 1) * L + i] = e1 / s; wout[(b * 3 + 2) * L + i] = e2 / s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TF; e += blockDim.x) {
        float x0 = f0[(long)b * TF + e], x1 = f1[(long)b * TF + e], x2 = f2[(long)b * TF + e];
        for (int i = 0; i < L; ++i) blend[((long)i * B + b) * TF + e] = x0 * w[i] + x1 * w[L + i] + x2 * w[2 * L + i];
    }
}
// inputs: dblend [L][B][TF], dw_ext [B][3][L] (gradient flowing into the returned `weight`, may be null);
// outputs: df_g (+= into provided buffers, which the caller pre-fills with the direct feat gradients or zeros),
// dlogits [B][3][L].
__global__ __launch_bounds__(256) void blend_bwd_kernel(const float* __restrict__ dblend, const float* __restrict__ dw_ext,
                                                        const float* __restrict__ w, const float* __restrict__ f0,
                                                        const float* __restrict__ f1, const float* __restrict__ f2,
                                                        float* __restrict__ df0, float* __restrict__ df1, float* __restrict__ df2,
                                                        float* __restrict__ dlogits, int B, int L, int TF) {
    const int b = blockIdx.x;
    __shared__ float red[16];
    __shared__ float dw[3 * 8];
    float acc[3 * 8];
    for (int i = 0; i < 3 * L; ++i) acc[i] = 0.f;
    for (int e = threadIdx.x; e < TF; e += 256) {
        float x0 = f0[(long)b * TF + e], x1 = f1[(long)b * TF + e], x2 = f2[(long)b * TF + e];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        for (int i = 0; i < L; ++i) {
            float d = dblend[((long)i * B + b) * TF + e];
            g0 += d * w[(b * 3 + 0) * L + i]; g1 += d * w[(b * 3 + 1) * L + i]; g2 += d * w[(b * 3 + 2) * L + i];
            acc[i] += d * x0; acc[L + i] += d * x1; acc[2 * L + i] += d * x2;
        }
        df0[(long)b * TF + e] += g0; df1[(long)b * TF + e] += g1; df2[(long)b * TF + e] += g2;
    }
    for (int i = 0; i < 3 * L; ++i) {
        float s = block_sum(acc[i], red);
        if (threadIdx.x == 0) dw[i] = s + (dw_ext ? dw_ext[b * 3 * L + i] : 0.f);
    }
    __syncthreads();
    if (threadIdx.x < L) {           // softmax backward over g for level i
        int i = threadIdx.x;
        float w0 = w[(b * 3 + 0) * L + i], w1 = w[(b * 3 + 1) * L + i], w2 = w[(b * 3 + 2) * L + i];
        float dot = dw[i] * w0 + dw[L + i] * w1 + dw[2 * L + i] * w2;
        dlogits[(b * 3 + 0) * L + i] = w0 * (dw[i] - dot);
        dlogits[(b * 3 + 1) * L + i] = w1 * (dw[L + i] - dot);
        dlogits[(b * 3 + 2) * L + i] = w2 * (dw[2 * L + i] - dot);
    }
}

}  // namespace

extern "C" {

int ha2g_embedding_fwd_f32(const long* tok, const float* W, float* out, long n, int C, void* stream) {
    HA2G_REQUIRE(C % 4 == 0, "embedding: width %d must be a multiple of 4", C);
    if (n == 0) return 0;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(n * (C / 4))), dim3(EB), 0, (hipStream_t)stream, tok, W, out, n, C / 4);
    HA2G_CHECK_LAUNCH("embedding_fwd");
    return 0;
}
// dW [V][C] += scatter of dY [n][C] by tok [n]; `heavy` = the most frequent id (padding, 0); ws >= 64*C doubles
int ha2g_embedding_bwd_f32(const long* tok, const float* dY, float* dW, int n, int C, long heavy, float* ws, void* stream) {
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nchunk = n / 128 < 1 ? 1 : (n / 128 > 64 ? 64 : n / 128);
    if (heavy >= 0) {
        hipLaunchKernelGGL(embedding_bwd_heavy_partial, dim3(ceil_div(C, 64), nchunk), dim3(256), 0, st, tok, dY, (double*)ws, n, C, heavy);
        hipLaunchKernelGGL(embedding_bwd_heavy_final, dim3(ceil_div(C, 256)), dim3(256), 0, st, (const double*)ws, nchunk, C, dW, heavy);
    }
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(n, ceil_div(C, 64)), dim3(256), 0, st, tok, dY, dW, n, C, heavy);
    HA2G_CHECK_LAUNCH("embedding_bwd");
    return 0;
}
int ha2g_im2col1d_f32(const float* x, float* col, int B, int T, int C, int k, int dil, int pad_left, int To, void* stream) {
    long total = (long)B * To * C * k;
    if (total == 0) return 0;
    if (k == 2 && C % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)col & 15) == 0) {
        hipLaunchKernelGGL(im2col1d_k2_kernel, dim3(grid_for(total / 8)), dim3(EB), 0, (hipStream_t)stream, x, col, B, T, C / 4, dil, pad_left, To);
        HA2G_CHECK_LAUNCH("im2col1d_k2");
        return 0;
    }
    hipLaunchKernelGGL(im2col1d_kernel, dim3(grid_for(total)), dim3(EB), 0, (hipStream_t)stream, x, col, B, T, C, k, dil, pad_left, To);
    HA2G_CHECK_LAUNCH("im2col1d");
    return 0;
}
int ha2g_im2col1d_drop_supported(int C, int k) { return k == 2 && C % 4 == 0; }
int ha2g_im2col1d_drop_f32(const float* x, float* col, int B, int T, int C, int k, int dil, int pad_left, int To, float p, const void* rng_state,
                           unsigned stream_id, void* stream) {
    HA2G_REQUIRE(ha2g_im2col1d_drop_supported(C, k), "im2col1d_drop: k = %d, C = %d (k = 2, C %% 4 == 0)", k, C);
    HA2G_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)col & 15) == 0, "im2col1d_drop: 16-byte aligned buffers");
    HA2G_REQUIRE(p >= 0.f && p < 1.f && rng_state != nullptr, "im2col1d_drop: p = %f / null state", (double)p);
    const long total = (long)B * To * C * k;
    if (total == 0) return 0;
    hipLaunchKernelGGL(im2col1d_k2_drop_kernel, dim3(grid_for(total / 8)), dim3(EB), 0, (hipStream_t)stream, x, col, B, T, C / 4, dil, pad_left, To, p,
                       (const unsigned long long*)rng_state, stream_id);
    HA2G_CHECK_LAUNCH("im2col1d_k2_drop");
    return 0;
}
int ha2g_col2im1d_f32(const float* dcol, float* dx, int B, int T, int C, int k, int dil, int pad_left, int To, void* stream) {
    long total = (long)B * T * C;
    if (total == 0) return 0;
    if (k == 2 && C % 4 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dcol & 15) == 0) {
        hipLaunchKernelGGL(col2im1d_k2_kernel, dim3(grid_for(total / 4)), dim3(EB), 0, (hipStream_t)stream, dcol, dx, B, T, C / 4, dil, pad_left, To);
        HA2G_CHECK_LAUNCH("col2im1d_k2");
        return 0;
    }
    hipLaunchKernelGGL(col2im1d_kernel, dim3(grid_for(total)), dim3(EB), 0, (hipStream_t)stream, dcol, dx, B, T, C, k, dil, pad_left, To);
    HA2G_CHECK_LAUNCH("col2im1d");
    return 0;
}
int ha2g_weight_norm_fwd_f32(const float* g, const float* v, float* w, float* norm, int Cout, int n, void* stream) {
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, g, v, w, norm, n);
    HA2G_CHECK_LAUNCH("weight_norm_fwd");
    return 0;
}
int ha2g_weight_norm_bwd_f32(const float* dw, const float* g, const float* v, const float* norm, float* dg, float* dv, int Cout,
                             int n, float beta, void* stream) {
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dw, g, v, norm, dg, dv, n, beta);
    HA2G_CHECK_LAUNCH("weight_norm_bwd");
    return 0;
}
// `count` (<= 32) weight-norm layers with the same [Cout][n] shape in one launch; host arrays of device pointers
int ha2g_weight_norm_multi_fwd_f32(int count, const float* const* g, const float* const* v, float* const* w, float* const* norm, int Cout, int n,
                                   void* stream) {
    HA2G_REQUIRE(count >= 1 && count <= 32, "weight_norm_multi: 1..32 layers, got %d", count);
    WnMulti m{};
    for (int t = 0; t < count; ++t) { m.g[t] = g[t]; m.v[t] = v[t]; m.w[t] = w[t]; m.norm[t] = norm[t]; }
    hipLaunchKernelGGL(weight_norm_multi_fwd_kernel, dim3(Cout, count), dim3(256), 0, (hipStream_t)stream, m, n);
    HA2G_CHECK_LAUNCH("weight_norm_multi_fwd");
    return 0;
}
int ha2g_weight_norm_multi_bwd_f32(int count, const float* const* dw, const float* const* g, const float* const* v, const float* const* norm,
                                   float* const* dg, float* const* dv, int Cout, int n, float beta, void* stream) {
    HA2G_REQUIRE(count >= 1 && count <= 32, "weight_norm_multi: 1..32 layers, got %d", count);
    WnMulti m{};
    for (int t = 0; t < count; ++t) { m.dw[t] = dw[t]; m.g[t] = g[t]; m.v[t] = v[t]; m.norm[t] = const_cast<float*>(norm[t]); m.dg[t] = dg[t]; m.dv[t] = dv[t]; }
    hipLaunchKernelGGL(weight_norm_multi_bwd_kernel, dim3(Cout, count), dim3(256), 0, (hipStream_t)stream, m, n, beta);
    HA2G_CHECK_LAUNCH("weight_norm_multi_bwd");
    return 0;
}
// out = op(a, b, c); b / c may be null for unary ops (float4 path when every pointer is 16-byte aligned).
int ha2g_eltwise_f32(int op, const float* a, const float* b, const float* c, float* out, long n, float alpha, float beta,
                     void* stream) {
    if (n == 0) return 0;
    HA2G_REQUIRE(op >= 0 && op <= OP_LEAKY_A, "eltwise: unknown op %d", op);
    int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0 && op != OP_MUL_SCALAR;
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for(n, 4)), dim3(EB), 0, (hipStream_t)stream, op, a, b, c, out, n, alpha, beta, vec);
    HA2G_CHECK_LAUNCH("eltwise");
    return 0;
}
// state: device uint64[2] = {seed, step}.  out and/or mask may be null.
int ha2g_dropout_f32(const float* x, float* out, float* mask, long n, float p, const void* state, unsigned stream_id, void* stream) {
    if (n == 0) return 0;
    HA2G_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%f out of range", p);
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 4)), dim3(EB), 0, (hipStream_t)stream, x, out, mask, n, p,
                       (const unsigned long long*)state, stream_id, vec);
    HA2G_CHECK_LAUNCH("dropout");
    return 0;
}
// mode 1: out = relu(x * mask + b); mode 2: out = b > 0 ? x * mask : 0 (see dropout_kernel); the mask is the one ha2g_dropout_f32 draws for the same
// (state, stream_id, element)
int ha2g_dropout_fused_f32(const float* x, const float* b, float* out, long n, float p, const void* state, unsigned stream_id, int mode, void* stream) {
    if (n == 0) return 0;
    HA2G_REQUIRE(p >= 0.f && p < 1.f, "dropout_fused: p=%f out of range", p);
    HA2G_REQUIRE((mode == 1 || mode == 2) && x && b && out, "dropout_fused: mode %d (1, 2) / null operand", mode);
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 4)), dim3(EB), 0, (hipStream_t)stream, x, out, (float*)nullptr, n, p,
                       (const unsigned long long*)state, stream_id, vec, b, mode);
    HA2G_CHECK_LAUNCH("dropout_fused");
    return 0;
}
// out = x * mask over the elements [elem_offset, elem_offset + n) of the tensor the mask of (state, stream_id) is defined on: the backward of a dropout
// whose gradient arrives for a ROW SLICE only (the GRU's inter-layer dropout under the fused chains: 128 of 384 rows carry gradient)
int ha2g_dropout_slice_f32(const float* x, float* out, long n, long elem_offset, float p, const void* state, unsigned stream_id, void* stream) {
    if (n == 0) return 0;
    HA2G_REQUIRE(p >= 0.f && p < 1.f, "dropout_slice: p=%f out of range", p);
    HA2G_REQUIRE(elem_offset >= 0 && elem_offset % 4 == 0, "dropout_slice: offset %ld must be a non-negative multiple of 4", elem_offset);
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 4)), dim3(EB), 0, (hipStream_t)stream, x, out, (float*)nullptr, n, p,
                       (const unsigned long long*)state, stream_id, vec, (const float*)nullptr, 0, elem_offset / 4);
    HA2G_CHECK_LAUNCH("dropout_slice");
    return 0;
}
int ha2g_dirsum_f32(const float* y, float* out, long rows, int H, int inverse, void* stream) {
    HA2G_REQUIRE(H % 4 == 0, "dirsum: H %% 4");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(dirsum_kernel, dim3(grid_for(rows * (H / 4))), dim3(EB), 0, (hipStream_t)stream, y, out, rows, H / 4, inverse);
    HA2G_CHECK_LAUNCH("dirsum");
    return 0;
}
int ha2g_rng_advance(void* state, void* stream) {
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)state);
    HA2G_CHECK_LAUNCH("rng_advance");
    return 0;
}
// NHWC pixel shuffle; in [N,H,W,Cout*r*r] -> out [N,H*r,W*r,Cout].  inverse=1: `in` is the gradient in the
// shuffled layout and `out` receives the gradient in the unshuffled layout.
int ha2g_pixel_shuffle_f32(const float* in, float* out, int N, int H, int W, int Cout, int r, int inverse, void* stream) {
    long total = (long)N * H * r * W * r * Cout;
    if (total == 0) return 0;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for(total)), dim3(EB), 0, (hipStream_t)stream, in, out, N, H, W, Cout, r, inverse);
    HA2G_CHECK_LAUNCH("pixel_shuffle");
    return 0;
}
int ha2g_nhwc_to_nwch_f32(const float* in, float* out, int N, int H, int W, int C, int inverse, void* stream) {
    long total = (long)N * H * W * C;
    if (total == 0) return 0;
    const int R = inverse ? W * C : H, Cc = inverse ? H : W * C;       // in [N][R][Cc] -> out [N][Cc][R]
    HA2G_REQUIRE(N <= 65535 && (R + 31) / 32 <= 65535, "nhwc_to_nwch: grid too large");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, N), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc);
    HA2G_CHECK_LAUNCH("nhwc_to_nwch");
    return 0;
}
int ha2g_blend_fwd_f32(const float* logits, const float* f0, const float* f1, const float* f2, float* w, float* blend, int B, int L,
                       int TF, void* stream) {
    HA2G_REQUIRE(L <= 8, "blend: pose_level %d > 8", L);
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, f0, f1, f2, w, blend, B, L, TF);
    HA2G_CHECK_LAUNCH("blend_fwd");
    return 0;
}
int ha2g_blend_bwd_f32(const float* dblend, const float* dw_ext, const float* w, const float* f0, const float* f1, const float* f2,
                       float* df0, float* df1, float* df2, float* dlogits, int B, int L, int TF, void* stream) {
    HA2G_REQUIRE(L <= 8, "blend: pose_level %d > 8", L);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dblend, dw_ext, w, f0, f1, f2, df0, df1, df2,
                       dlogits, B, L, TF);
    HA2G_CHECK_LAUNCH("blend_bwd");
    return 0;
}

}  // extern "C"
