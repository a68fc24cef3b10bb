// Generator input pack / hierarchy scatter and the scalar loss combination -- the "gen_pack_input" row of the hot path
// (reference: scripts/train_eval/train_hierarchy.py:153-169 pre_seq construction + coarse-to-fine scatter, expressive twin
// train_hierarchy_expressive.py:163-212; scripts/model/hierarchy_net.py:121-141 torch.cat of the GRU input; loss assembly
// train_hierarchy.py:226-262).  Pure layout work, HBM-bound, one launch each instead of ~10 slice / cat / fill launches.
#include "common.h"

namespace {

constexpr int EB = 256;
inline int grid_for(long n) { long g = (n + EB - 1) / EB; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// pre_seq[r][t][c], c in [0, P]:  t < n_pre: (c < P ? target[r][t][c] : 1);  t >= n_pre: (map[c] >= 0 ? prev[r][t][map[c]] : 0)
// map = the level's scatter table applied in the reference's assignment order (later slices win), incl. the expressive
// step's one-column shift of the 15 head values.
__global__ void pre_seq_fwd_kernel(const float* __restrict__ target, const float* __restrict__ prev, const int* __restrict__ map,
                                   float* __restrict__ out, long rows_t, int T, int P, int Pprev, int n_pre) {
    const int W = P + 1;
    const long total = rows_t * W;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int c = (int)(i % W);
        const long rt = i / W;
        const int t = (int)(rt % T);
        float v;
        if (t < n_pre) v = c < P ? target[rt * P + c] : 1.f;
        else {
            const int m = prev ? map[c] : -1;
            v = m >= 0 ? prev[rt * Pprev + m] : 0.f;
        }
        out[i] = v;
    }
}
// dprev[r][t][j] = t >= n_pre ? sum over the (at most two) pre_seq columns fed by j of dpre : 0
__global__ void pre_seq_bwd_kernel(const float* __restrict__ dpre, const int* __restrict__ inv, float* __restrict__ dprev, long rows_t,
                                   int T, int P, int Pprev, int n_pre) {
    const long total = rows_t * Pprev;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int j = (int)(i % Pprev);
        const long rt = i / Pprev;
        const int t = (int)(rt % T);
        float s = 0.f;
        if (t >= n_pre) {
            const int c0 = inv[2 * j], c1 = inv[2 * j + 1];
            if (c0 >= 0) s = dpre[rt * (P + 1) + c0];
            if (c1 >= 0) s += dpre[rt * (P + 1) + c1];
        }
        dprev[i] = s;
    }
}

// in_data[r][t][:] = [a (Wa) | b (Wb) | c (Wc) | z[r] (Wz)]  (hierarchy_net.py:121-141: pre_seq, audio, text, z expanded over t)
__global__ void gen_concat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                      const float* __restrict__ z, float* __restrict__ out, long rows_t, int T, int Wa, int Wb, int Wc, int Wz) {
    const int W = Wa + Wb + Wc + Wz;
    const long total = rows_t * W;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        const int col = (int)(i % W);
        const long rt = i / W;
        float v;
        if (col < Wa) v = a[rt * Wa + col];
        else if (col < Wa + Wb) v = b[rt * Wb + col - Wa];
        else if (col < Wa + Wb + Wc) v = c[rt * Wc + col - Wa - Wb];
        else v = z[(rt / T) * Wz + col - Wa - Wb - Wc];
        out[i] = v;
    }
}
// the inverse: column blocks back to their sources; dz[r][k] = sum_t d[r][t][..+k] (fixed order => deterministic)
__global__ void gen_concat_bwd_kernel(const float* __restrict__ d, float* __restrict__ da, float* __restrict__ db, float* __restrict__ dc,
                                      float* __restrict__ dz, long rows, int T, int Wa, int Wb, int Wc, int Wz) {
    const int W = Wa + Wb + Wc + Wz, Wabc = Wa + Wb + Wc;
    const long n1 = rows * T * Wabc, total = n1 + rows * Wz;
    for (long i = (long)blockIdx.x * EB + threadIdx.x; i < total; i += (long)gridDim.x * EB) {
        if (i < n1) {
            const int col = (int)(i % Wabc);
            const long rt = i / Wabc;
            const float v = d[rt * W + col];
            if (col < Wa) { if (da) da[rt * Wa + col] = v; }
            else if (col < Wa + Wb) { if (db) db[rt * Wb + col - Wa] = v; }
            else if (dc) dc[rt * Wc + col - Wa - Wb] = v;
        } else if (dz) {
            const long k = i - n1;
            const long r = k / Wz;
            const int zc = (int)(k % Wz);
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += d[(r * T + t) * W + Wabc + zc];
            dz[k] = s;
        }
    }
}

// Sliding-window synthesis (scripts/synthesize_hierarchy.py:150-161): window i covers frames [i*(T-n), i*(T-n)+T); its first n frames
// overlap the previous window's last n and are cross-faded  out[j] = prev[j]*(n-j)/(n+1) + next[j]*(j+1)/(n+1)  -- same operation
// order in fp32 as the reference's numpy expression -- the rest is copied.
__global__ void window_blend_kernel(const float* __restrict__ win, float* __restrict__ out, int first, int T, int n, int P) {
    const int total = T * P;
    for (int i = blockIdx.x * EB + threadIdx.x; i < total; i += gridDim.x * EB) {
        const int j = i / P;
        float v = win[i];
        if (!first && j < n) v = out[i] * (float)(n - j) / (float)(n + 1) + v * (float)(j + 1) / (float)(n + 1);
        out[i] = v;
    }
}

constexpr int MAXTERMS = 24;
struct Terms { const float* p[MAXTERMS]; float w[MAXTERMS]; int n; };
// out[0] = sum_i w_i * *p_i, accumulated left to right in fp32 (the order the reference's Python expression adds them)
__global__ void weighted_sum_kernel(Terms tm, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = tm.w[0] * *tm.p[0];
        for (int i = 1; i < tm.n; ++i) s += tm.w[i] * *tm.p[i];
        out[0] = s;
    }
}
struct Weights { float w[MAXTERMS]; int n; };
__global__ void weighted_sum_bwd_kernel(Weights wt, const float* __restrict__ g, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < wt.n) out[i] = wt.w[i] * g[0];
}

}  // namespace

extern "C" {

int ha2g_pre_seq_fwd_f32(const float* target, const float* prev, const int* map, float* out, long rows, int T, int P, int Pprev,
                         int n_pre, void* stream) {
    if (rows == 0) return 0;
    HA2G_REQUIRE(prev == nullptr || map != nullptr, "pre_seq: a scatter map is required with a coarser level");
    hipLaunchKernelGGL(pre_seq_fwd_kernel, dim3(grid_for(rows * T * (P + 1))), dim3(EB), 0, (hipStream_t)stream, target, prev, map, out,
                       rows * T, T, P, Pprev, n_pre);
    HA2G_CHECK_LAUNCH("pre_seq_fwd");
    return 0;
}
int ha2g_pre_seq_bwd_f32(const float* dpre, const int* inv, float* dprev, long rows, int T, int P, int Pprev, int n_pre, void* stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(pre_seq_bwd_kernel, dim3(grid_for(rows * T * Pprev)), dim3(EB), 0, (hipStream_t)stream, dpre, inv, dprev, rows * T,
                       T, P, Pprev, n_pre);
    HA2G_CHECK_LAUNCH("pre_seq_bwd");
    return 0;
}
int ha2g_gen_concat_fwd_f32(const float* a, const float* b, const float* c, const float* z, float* out, long rows, int T, int Wa, int Wb,
                            int Wc, int Wz, void* stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(gen_concat_fwd_kernel, dim3(grid_for(rows * T * (Wa + Wb + Wc + Wz))), dim3(EB), 0, (hipStream_t)stream, a, b, c, z,
                       out, rows * T, T, Wa, Wb, Wc, Wz);
    HA2G_CHECK_LAUNCH("gen_concat_fwd");
    return 0;
}
int ha2g_gen_concat_bwd_f32(const float* d, float* da, float* db, float* dc, float* dz, long rows, int T, int Wa, int Wb, int Wc, int Wz,
                            void* stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(gen_concat_bwd_kernel, dim3(grid_for(rows * T * (Wa + Wb + Wc) + rows * Wz)), dim3(EB), 0, (hipStream_t)stream, d, da,
                       db, dc, dz, rows, T, Wa, Wb, Wc, Wz);
    HA2G_CHECK_LAUNCH("gen_concat_bwd");
    return 0;
}
/* out_all [frames][P] with frames >= index*(T-n_pre)+T: blend window `index` ([T][P]) into the running sequence */
int ha2g_window_blend_f32(const float* win, float* out_all, int index, int T, int n_pre, int P, void* stream) {
    HA2G_REQUIRE(index >= 0 && n_pre >= 0 && n_pre < T, "window_blend: bad geometry");
    hipLaunchKernelGGL(window_blend_kernel, dim3(grid_for((long)T * P)), dim3(EB), 0, (hipStream_t)stream, win,
                       out_all + (long)index * (T - n_pre) * P, index == 0, T, n_pre, P);
    HA2G_CHECK_LAUNCH("window_blend");
    return 0;
}
/* terms_host: HOST array of n device pointers to fp32 scalars; weights_host: HOST array of n weights (both read at call time) */
int ha2g_weighted_sum_f32(const void* const* terms_host, const float* weights_host, int n, float* out, void* stream) {
    HA2G_REQUIRE(n >= 1 && n <= MAXTERMS, "weighted_sum: %d terms (1..%d)", n, MAXTERMS);
    Terms tm{};
    tm.n = n;
    for (int i = 0; i < n; ++i) { tm.p[i] = (const float*)terms_host[i]; tm.w[i] = weights_host[i]; }
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tm, out);
    HA2G_CHECK_LAUNCH("weighted_sum");
    return 0;
}
/* out[i] = weights_host[i] * g[0]: the n upstream gradients of the terms in one launch */
int ha2g_weighted_sum_bwd_f32(const float* weights_host, int n, const float* g, float* out, void* stream) {
    HA2G_REQUIRE(n >= 1 && n <= MAXTERMS, "weighted_sum_bwd: %d terms (1..%d)", n, MAXTERMS);
    Weights wt{};
    wt.n = n;
    for (int i = 0; i < n; ++i) wt.w[i] = weights_host[i];
    hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, wt, g, out);
    HA2G_CHECK_LAUNCH("weighted_sum_bwd");
    return 0;
}

}  // extern "C"
