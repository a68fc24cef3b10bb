// Loss terms of train_iter_hierarchy (reference scripts/train_eval/train_hierarchy.py:54-68,173-262).
// Every kernel produces the scalar term AND the unit gradient w.r.t. its differentiable inputs in one go
// (the autograd wrapper only rescales by the upstream scalar), reductions are fixed-order (deterministic).
#include "common.h"
#include "../../include/ha2g_hip.h"   // ha2g_gemm_f32 / ha2g_colsum_f32 used by the contrastive loss

namespace {

// deterministic sum of n floats (n up to a few 100k) by ONE block: out = scale * sum(x) (+ out if accumulate)
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, long n, float* __restrict__ out, float scale,
                                                   int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + scale * s;
}

// ---- Huber: smooth_l1(x/beta, y/beta)*beta == 0.5 d^2/beta (|d|<beta) else |d| - beta/2 ------------------
__global__ __launch_bounds__(256) void huber_kernel(const float* __restrict__ x, const float* __restrict__ y, long n, float beta,
                                                    float gscale, float* __restrict__ part, float* __restrict__ dx) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float d = x[i] - y[i], ad = fabsf(d);
        bool q = ad < beta;
        s += q ? 0.5f * d * d / beta : ad - 0.5f * beta;
        if (dx) dx[i] = gscale * (q ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- KLD = -0.5 * mean(1 + lv - mu^2 - exp(lv)) ---------------------------------------------------------
__global__ __launch_bounds__(256) void kld_kernel(const float* __restrict__ mu, const float* __restrict__ lv, int n,
                                                  float* __restrict__ loss, float* __restrict__ dmu, float* __restrict__ dlv) {
    __shared__ float red[16];
    float s = 0.f;
    const float inv = 1.f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        float m = mu[i], l = lv[i], e = expf(l);
        s += 1.f + l - m * m - e;
        dmu[i] = m * inv;
        dlv[i] = -0.5f * (1.f - e) * inv;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) *loss = -0.5f * s * inv;
}

// ---- diversity regulariser (train_hierarchy.py:213-222): one block per sample --------------------------
__global__ __launch_bounds__(256) void divreg_kernel(const float* __restrict__ out, const float* __restrict__ rnd,
                                                     const float* __restrict__ z, const float* __restrict__ zr, int B, int TP,
                                                     int Z, float beta, float* __restrict__ per_sample, float* __restrict__ dout) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const float* o = out + (long)b * TP;
    const float* r = rnd + (long)b * TP;
    float s = 0.f;
    for (int i = threadIdx.x; i < TP; i += 256) {
        float d = o[i] - r[i], ad = fabsf(d);
        s += ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
    }
    const float pose = block_sum(s, red);
    float zs = 0.f;
    for (int i = threadIdx.x; i < Z; i += 256) zs += fabsf(z[b * Z + i] - zr[b * Z + i]);
    const float zl1 = block_sum(zs, red) / (float)Z;
    const float v = -(pose / (zl1 + 1.0e-5f));
    const bool live = v >= -1000.f;                           // clamp(min=-1000) passes gradient on its boundary
    if (threadIdx.x == 0) per_sample[b] = live ? v : -1000.f;
    const float g = live ? -1.f / ((zl1 + 1.0e-5f) * (float)B) : 0.f;
    for (int i = threadIdx.x; i < TP; i += 256) {
        float d = o[i] - r[i], ad = fabsf(d);
        dout[(long)b * TP + i] = g * (ad < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
}

// ---- physical angle prior (train_hierarchy.py:242-262; expressive :421-447): one thread per (b,t) row ------------
// Bones >= nb are synthetic "palm normals": cross products of two raw (un-normalised) bone vectors, appended before
// the normalisation exactly as the reference concatenates them (train_hierarchy_expressive.py:430-432).
constexpr int MAXV = 48;
struct PalmSpec { int n; int a[4]; int b[4]; };

__device__ __forceinline__ void phys_vec(const float* o, const float* md, int nb, const PalmSpec& pm, int idx, float* v) {
    if (idx < nb) {
        for (int c = 0; c < 3; ++c) v[c] = o[idx * 3 + c] + md[idx * 3 + c];
    } else {
        const int pa = pm.a[idx - nb], pb = pm.b[idx - nb];
        float u[3], w[3];
        for (int c = 0; c < 3; ++c) { u[c] = o[pa * 3 + c] + md[pa * 3 + c]; w[c] = o[pb * 3 + c] + md[pb * 3 + c]; }
        v[0] = u[1] * w[2] - u[2] * w[1]; v[1] = u[2] * w[0] - u[0] * w[2]; v[2] = u[0] * w[1] - u[1] * w[0];
    }
}
// scatter d L / d v (v = raw vector of bone idx) into the pose gradient; c = u x w  =>  dL/du = w x g, dL/dw = g x u
__device__ __forceinline__ void phys_grad(const float* o, const float* md, int nb, const PalmSpec& pm, int idx, const float* gv, float* g) {
    if (idx < nb) {
        for (int c = 0; c < 3; ++c) g[idx * 3 + c] += gv[c];
    } else {
        const int pa = pm.a[idx - nb], pb = pm.b[idx - nb];
        float u[3], w[3];
        for (int c = 0; c < 3; ++c) { u[c] = o[pa * 3 + c] + md[pa * 3 + c]; w[c] = o[pb * 3 + c] + md[pb * 3 + c]; }
        g[pa * 3 + 0] += w[1] * gv[2] - w[2] * gv[1]; g[pa * 3 + 1] += w[2] * gv[0] - w[0] * gv[2]; g[pa * 3 + 2] += w[0] * gv[1] - w[1] * gv[0];
        g[pb * 3 + 0] += gv[1] * u[2] - gv[2] * u[1]; g[pb * 3 + 1] += gv[2] * u[0] - gv[0] * u[2]; g[pb * 3 + 2] += gv[0] * u[1] - gv[1] * u[0];
    }
}

__global__ __launch_bounds__(64) void phys_kernel(const float* __restrict__ out, const float* __restrict__ mean_dir, int rows,
                                                  int nb, PalmSpec pm, const int* __restrict__ pairs, int npairs,
                                                  const float* __restrict__ avg, const float* __restrict__ var,
                                                  float* __restrict__ per_row, float* __restrict__ dout) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    const int P = nb * 3;
    const float* o = out + (long)r * P;
    float* g = dout + (long)r * P;
    for (int i = 0; i < P; ++i) g[i] = 0.f;
    float total = 0.f;
    const float inv_rows = 1.f / (float)rows, PI = 3.14159265358979323846f;
    for (int k = 0; k < npairs; ++k) {
        const int a = pairs[2 * k], b = pairs[2 * k + 1];
        float va[3], vb[3];
        phys_vec(o, mean_dir, nb, pm, a, va);
        phys_vec(o, mean_dir, nb, pm, b, vb);
        float na = sqrtf(va[0] * va[0] + va[1] * va[1] + va[2] * va[2]), nbn = sqrtf(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
        float da = fmaxf(na, 1e-12f), db = fmaxf(nbn, 1e-12f);
        float ua[3], ub[3];
        for (int c = 0; c < 3; ++c) { ua[c] = va[c] / da; ub[c] = vb[c] / db; }
        float ip = ua[0] * ub[0] + ua[1] * ub[1] + ua[2] * ub[2];
        const float lo = -1.f + 1e-7f, hi = 1.f - 1e-7f;
        const bool inside = ip >= lo && ip <= hi;
        float ipc = fminf(fmaxf(ip, lo), hi);
        float ang = acosf(ipc) / PI;
        float diff = ang - avg[k];
        total += diff * diff / (2.f * var[k]);
        float gip = inside ? (diff / var[k]) * (-1.f / (PI * sqrtf(1.f - ipc * ipc))) * inv_rows : 0.f;
        // through the two normalisations: d v = (d u - u (u . d u)) / |v|
        float dua[3] = {gip * ub[0], gip * ub[1], gip * ub[2]}, dub[3] = {gip * ua[0], gip * ua[1], gip * ua[2]};
        float pa = ua[0] * dua[0] + ua[1] * dua[1] + ua[2] * dua[2], pb = ub[0] * dub[0] + ub[1] * dub[1] + ub[2] * dub[2];
        float gva[3], gvb[3];
        for (int c = 0; c < 3; ++c) {
            gva[c] = (na > 1e-12f ? dua[c] - ua[c] * pa : dua[c]) / da;
            gvb[c] = (nbn > 1e-12f ? dub[c] - ub[c] * pb : dub[c]) / db;
        }
        phys_grad(o, mean_dir, nb, pm, a, gva, g);
        phys_grad(o, mean_dir, nb, pm, b, gvb, g);
    }
    per_row[r] = total * inv_rows;
}

// ---- GAN terms: mode 0  gen = -mean(log(d+1e-8));  mode 1  dis = -mean(log(real+1e-8) + log(1-fake+1e-8)) ----
__global__ __launch_bounds__(256) void gan_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b, int n,
                                                  float* __restrict__ loss, float* __restrict__ da, float* __restrict__ db) {
    __shared__ float red[16];
    float s = 0.f;
    const float inv = 1.f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (mode == 0) {
            s += logf(a[i] + 1e-8f);
            da[i] = -inv / (a[i] + 1e-8f);
        } else {
            s += logf(a[i] + 1e-8f) + logf(1.f - b[i] + 1e-8f);
            da[i] = -inv / (a[i] + 1e-8f);
            db[i] = inv / (1.f - b[i] + 1e-8f);
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) *loss = -s * inv;
}

// ---- softmax contrastive loss (train_hierarchy.py:54-68; expressive variant :107-121) -------------------
constexpr int CD = 32;      // feature width of text / audio features

__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ xn, float* __restrict__ nrm, int N) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float s = 0.f, v[CD];
    for (int d = 0; d < CD; ++d) { v[d] = x[(long)i * CD + d]; s += v[d] * v[d]; }
    float n = fmaxf(sqrtf(s), 1e-12f);
    nrm[i] = n;
    for (int d = 0; d < CD; ++d) xn[(long)i * CD + d] = v[d] / n;
}

__device__ __forceinline__ float logit_of(float dist, int expressive) {
    return expressive ? 1.0f / dist : fmaxf(1.0f / (dist + 1e-8f), 1e-8f);
}
// d logit / d dist
__device__ __forceinline__ float dlogit_of(float dist, float l, int expressive) {
    if (expressive) return -l * l;
    return (1.0f / (dist + 1e-8f) > 1e-8f) ? -l * l : 0.f;
}

// Pair work, three stages per block of R rows (R x N floats of scratch, R chosen to fit the workspace):
//   1. contrastive_dist_kernel   D[i][j] = |a_i - b_j|  -- 64x64 tile per block, 4x4 pairs per thread out of LDS (VALU-bound,
//                                N^2/4096 blocks fill the chip; the reference materialises an N x N x 32 tensor instead);
//   2. contrastive_row_kernel    one wave per row: online log-sum-exp of the logits, loss_i, then the row of coefficients
//                                c_ij = d loss / d dist_ij / dist_ij  written over D, plus its row sum;
//   3. MFMA GEMMs                d a_i = a_i * sum_j c_ij - (C b)_i ,   d b_j = b_j * sum_i c_ij - (C^T a)_j
//                                (ha2g_gemm_f32; column sums by ha2g_colsum_f32).
// Every reduction has a fixed order => deterministic.
constexpr int LDT = CD + 1;

__global__ __launch_bounds__(256) void contrastive_dist_kernel(const float* __restrict__ a, const float* __restrict__ b, int N,
                                                               int r0, int R, float* __restrict__ D) {
    __shared__ float ta[64 * LDT], tb[64 * LDT];
    const int i0 = r0 + blockIdx.x * 64, j0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * CD; e += 256) {
        const int r = e / CD, d = e % CD;
        ta[r * LDT + d] = (i0 + r < r0 + R) ? a[(long)(i0 + r) * CD + d] : 0.f;
        tb[r * LDT + d] = (j0 + r < N) ? b[(long)(j0 + r) * CD + d] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // rows ty + 16*r, columns tx + 16*c
    float s[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[r][c] = 0.f;
#pragma unroll 8
    for (int d = 0; d < CD; ++d) {
        float av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = ta[(ty + 16 * r) * LDT + d];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = tb[(tx + 16 * c) * LDT + d];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = av[r] - bv[c]; s[r][c] += t * t; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 16 * r;
        if (i >= r0 + R) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx + 16 * c;
            if (j < N) D[(long)(i - r0) * N + j] = sqrtf(s[r][c]);
        }
    }
}

__global__ __launch_bounds__(256) void contrastive_row_kernel(float* __restrict__ D, const float* __restrict__ an, int N, int r0, int R,
                                                              int expressive, float* __restrict__ loss_i, float* __restrict__ dan) {
    const int lane = threadIdx.x & 63, il = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (il >= R) return;
    const int i = r0 + il;
    float* row = D + (long)il * N;
    float m = -INFINITY, z = 0.f, lii = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float l = logit_of(row[j], expressive);
        if (j == i) lii = l;
        if (l > m) { z = z * expf(m - l) + 1.f; m = l; } else z += expf(l - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), z2 = __shfl_xor(z, o, 64);
        const float mm = fmaxf(m, m2);
        const float za = (m == -INFINITY) ? 0.f : z * expf(m - mm), zb = (m2 == -INFINITY) ? 0.f : z2 * expf(m2 - mm);
        const bool low = ((lane & o) == 0);               // both partners add in the same order
        z = low ? za + zb : zb + za;
        m = mm;
        lii += __shfl_xor(lii, o, 64);
    }
    if (lane == 0) loss_i[i] = (m + logf(z)) - lii;
    const float invN = 1.f / (float)N, invz = 1.f / z;
    float csum = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float dist = row[j];
        const float l = logit_of(dist, expressive);
        const float p = expf(l - m) * invz - ((j == i) ? 1.f : 0.f);
        const float c = dist > 0.f ? p * dlogit_of(dist, l, expressive) / dist * invN : 0.f;    // norm backward: 0 at dist = 0
        row[j] = c;
        csum += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) csum += __shfl_xor(csum, o, 64);
    if (lane < CD) dan[(long)i * CD + lane] = an[(long)i * CD + lane] * csum;
}

// gradient through x / max(|x|, 1e-12): dx = (dn - n (n . dn)) / |x|
// cs (nullable): the gradient w.r.t. the normalised row is dn + xn * cs[i]
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ nrm,
                                                          const float* __restrict__ dn, const float* __restrict__ cs,
                                                          float* __restrict__ dx, int N) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float c = cs ? cs[i] : 0.f;
    float p = 0.f;
    for (int d = 0; d < CD; ++d) p += xn[(long)i * CD + d] * (dn[(long)i * CD + d] + xn[(long)i * CD + d] * c);
    float n = nrm[i];
    for (int d = 0; d < CD; ++d) {
        float g = dn[(long)i * CD + d] + xn[(long)i * CD + d] * c;
        dx[(long)i * CD + d] = (n > 1e-12f ? g - xn[(long)i * CD + d] * p : g) / n;
    }
}

}  // namespace

extern "C" {

int ha2g_sum_f32(const float* x, long n, float* out, float scale, int accumulate, void* stream) {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out, scale, accumulate);
    HA2G_CHECK_LAUNCH("sum");
    return 0;
}

// loss = mean Huber_beta(x - y); dx (nullable) = d loss / d x.  ws: >= 1024 floats.
int ha2g_huber_f32(const float* x, const float* y, long n, float beta, float* loss, float* dx, float* ws, void* stream) {
    HA2G_REQUIRE(n > 0, "huber: empty input");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n + 4095) / 4096);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(huber_kernel, dim3(nb), dim3(256), 0, st, x, y, n, beta, 1.f / (float)n, ws, dx);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, ws, (long)nb, loss, 1.f / (float)n, 0);
    HA2G_CHECK_LAUNCH("huber");
    return 0;
}

int ha2g_kld_f32(const float* mu, const float* logvar, int n, float* loss, float* dmu, float* dlogvar, void* stream) {
    hipLaunchKernelGGL(kld_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, logvar, n, loss, dmu, dlogvar);
    HA2G_CHECK_LAUNCH("kld");
    return 0;
}

// out/rnd [B][TP], z/zr [B][Z]; loss = mean_b clamp(-(huber_sum_b)/(mean|z-zr| + 1e-5), min=-1000); ws >= B floats
int ha2g_divreg_f32(const float* out, const float* rnd, const float* z, const float* zr, int B, int TP, int Z, float beta,
                    float* loss, float* dout, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(divreg_kernel, dim3(B), dim3(256), 0, st, out, rnd, z, zr, B, TP, Z, beta, ws, dout);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, ws, (long)B, loss, 1.f / (float)B, 0);
    HA2G_CHECK_LAUNCH("divreg");
    return 0;
}

// out [rows][nb*3]; pairs int32 [npairs][2] (indices >= nb address the palm normals); palm: host int32 [npalm][2]
// (bone pairs whose raw cross product forms synthetic bone nb+i), npalm <= 4; avg/var [npairs]; ws >= rows floats
int ha2g_phys_angle_f32(const float* out, const float* mean_dir, int rows, int nb, const int* pairs, int npairs, const float* avg,
                        const float* var, const int* palm_host, int npalm, float* loss, float* dout, float* ws, void* stream) {
    HA2G_REQUIRE(nb <= MAXV, "phys: too many bones (%d)", nb);
    HA2G_REQUIRE(npalm >= 0 && npalm <= 4, "phys: at most 4 palm normals");
    PalmSpec pm{};
    pm.n = npalm;
    for (int i = 0; i < npalm; ++i) { pm.a[i] = palm_host[2 * i]; pm.b[i] = palm_host[2 * i + 1]; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(phys_kernel, dim3(ceil_div(rows, 64)), dim3(64), 0, st, out, mean_dir, rows, nb, pm, pairs, npairs, avg, var, ws, dout);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, ws, (long)rows, loss, 1.f, 0);
    HA2G_CHECK_LAUNCH("phys");
    return 0;
}

int ha2g_gan_loss_f32(int mode, const float* a, const float* b, int n, float* loss, float* da, float* db, void* stream) {
    hipLaunchKernelGGL(gan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mode, a, b, n, loss, da, db);
    HA2G_CHECK_LAUNCH("gan_loss");
    return 0;
}

// a, b [N][32] raw features (rows = first argument of the reference's criterion = text); loss scalar;
// da, db unit gradients w.r.t. the raw inputs.  ws >= ha2g_contrastive_workspace_floats(N) floats: per-row vectors, the
// split-K / column-sum scratch of the inner GEMMs and an R x N block of the pair matrix (R = N when it fits in 80 MB,
// else the rows are processed in blocks -- same arithmetic, fixed order).
static int contrastive_rows_per_block(int N) {
    long r = (80L << 20) / (4L * N);
    r = r / 64 * 64;
    if (r < 64) r = 64;
    return (int)(r > N ? N : r);
}
static long contrastive_gemm_ws_floats(int N) { return 40L * N * CD + 64; }                  // <= 40 split-K partials of [N][32]
long ha2g_contrastive_workspace_floats(int N) {
    return (long)N * (4 * CD + 6) + contrastive_gemm_ws_floats(N) + ha2g_colsum_workspace_bytes(N) / 4 +
           (long)contrastive_rows_per_block(N) * N + 64;
}
int ha2g_contrastive_f32(const float* a, const float* b, int N, int expressive, float* loss, float* da, float* db, float* ws,
                         void* stream) {
    HA2G_REQUIRE(N > 0, "contrastive: empty input");
    hipStream_t st = (hipStream_t)stream;
    float* an = ws; float* bn = an + (long)N * CD; float* dan = bn + (long)N * CD; float* dbn = dan + (long)N * CD;
    float* na = dbn + (long)N * CD; float* nbv = na + N; float* li = nbv + N; float* cs = li + N;
    float* gws = cs + 3 * N;                                   // (keeps 16-byte alignment: N*(4*32+6) floats so far)
    gws += (4 - ((gws - ws) & 3)) & 3;
    const long gws_floats = contrastive_gemm_ws_floats(N);
    float* cws = gws + gws_floats;
    cws += ((cws - ws) & 1);                                   // doubles
    float* D = cws + ha2g_colsum_workspace_bytes(N) / 4;
    D += (4 - ((D - ws) & 3)) & 3;
    const int RB = contrastive_rows_per_block(N);
    hipLaunchKernelGGL(rownorm_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, a, an, na, N);
    hipLaunchKernelGGL(rownorm_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, b, bn, nbv, N);
    for (int r0 = 0; r0 < N; r0 += RB) {
        const int R = min(RB, N - r0);
        const float acc = r0 == 0 ? 0.f : 1.f;
        hipLaunchKernelGGL(contrastive_dist_kernel, dim3(ceil_div(R, 64), ceil_div(N, 64)), dim3(256), 0, st, an, bn, N, r0, R, D);
        hipLaunchKernelGGL(contrastive_row_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, D, an, N, r0, R, expressive, li, dan);
        HA2G_CHECK_LAUNCH("contrastive");
        // d an[r0:r0+R] = an * rowsum(C) - C bn ;  d bn (+)= - C^T an[r0:r0+R] ;  cs (+)= colsum(C)
        int rc = ha2g_gemm_f32(0, 0, R, CD, N, -1.f, D, N, bn, CD, 1.f, dan + (long)r0 * CD, CD, nullptr, 0, gws, gws_floats * 4, stream);
        if (rc) return rc;
        rc = ha2g_gemm_f32(1, 0, N, CD, R, -1.f, D, N, an + (long)r0 * CD, CD, acc, dbn, CD, nullptr, 0, gws, gws_floats * 4, stream);
        if (rc) return rc;
        rc = ha2g_colsum_f32(D, N, R, N, cs, acc, cws, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, li, (long)N, loss, 1.f / (float)N, 0);
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, an, na, dan, (const float*)nullptr, da, N);
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, bn, nbv, dbn, (const float*)cs, db, N);
    HA2G_CHECK_LAUNCH("contrastive");
    return 0;
}

}  // extern "C"

// ---- FGD evaluator statistics (model/embedding_space_evaluator.py), float64 accumulation, single-launch, deterministic ----
namespace {
__global__ __launch_bounds__(1024) void feat_stats_kernel(const float* __restrict__ f, int N, int D, double* __restrict__ sum, double* __restrict__ outer) {
    for (int e = threadIdx.x; e < D * D + D; e += 1024) {
        double s = 0.0;
        if (e < D * D) {
            const int i = e / D, j = e % D;
            for (int n = 0; n < N; ++n) s += (double)f[(long)n * D + i] * (double)f[(long)n * D + j];
            outer[e] += s;
        } else {
            const int i = e - D * D;
            for (int n = 0; n < N; ++n) s += (double)f[(long)n * D + i];
            sum[i] += s;
        }
    }
}
__global__ __launch_bounds__(1024) void l1_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, double* __restrict__ out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) s += fabs((double)a[i] - (double)b[i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] += red[0];
}
// one block per sample; per-sample values are combined in sample order by the last step (single extra block: blockIdx 0 after a second launch)
__global__ __launch_bounds__(256) void recon_metrics_partial(const float* __restrict__ r, const float* __restrict__ p, int T, int P, double* __restrict__ part) {
    __shared__ double red[3][256];
    const int b = blockIdx.x;
    const float* rb = r + (long)b * T * P;
    const float* pb = p + (long)b * T * P;
    double l1 = 0.0, dl1 = 0.0, cs = 0.0;
    for (int i = threadIdx.x; i < T * P; i += 256) {
        l1 += fabs((double)rb[i] - (double)pb[i]);
        if (i < (T - 1) * P) dl1 += fabs(((double)rb[i + P] - (double)rb[i]) - ((double)pb[i + P] - (double)pb[i]));
    }
    const int nb = P / 3;
    for (int i = threadIdx.x; i < T * nb; i += 256) {
        const float* x = rb + (long)i * 3; const float* y = pb + (long)i * 3;
        const double dot = (double)x[0] * y[0] + (double)x[1] * y[1] + (double)x[2] * y[2];
        const double nx = sqrt((double)x[0] * x[0] + (double)x[1] * x[1] + (double)x[2] * x[2]);
        const double ny = sqrt((double)y[0] * y[0] + (double)y[1] * y[1] + (double)y[2] * y[2]);
        cs += 1.0 - dot / (fmax(nx, 1e-8) * fmax(ny, 1e-8));
    }
    red[0][threadIdx.x] = l1; red[1][threadIdx.x] = dl1; red[2][threadIdx.x] = cs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; red[2][threadIdx.x] += red[2][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * b] = red[0][0] / (double)(T * P) + red[1][0] / (double)((T - 1) * P);
        part[2 * b + 1] = red[2][0];
    }
}
__global__ void recon_metrics_final(const double* __restrict__ part, int B, double* __restrict__ out) {
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += part[2 * b + threadIdx.x];
        out[threadIdx.x] += s;
    }
}
}  // namespace

extern "C" {
int ha2g_feat_stats_f64(const float* f, int N, int D, double* sum, double* outer, void* stream) {
    HA2G_REQUIRE(D >= 1 && D <= 128, "feat_stats: D=%d out of range", D);
    if (N == 0) return 0;
    hipLaunchKernelGGL(feat_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, f, N, D, sum, outer);
    HA2G_CHECK_LAUNCH("feat_stats");
    return 0;
}
int ha2g_l1_rows_f64(const float* a, const float* b, long n, double* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(l1_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out);
    HA2G_CHECK_LAUNCH("l1_rows");
    return 0;
}
/* out[0..1] += ; the first 2*B doubles after out[2] are scratch: out must hold 2 + 2*B doubles */
int ha2g_recon_metrics_f64(const float* recon, const float* poses, int B, int T, int P, double* out, void* stream) {
    HA2G_REQUIRE(P % 3 == 0 && T >= 2, "recon_metrics: P %% 3 == 0 and T >= 2");
    if (B == 0) return 0;
    hipLaunchKernelGGL(recon_metrics_partial, dim3(B), dim3(256), 0, (hipStream_t)stream, recon, poses, T, P, out + 2);
    hipLaunchKernelGGL(recon_metrics_final, dim3(1), dim3(64), 0, (hipStream_t)stream, out + 2, B, out);
    HA2G_CHECK_LAUNCH("recon_metrics");
    return 0;
}
}  // extern "C"

