// Sparse embedding-gradient path: the four n_words x 300 embedding tables (model/hierarchy_net.py:31-34) see at most
// B*34 + 1 distinct rows per step, yet a dense gradient / dense Adam streams all 4 x 24 MB through zeroing, all-reduce and the
// 28 B/parameter optimizer pass.  Here the gradient is a compact (row ids, row sums) list and Adam touches only those rows --
// with LAZY CATCH-UP so that the result is bit-identical to dense torch.optim.Adam: a row that receives no gradient still moves under
// dense Adam (m <- b1 m, v <- b2 v, p -= step_t m / (sqrt(v) rs2_t + eps)); those updates are replayed, in order and with the per-step
// scalars the dense kernel used (kept in a device table), the next time the row is read (embedding forward) or updated.
#include "common.h"
#include <limits.h>

namespace {

// Compact a token batch: uniq[0] = 0 (the padding id keeps slot 0 whether present or not), then the other distinct ids in order of first
// occurrence; remap[p] = slot of tok[p]; count = number of slots.  map: int32 [n_rows], all INT_MAX on entry and on exit.  Single block.
__global__ __launch_bounds__(1024) void unique_tokens_kernel(const long* __restrict__ tok, int n, int* __restrict__ map, int* __restrict__ cpos,
                                                             long* __restrict__ uniq, long* __restrict__ remap, int* __restrict__ count) {
    __shared__ int scan[1024];
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += 1024) atomicMin(&map[tok[p]], p);
    __threadfence_block();
    __syncthreads();
    const int per = (n + 1023) / 1024, p0 = tid * per, p1 = min(n, p0 + per);
    int c = 0;
    for (int p = p0; p < p1; ++p) c += (tok[p] != 0 && map[tok[p]] == p);
    scan[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                       // inclusive Hillis-Steele scan
        int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int base = 1 + scan[tid] - c;                              // slots 1.. in position order
    for (int p = p0; p < p1; ++p)
        if (tok[p] != 0 && map[tok[p]] == p) { cpos[p] = base; uniq[base] = tok[p]; ++base; }
    if (tid == 0) { uniq[0] = 0; *count = 1 + scan[1023]; }
    __threadfence_block();
    __syncthreads();
    for (int p = tid; p < n; p += 1024) remap[p] = tok[p] == 0 ? 0 : cpos[map[tok[p]]];
    __syncthreads();
    for (int p = tid; p < n; p += 1024) map[tok[p]] = INT_MAX;
}

// table[t] = {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} in float, from doubles -- the scalar prologue of adam_kernel (optim.hip)
__global__ void adam_scalars_kernel(const int* __restrict__ step, double lr, double b1, double b2, float2* __restrict__ table, int cap) {
    const int t = *step;
    if (t < 1 || t >= cap) return;
    const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
    table[t] = make_float2((float)(lr / bc1), (float)(1.0 / sqrt(bc2)));
}

// One block per listed row.  Replays the zero-gradient updates of steps last[row]+1 .. upto-1 (upto = *step if vals, *step + 1 otherwise),
// then -- with vals -- the real update of step *step.  ids must be distinct.
__global__ __launch_bounds__(128) void sparse_adam_kernel(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int* __restrict__ last,
                                                          const long* __restrict__ ids, const int* __restrict__ count, const float* __restrict__ vals,
                                                          const float2* __restrict__ table, const int* __restrict__ step, int C, float b1, float b2,
                                                          float omb1, float omb2, float eps, int cap, double lr_d, double b1_d, double b2_d,
                                                          const int* __restrict__ guard) {
    const int r = blockIdx.x;
    if (r >= *count) return;
    if (vals != nullptr && guard != nullptr && *guard != 0) return;      // flagged step: no real update (catch-ups replay VALID earlier steps and still run)
    const long row = ids[r];
    const int t = *step, t0 = last[row];
    const int zero_to = vals ? t - 1 : t;                      // last step replayed with a zero gradient
    const bool fresh = t0 == 0;                                // never updated: m = v = 0, zero-gradient steps are exact no-ops
    // per-step scalars: the device table up to `cap` steps (adam_scalars_kernel stops writing there: a captured graph can replay past any
    // host-side count), beyond it the same double-precision prologue recomputed in place -- same values, never an out-of-bounds read
    auto scalars = [&](int s) -> float2 {
        if (s < cap) return table[s];
        const double bc1 = 1.0 - pow(b1_d, (double)s), bc2 = 1.0 - pow(b2_d, (double)s);
        return make_float2((float)(lr_d / bc1), (float)(1.0 / sqrt(bc2)));
    };
    for (int c = threadIdx.x; c < C; c += 128) {
        const long o = row * C + c;
        float p = W[o], m = M[o], v = V[o];
        if (!fresh)
            for (int s = t0 + 1; s <= zero_to; ++s) {
                const float2 sc = scalars(s);
                m = b1 * m; v = b2 * v;
                p -= sc.x * m / (sqrtf(v) * sc.y + eps);
            }
        if (vals) {
            const float2 sc = scalars(t);
            const float g = vals[(long)r * C + c];
            m = b1 * m + omb1 * g; v = b2 * v + omb2 * g * g;
            p -= sc.x * m / (sqrtf(v) * sc.y + eps);
        }
        W[o] = p; M[o] = m; V[o] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) last[row] = vals ? t : (fresh ? 0 : t);
}
__global__ void iota_kernel(long* __restrict__ ids, int* __restrict__ count, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) ids[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
}

}  // namespace

extern "C" {

int ha2g_unique_tokens(const long* tok, int n, int* map, int* cpos, long* uniq, long* remap, int* count, void* stream) {
    HA2G_REQUIRE(n >= 1, "unique_tokens: empty batch");
    hipLaunchKernelGGL(unique_tokens_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tok, n, map, cpos, uniq, remap, count);
    HA2G_CHECK_LAUNCH("unique_tokens");
    return 0;
}
int ha2g_adam_scalars(const int* step, double lr, double b1, double b2, void* table, int cap, void* stream) {
    hipLaunchKernelGGL(adam_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, lr, b1, b2, (float2*)table, cap);
    HA2G_CHECK_LAUNCH("adam_scalars");
    return 0;
}
// ABI 2: the round-3 entry point ha2g_sparse_adam_f32 had grown two parameters under an unchanged name; this is the same operation under a NEW
// name (plus the optional guard word of ha2g_adam_guarded_f32), the old symbol keeps its round-3 signature and forwards here.  Past
// `table_steps` the per-step scalars of replayed steps are recomputed with the CURRENT lr: exact only while lr is constant (the reference never
// changes it, scripts/train.py:155-170).
int ha2g_sparse_adam2_f32(float* W, float* M, float* V, int* last, const long* ids, const int* count, int max_rows, const float* vals,
                          const void* table, const int* step, int C, double b1, double b2, double eps, int table_steps, double lr, const int* guard,
                          void* stream) {
    if (max_rows <= 0) return 0;
    HA2G_REQUIRE(table_steps >= 1, "sparse_adam: empty scalar table");
    hipLaunchKernelGGL(sparse_adam_kernel, dim3(max_rows), dim3(128), 0, (hipStream_t)stream, W, M, V, last, ids, count, vals,
                       (const float2*)table, step, C, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, table_steps, lr, b1, b2,
                       guard);
    HA2G_CHECK_LAUNCH("sparse_adam");
    return 0;
}
int ha2g_sparse_adam_f32(float* W, float* M, float* V, int* last, const long* ids, const int* count, int max_rows, const float* vals,
                         const void* table, const int* step, int C, double b1, double b2, double eps, int table_steps, double lr, void* stream) {
    return ha2g_sparse_adam2_f32(W, M, V, last, ids, count, max_rows, vals, table, step, C, b1, b2, eps, table_steps, lr, nullptr, stream);
}
int ha2g_iota_ids(long* ids, int* count, int n, void* stream) {
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, count, n);
    HA2G_CHECK_LAUNCH("iota_ids");
    return 0;
}

}  // extern "C"
