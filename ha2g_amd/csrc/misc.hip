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

extern "C" int ha2g_abi_version(void) { return 6; }      // 2: guarded Adam, ha2g_sparse_adam2_f32, N-piece plane entry points (*_np); 3: ha2g_gru_cluster_tile_cap; 4: BatchNorm statistics from the forward convolution's epilogue; 5: ha2g_splitk_set_tickets (in-kernel split-K reduction); 6: ha2g_se_bn_bwd_* (SE + bn2 backward in two passes)

// ---- split-K arrival tickets per (device, stream): see common.h ----
#include <mutex>
#include <map>
static std::mutex g_tk_mu;
static std::map<std::pair<int, void*>, int*> g_tk;
static int g_tk_on = 0;       // OFF by default: measured slower than the reduce launch on this path (profiles/r06_splitk_inkernel.txt)
int* splitk_tickets_for(hipStream_t st) {
    if (!g_tk_on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_tk_mu);
    auto it = g_tk.find({dev, (void*)st});
    return it == g_tk.end() ? nullptr : it->second;
}
// tickets: HA2G_SPLITK_TICKETS (16384) ints of device memory, ZEROED by the caller, owned by the caller, used by every split-K launch on `stream`
// of the current device from now on (nullptr: unregister -> those launches use the separate reduce launch again)
extern "C" int ha2g_splitk_set_tickets(void* tickets, long n_ints, void* stream) {
    HA2G_REQUIRE(tickets == nullptr || n_ints >= HA2G_SPLITK_TICKETS, "splitk_set_tickets: %ld ticket words (< %d)", n_ints, HA2G_SPLITK_TICKETS);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ha2g_set_error(-2, "splitk_set_tickets: no current device");
    std::lock_guard<std::mutex> lk(g_tk_mu);
    if (tickets) g_tk[{dev, stream}] = (int*)tickets; else g_tk.erase({dev, stream});
    return 0;
}
extern "C" int ha2g_splitk_ticket_words(void) { return HA2G_SPLITK_TICKETS; }
extern "C" void ha2g_splitk_in_kernel(int on) { g_tk_on = on; }      // 1 = add the slabs in the kernel where a ticket buffer is registered; 0 (default) = the reduce launch

namespace {

// out[c] = beta*out[c] + sum_r X[r*ld + c].  Two levels, both fixed-order (deterministic): grid (col chunks of 64,
// row chunks); each block's 4 waves stride its row chunk with double accumulators, combine through LDS and write one
// partial per (row chunk, column); a second kernel adds the row-chunk partials per column in ascending order.
constexpr int CS_MAXCHUNK = 256;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long ld, long rows, int cols,
                                                             double* __restrict__ part) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long rbeg = (long)blockIdx.y * per, rend = min(rows, rbeg + per);
    double s0 = 0.0, s1 = 0.0;
    if (c < cols) {
        long r = rbeg + w;
        for (; r + 4 < rend; r += 8) { s0 += X[r * ld + c]; s1 += X[(r + 4) * ld + c]; }
        for (; r < rend; r += 4) s0 += X[r * ld + c];
    }
    sh[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0 && c < cols) part[(long)blockIdx.y * cols + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
// narrow matrices (cols <= 1024, cols / 4 divides 256, 16-byte aligned rows): 16-byte loads, 256 / (cols / 4) rows per trip and four trips in flight --
// the 64-lane form above keeps cols of 64 lanes busy with 4-byte loads (the audio taps' bias gradients, 16 / 32 / 64 columns x 0.3-1.2 M rows:
// 48-71 us each on the main queue of the tower's backward).  Same partial layout, fixed order.
__global__ __launch_bounds__(256) void colsum_partial_v4_kernel(const float* __restrict__ X, long ld, long rows, int cols, double* __restrict__ part) {
    __shared__ double sh[256][4];
    const int CV = cols >> 2, cv = threadIdx.x % CV, r0 = threadIdx.x / CV, rstep = 256 / CV;
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long rbeg = (long)blockIdx.x * per, rend = min(rows, rbeg + per);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    long r = rbeg + r0;
    for (; r + 3L * rstep < rend; r += 4L * rstep) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(X + (r + (long)j * rstep) * ld + cv * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[0] += (double)v[j][0]; a[1] += (double)v[j][1]; a[2] += (double)v[j][2]; a[3] += (double)v[j][3]; }
    }
    for (; r < rend; r += rstep) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ld + cv * 4);
        a[0] += (double)v[0]; a[1] += (double)v[1]; a[2] += (double)v[2]; a[3] += (double)v[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = a[k];
    __syncthreads();
    if (threadIdx.x < CV) {
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t = threadIdx.x; t < 256; t += CV) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += sh[t][k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) part[(long)blockIdx.x * cols + threadIdx.x * 4 + k] = s4[k];
    }
}
// one wave per column: lanes stride the row-chunk partials, fixed xor tree
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, int nchunk, int cols, float* __restrict__ out,
                                                           float beta) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= cols) return;
    double s = 0.0;
    for (int k = lane; k < nchunk; k += 64) s += part[(long)k * cols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[c] = (beta != 0.f ? beta * out[c] : 0.f) + (float)s;
}

}  // namespace

// ws: >= CS_MAXCHUNK*cols doubles of scratch (ha2g_colsum_workspace_bytes)
extern "C" long ha2g_colsum_workspace_bytes(int cols) { return (long)CS_MAXCHUNK * cols * 8; }
extern "C" int ha2g_colsum_f32(const float* X, long ld, long rows, int cols, float* out, float beta, float* ws, void* stream) {
    if (cols <= 0) return 0;
    HA2G_REQUIRE(ws != nullptr, "colsum: workspace required");
    hipStream_t st = (hipStream_t)stream;
    if (cols % 4 == 0 && cols <= 1024 && 256 % (cols / 4) == 0 && ld % 4 == 0 && ((uintptr_t)X & 15) == 0 && rows >= 4096) {
        long want = rows / 256;
        const int nchunk = (int)(want > CS_MAXCHUNK ? CS_MAXCHUNK : want);
        hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(nchunk), dim3(256), 0, st, X, ld, rows, cols, (double*)ws);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 4)), dim3(256), 0, st, (const double*)ws, nchunk, cols, out, beta);
        HA2G_CHECK_LAUNCH("colsum");
        return 0;
    }
    int cchunks = ceil_div(cols, 64);
    long want = rows / 64;                                // >= 64 rows per block
    int nchunk = (int)(want < 1 ? 1 : (want > CS_MAXCHUNK ? CS_MAXCHUNK : want));
    while (nchunk > 1 && (long)nchunk * cchunks > 2048) nchunk /= 2;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cchunks, nchunk), dim3(256), 0, st, X, ld, rows, cols, (double*)ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 4)), dim3(256), 0, st, (const double*)ws, nchunk, cols, out, beta);
    HA2G_CHECK_LAUNCH("colsum");
    return 0;
}


// ---- test aid: hold compute units for a while (tests/test_gpu_ddp2.py: the cluster GRU under foreign co-resident work) ----
namespace {
// One 256-thread workgroup that owns its CU's whole LDS (no other LDS-using workgroup fits beside it) and spins on the 100 MHz wall clock.
__global__ __launch_bounds__(256) void occupy_kernel(long ticks, int* __restrict__ sink) {
    __shared__ int hog[(160 * 1024 - 64) / 4];
    hog[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const long t0 = wall_clock64();
    int v = 0;
    while (wall_clock64() - t0 < ticks) { v += hog[(threadIdx.x + v) & 255]; __builtin_amdgcn_s_sleep(8); }
    if (v == 0x7fffffff) *sink = v;                    // keeps the LDS reads alive
}
}  // namespace

/* `blocks` workgroups, each monopolising one compute unit's LDS for `microseconds` (a stand-in for a foreign kernel, e.g. an RCCL
 * collective, resident while a cluster GRU launch needs its 240 workgroups co-resident).  sink: any device int. */
extern "C" int ha2g_debug_occupy(int blocks, long microseconds, int* sink, void* stream) {
    HA2G_REQUIRE(blocks >= 1 && blocks <= 256 && microseconds >= 0, "debug_occupy: 1..256 blocks");
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, microseconds * 100, sink);
    HA2G_CHECK_LAUNCH("debug_occupy");
    return 0;
}
