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

// internal (not part of the C ABI): direct 32->32 channel 3x3 convolution, conv_c32.hip
int conv3x3_c32_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta,
                              hipStream_t st);
int conv3x3_c32_x3_launch(const float* x, const float* w, float* y, int N, int H, int W, int flip, int act, float beta,
                          hipStream_t st, const float* rsd = nullptr, const unsigned* rbits = nullptr);
int conv3x3_c32pp_serves(int H, int W);
int gemm_split_dgrad_enabled();      // gemm.hip: ha2g_gemm_set_mode bit 2 (and not the plain-bf16 mode)
extern int g_side_cus;      // conv_planes.hip: ha2g_side_cus
int conv3x3_c32_wgrad_blocks(int N, int H, int W);
int conv3x3_c32_wgrad_launch(const float* x, const float* dy, float* part, int N, int H, int W, hipStream_t st);
int conv3x3_c32_wgrad_b16_launch(const void* x, const void* dy, float* part, int N, int H, int W, hipStream_t st);
// internal: plane-based 3x3 weight gradient, conv_planes.hip (the wide split-K reduce that follows lives in gemm.hip)
int pconv_wgrad_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
long pconv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int pconv_wgrad_launch(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* part, int N, int H, int W, int Cin, int Cout,
                       hipStream_t st);
int pconv_wgrad_launch_np(const void* x, long x_ps, const void* dy, long dy_ps, int np, float* part, int N, int H, int W, int Cin, int Cout,
                          hipStream_t st);

// internal: dense product on the plane kernel (conv_planes.hip); the split-K reduce lives in gemm.hip
int plane_gemm_plan(int M, int N, int ksplit, int* mt, int* bn);
int plane_gemm_launch(const void* a, long a_ps, long lda, const void* b, long b_ps, long ldb, int M, int N, int K, float* C, long ldc, float beta,
                      const float* bias, int act, float* ws, int ksplit, int* tickets, hipStream_t st);

// ---- in-kernel split-K reduction (round 6): arrival tickets -------------------------------------------------------------------------------
// A split-K launch used to leave `splits` raw partial slabs in the workspace for a SECOND launch (splitk_reduce[_wide]_kernel) to add up: 157
// extra launches, 2.5 ms of kernel time and 1.5 GB of re-read per train step, and an HBM burst that doubled whatever main-queue kernel ran beside
// it (profiles/r05_queue_overlap.txt).  With a ticket buffer registered for the launch stream (ha2g_splitk_set_tickets) the k slices of an
// output tile instead take a ticket when their slab is written; the LAST arriver adds the slabs IN SLICE ORDER (double accumulation, exactly
// the reduce kernel's arithmetic: no float atomics, bitwise reproducible whatever the arrival order), applies the epilogue and stores the tile.
// Hand-off (MI355X guide, persistent-kernel forms): plain slab stores -> __syncthreads() (drains vmcnt) -> one lane: agent-scope release
// (buffer_wbl2 sc1) + explicit s_waitcnt vmcnt(0) -> relaxed agent atomic on the ticket; the last arriver: agent-scope acquire (buffer_inv sc1)
// -> __syncthreads() -> plain loads.  The last arriver re-zeroes its ticket: the buffer is zero between launches (graph-replay safe).
#define HA2G_SPLITK_TICKETS 16384
int* splitk_tickets_for(hipStream_t st);        // misc.hip: the buffer registered for (current device, stream), or nullptr
// every thread of the workgroup calls it after its slab stores; returns true in every thread of the last-arriving workgroup of `ticket`
__device__ __forceinline__ bool splitk_last_arriver(int* __restrict__ ticket, int nslices, int* __restrict__ sh) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == nslices - 1;
        if (last) {
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *sh = last;
    }
    __syncthreads();
    return *sh != 0;
}
// sum over z = 0 .. n-1 of p[z * stride] accumulated in double IN THAT ORDER (eight loads in flight): the arithmetic of splitk_reduce_kernel
__device__ __forceinline__ double splitk_ordered_sum(const float* __restrict__ p, long stride, int n) {
    double sd = 0.0;
    int z = 0;
    for (; z + 7 < n; z += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(long)(z + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) sd += (double)v[j];
    }
    for (; z < n; ++z) sd += (double)p[(long)z * stride];
    return sd;
}
// the same for four consecutive floats (16-byte aligned)
__device__ __forceinline__ void splitk_ordered_sum4(const float* __restrict__ p, long stride, int n, double* __restrict__ sd) {
    sd[0] = sd[1] = sd[2] = sd[3] = 0.0;
    int z = 0;
    for (; z + 3 < n; z += 4) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (long)(z + j) * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sd[0] += (double)v[j][0]; sd[1] += (double)v[j][1]; sd[2] += (double)v[j][2]; sd[3] += (double)v[j][3]; }
    }
    for (; z < n; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + (long)z * stride);
        sd[0] += (double)v[0]; sd[1] += (double)v[1]; sd[2] += (double)v[2]; sd[3] += (double)v[3];
    }
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers ----
// The two-piece bf16 split of the split-bf16 inner product (gemm.hip, conv_planes.hip): hi = bf16(x) round-to-nearest-even (v_cvt_pk_bf16_f32),
// lo = bf16(x - hi); two values per call, packed {b | a} like the MFMA operand words.  Producers that write pre-split planes and consumers
// that split at staging time must use THIS function: the bit-identity of the plane-based kernels rests on it.
typedef __bf16 ha2g_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float ha2g_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_bf16(float a, float b, unsigned& hi, unsigned& lo) {
    ha2g_f32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, ha2g_bf16x2_t));
    ha2g_f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ha2g_bf16x2_t));
}

// The three-piece split (x = p0 + p1 + p2, each the bf16 rounding of what is left): three bf16 pieces hold all 24 mantissa bits of an fp32
// value, and the six products p0 q0 + (p0 q1 + p1 q0) + (p0 q2 + p1 q1 + p2 q0) reproduce the fp32 product to 2^-24 -- the arithmetic class
// of the fp32 MFMA chain (tests/test_gpu_kernels.py::test_three_piece_split_core_is_fp32_accurate).  p0 / p1 are split2_bf16's hi / lo.
__device__ __forceinline__ void split3_bf16(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    ha2g_f32x2_t v = {a, b};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, ha2g_bf16x2_t));
    ha2g_f32x2_t r = {a - __uint_as_float(p0 << 16), b - __uint_as_float(p0 & 0xffff0000u)};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ha2g_bf16x2_t));
    ha2g_f32x2_t q = {r[0] - __uint_as_float(p1 << 16), r[1] - __uint_as_float(p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, ha2g_bf16x2_t));
}
// NP pieces of two values -> out[0..NP-1] (packed {b | a} words)
template <int NP> __device__ __forceinline__ void splitn_bf16(float a, float b, unsigned* out) {
    if constexpr (NP == 3) split3_bf16(a, b, out[0], out[1], out[2]);
    else if constexpr (NP == 2) split2_bf16(a, b, out[0], out[1]);
    else { unsigned l; split2_bf16(a, b, out[0], l); }
}
// NP pre-split planes of one tensor: piece q lives at base + q * ps (elements).  The two-plane entry points of round 3 (hi, lo) are the case
// ps = lo - hi, np = 2.
struct PlaneSet { const unsigned short* p; long ps; };
__host__ __device__ __forceinline__ const unsigned short* plane_of(const PlaneSet& s, int q) { return s.p + (long)q * s.ps; }

// number of bf16 pieces of the split backward products (ha2g_gemm_set_mode bit 6): 2 = 16-bit operand mantissa, 3 = all 24 bits (fp32-class)
int gemm_bwd_pieces();

// element access of the activation tensors: fp32, or bf16 (bf16-storage mode, BASELINE config 5: the tensors live in HBM as bf16 -- round to
// nearest even on every store -- while statistics, accumulators and the arithmetic of every pass stay fp32 / double exactly as in the fp32 mode)
typedef unsigned short b16;
__device__ __forceinline__ float4 ld4(const float* p, long i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ void st4(float* p, long i4, const float4& v) { reinterpret_cast<float4*>(p)[i4] = v; }
__device__ __forceinline__ float4 ld4(const b16* p, long i4) {
    const uint2 w = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(b16* p, long i4, const float4& v) {
    uint2 h; unsigned l;
    split2_bf16(v.x, v.y, h.x, l); split2_bf16(v.z, v.w, h.y, l);
    reinterpret_cast<uint2*>(p)[i4] = h;
}
__device__ __forceinline__ float ld1(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float ld1(const b16* p, long i) { return __uint_as_float((unsigned)p[i] << 16); }

// Gate non-linearities on the hardware exp2/rcp units (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute error
// ~2e-7 on values in (0,1) / (-1,1), far inside the 1e-4 parity budget, and ~10x fewer VALU slots than libm's
// expf/tanhf in the recurrent kernels' serial epilogue.  Saturate cleanly: exp -> inf gives rcp -> 0.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

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

// Reduce-scatter over the 16 lanes of a row (lanes sharing lane >> 4): every lane brings NV (16 or 8) doubles; on return v[0] of lane r (= lane & 15)
// is the sum over the row's 16 lanes of value index r & (NV - 1) (NV = 8: both halves of the row hold the same eight sums).  A fixed exchange tree
// (partners lane ^ 8, ^ 4, ^ 2, ^ 1; a + b is commutative, so both partners compute the same bits): deterministic.  15 (NV = 16) / 15 (NV = 8)
// double exchanges instead of NV x 4 for a plain butterfly.
template <int NV> __device__ __forceinline__ void row16_reduce_scatter(double (&v)[NV], int lane) {
    static_assert(NV == 16 || NV == 8, "row16_reduce_scatter: 8 or 16 values");
    if constexpr (NV == 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], 8, 64);
    }
    constexpr int TOP = NV == 16 ? 8 : 4;
#pragma unroll
    for (int m = TOP, n = NV / 2; m >= 1; m >>= 1, n >>= 1) {
        const bool up = (lane & m) != 0;                    // lanes with the bit set keep the upper half of the values
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const double send = up ? v[k] : v[k + n], keep = up ? v[k + n] : v[k];
            v[k] = keep + __shfl_xor(send, m, 64);
        }
    }
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

// Compute units of the current device (256 on an unpartitioned MI355X; fewer under CPX/DPX partitioning), queried ONCE per device and translation
// unit: the tile / split heuristics run per launch and hipDeviceGetAttribute is a driver call (~100 queries per train step before).
static inline int hw_cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        n[dev] = c;
    }
    return n[dev];
}
