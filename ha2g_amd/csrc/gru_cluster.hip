// Bidirectional GRU layer forward, "workgroup cluster" form for H = 300 on gfx950.
//
// gru.hip's kernel gives one workgroup 16 batch rows x ALL 3H gate rows: 48 workgroups at B = 384, each issuing 15 us of
// fp32 MFMA per step while streaming the whole 1.08 MB W_hh from L2.  Here a (16-row batch tile, direction) pair is served
// by a CLUSTER of G = 5 workgroups; each owns 4 of the 19 sixteen-unit tiles of the hidden state -- one tile per wave,
// one wave per SIMD -- so that
//   * its 3 x 16 x 304 slice of W_hh lives in VGPRs for all T steps (228 registers per lane; W_hh is never re-read),
//   * the per-step MFMA chain is 228 instructions (~3 us) instead of 1140,
//   * 240 of the 256 CUs work on the recurrence instead of 48.
// The price is an all-gather of the new hidden state inside the cluster every step.  It uses the placement-independent
// "data is the flag" hand-off of the CDNA4 guide (cdna_hip_programming.md G16, recipe R2): every float travels as one
// naturally aligned 8-byte {tag = step + 1, value} granule written with ONE agent-scope (sc1, write-through) store and
// read with agent-scope loads until its tag matches; no fences, no flags, correct for any workgroup->XCD placement.
// Slots are double-buffered by step parity; the buffer is zeroed by a memset node before every launch; spins are bounded
// (a timeout sets *err and lets the kernel finish with garbage rather than hang the device).
// All workgroups of a launch must be co-resident: the host wrapper caps a launch at 24 batch tiles (240 workgroups of
// 256 threads, one per CU) and loops over larger batches.
#include "common.h"

namespace {

constexpr int H = 300;
constexpr int NJT = 19;              // 16-unit tiles (304 padded units)
constexpr int HP = NJT * 16;
constexpr int LDH = HP + 4;
constexpr int TPW = 4;               // tiles (= waves) per workgroup
constexpr int G = (NJT + TPW - 1) / TPW;   // 5 workgroups per cluster
constexpr int NT = 64 * TPW;
constexpr int MAX_TILES = 24;        // 24 tiles x 2 directions x 5 = 240 workgroups <= 256 CUs
constexpr unsigned SPIN_LIMIT = 1u << 18;

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Two {value, tag} granules travel in ONE 16-byte write-through (sc1) store / agent-scope (sc1) load: each 8-byte half
// validates itself, so a torn 16-byte access cannot pair a new tag with stale data, and the fabric sees half as many
// (and 2.7x cheaper per byte) writes as with scalar 8-byte granule stores (MI355X_MICROARCH.md, hand-off price list).
__device__ __forceinline__ void store_granule_pair(__amdgpu_buffer_rsrc_t r, int byte_off, unsigned tag, float v0, float v1) {
    u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, /*aux = sc1*/ 16);
}
__device__ __forceinline__ u32x4 load_granule_pair(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, /*aux = sc1*/ 16);
}

__global__ __launch_bounds__(NT, 1) void gru_fwd_cluster_kernel(const float* __restrict__ gi,      // [B][T][2][3H]
                                                                const float* __restrict__ wp,      // packed fwd images, 2 dirs
                                                                const float* __restrict__ bhh0, const float* __restrict__ bhh1,
                                                                float* __restrict__ y,             // [B][T][2H]
                                                                float* __restrict__ rs,            // [B][T][2][4][H] or null
                                                                u64* __restrict__ xch,             // [clusters][2][16][HP] granules
                                                                int* __restrict__ err, int B, int T, int tile0, int nclusters, int dbg) {
    __shared__ __attribute__((aligned(16))) float hs[16 * LDH];
    // block -> (cluster, member): the G members of a cluster share blockIdx % 8, i.e. (observed) one XCD -- speed only
    const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
    const int q = r % G, c = (r / G) * 8 + xcd;
    if (c >= nclusters) return;
    const int dir = c & 1, b0 = (tile0 + (c >> 1)) * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const int b = b0 + lb;
    const bool bok = b < B;
    const int jt = q * TPW + wave;
    const bool tile_on = jt < NJT;
    const int j = 16 * jt + 4 * g;                       // first of this lane's 4 hidden units
    const bool jok = tile_on && j < H;
    const float* bhh = dir ? bhh1 : bhh0;

    // ---- this wave's slice of W_hh: 3 gates x 19 k-blocks, resident in registers for the whole sequence ----
    float4 wf[3 * NJT];
    {
        const float4* wsrc = reinterpret_cast<const float4*>(wp) + ((long)dir * (NJT * 3 * NJT) + (long)(tile_on ? jt : 0) * 3 * NJT) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 3 * NJT; ++f) wf[f] = wsrc[f * 64];
    }
    float4 br = make_float4(0.f, 0.f, 0.f, 0.f), bz = br, bn = br;
    if (jok) {
        br = *reinterpret_cast<const float4*>(bhh + j);
        bz = *reinterpret_cast<const float4*>(bhh + H + j);
        bn = *reinterpret_cast<const float4*>(bhh + 2 * H + j);
    }
    for (int i = tid; i < 16 * LDH; i += NT) hs[i] = 0.f;

    // own / foreign column ranges of the gathered hidden state
    const int k0 = q * TPW * 16, k1 = min(H, k0 + TPW * 16), nown = k1 - k0, nother = H - nown;
    u64* xc = xch + (long)c * 2 * 16 * HP;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xc, 0, 2 * 16 * HP * 8, 0x00020000);

    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        __syncthreads();                                  // A: hs holds the complete h_s
        float4 hb[NJT];
#pragma unroll
        for (int m = 0; m < NJT; ++m) hb[m] = *reinterpret_cast<const float4*>(&hs[lb * LDH + 16 * m + 4 * g]);
        float4 gir = make_float4(0.f, 0.f, 0.f, 0.f), giz = gir, gin = gir;
        if (jok && bok) {
            const float* gp = gi + ((long)(b * T + t) * 2 + dir) * 3 * H + j;
            gir = *reinterpret_cast<const float4*>(gp);
            giz = *reinterpret_cast<const float4*>(gp + H);
            gin = *reinterpret_cast<const float4*>(gp + 2 * H);
        }
        __syncthreads();                                  // B: every wave has its B operand; hs may be overwritten
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
        if (tile_on) {
#pragma unroll
            for (int m = 0; m < NJT; ++m) {
                const float* ph = &hb[m].x;
                const float* pr = &wf[m].x; const float* pz = &wf[NJT + m].x; const float* pn = &wf[2 * NJT + m].x;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[u], ph[u], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[u], ph[u], az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(pn[u], ph[u], an, 0, 0, 0);
                }
            }
        }
        // ---- gates (C/D layout: col = batch lane&15, row = 4*(lane>>4) + reg) ----
        float4 hn4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jok) {
            if (bok) {
                // h_prev of this lane's units: the wave's own columns of hs (only this wave ever writes them, later in this step)
                const float4 hprev = *reinterpret_cast<const float4*>(&hs[lb * LDH + j]);
                const float* hpp = &hprev.x;
                float4 r4, z4, n4, q4;
                float* pr = &r4.x; float* pz = &z4.x; float* pn = &n4.x; float* pq = &q4.x; float* ph = &hn4.x;
                const float* gr = &gir.x; const float* gz = &giz.x; const float* gn = &gin.x;
                const float* cbr = &br.x; const float* cbz = &bz.x; const float* cbn = &bn.x;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float rr = sigmoidf_(gr[u] + ar[u] + cbr[u]);
                    const float zz = sigmoidf_(gz[u] + az[u] + cbz[u]);
                    const float qq = an[u] + cbn[u];
                    const float nn = tanhf_(gn[u] + rr * qq);
                    pr[u] = rr; pz[u] = zz; pn[u] = nn; pq[u] = qq;
                    ph[u] = (1.f - zz) * nn + zz * hpp[u];
                }
                *reinterpret_cast<float4*>(y + (long)(b * T + t) * 2 * H + dir * H + j) = hn4;
                if (rs) {
                    float* rp = rs + ((long)(b * T + t) * 2 + dir) * 4 * H + j;
                    *reinterpret_cast<float4*>(rp) = r4;
                    *reinterpret_cast<float4*>(rp + H) = z4;
                    *reinterpret_cast<float4*>(rp + 2 * H) = n4;
                    *reinterpret_cast<float4*>(rp + 3 * H) = q4;
                }
            }
            *reinterpret_cast<float4*>(&hs[lb * LDH + j]) = hn4;       // own columns of h_{s+1} (zeros for padded batch rows)
            if (s + 1 < T && dbg != 2) {                               // publish: 4 floats = 2 granule pairs = 2 x 16 B
                const int go = (((s & 1) * 16 + lb) * HP + j) * 8;
                const unsigned tag = (unsigned)(s + 1);
                store_granule_pair(xr, go, tag, hn4.x, hn4.y);
                store_granule_pair(xr, go + 16, tag, hn4.z, hn4.w);
            }
        }
        // ---- gather the other members' columns of h_{s+1} ----
        if (s + 1 < T && dbg != 2) {
            const unsigned tag = (unsigned)(s + 1);
            const int sbase = (s & 1) * 16 * HP;                      // granule index of this parity's slot
            constexpr int NPMAX = (16 * (H - 32) / 2 + NT - 1) / NT;  // granule PAIRS per thread (upper bound)
            const int npair = nother / 2;                             // own / foreign ranges are multiples of 4 units
            int off[NPMAX];                                           // granule index (b * HP + k) of the pair's first float
#pragma unroll
            for (int i = 0; i < NPMAX; ++i) {
                const int p = tid + i * NT;
                if (p < 16 * npair) {
                    const int bb = p / npair, kk = (p % npair) * 2;
                    off[i] = bb * HP + (kk < k0 ? kk : kk + nown);
                } else off[i] = -1;
            }
            float v0[NPMAX], v1[NPMAX];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < NPMAX; ++i)
                    if (off[i] >= 0) {
                        const u32x4 x = load_granule_pair(xr, (sbase + off[i]) * 8);
                        v0[i] = __uint_as_float(x[0]); v1[i] = __uint_as_float(x[2]);
                        ok = ok && x[1] == tag && x[3] == tag;
                    }
                if (__all(ok) || dbg == 1) break;
                if (spins > SPIN_LIMIT) { if (lane == 0) atomicExch(err, 1); break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int i = 0; i < NPMAX; ++i)
                if (off[i] >= 0) {
                    float* d = &hs[(off[i] / HP) * LDH + (off[i] % HP)];
                    d[0] = v0[i]; d[1] = v1[i];
                }
        }
    }
}

}  // namespace

static int g_dbg = 0;
extern "C" {

void ha2g_gru_cluster_debug(int m) { g_dbg = m; }
long ha2g_gru_cluster_workspace_bytes(void) { return (long)MAX_TILES * 2 * 2 * 16 * HP * 8 + 64; }
int ha2g_gru_cluster_supported(int H_) { return H_ == H; }

// Same contract as ha2g_gru_layer_fwd (H = 300 only) plus: xch = scratch of ha2g_gru_cluster_workspace_bytes() bytes,
// err = device int32 set to 1 if a hand-off timed out (results are then invalid; the kernel still terminates).
int ha2g_gru_layer_fwd_cluster(const float* gi, const float* wp, const float* bhh_fwd, const float* bhh_rev, float* y, float* rs,
                               void* xch, int* err, int B, int T, int H_, void* stream) {
    HA2G_REQUIRE(H_ == H, "gru cluster kernel: H=%d not instantiated (300)", H_);
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || T == 0) return 0;
    const int tiles = ceil_div(B, 16);
    for (int t0 = 0; t0 < tiles; t0 += MAX_TILES) {
        const int nt = tiles - t0 < MAX_TILES ? tiles - t0 : MAX_TILES;
        const int nclusters = nt * 2;
        hipError_t e = hipMemsetAsync(xch, 0, (size_t)nclusters * 2 * 16 * HP * 8, st);
        if (e != hipSuccess) return ha2g_set_error(-2, "gru cluster: memset failed: %s", hipGetErrorString(e));
        const int grid = ceil_div(nclusters, 8) * 8 * G;
        hipLaunchKernelGGL(gru_fwd_cluster_kernel, dim3(grid), dim3(NT), 0, st, gi, wp, bhh_fwd, bhh_rev, y, rs, (u64*)xch, err, B, T,
                           t0, nclusters, g_dbg);
        HA2G_CHECK_LAUNCH("gru_layer_fwd_cluster");
    }
    return 0;
}

}  // extern "C"
