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


// ---- backward (BPTT), cluster form -------------------------------------------------------------------------------------
// Member q owns hidden units [64q, 64q+64).  Per step (reverse of the forward order):
//   phase 1  gate gradients of the OWN units (one (row, 4-unit) group per thread): dg -> HBM, d gh -> LDS, dh*z kept in regs;
//   phase 2  partial[b][k] = sum_{gate, j in own} dgh[b][gate,j] * W_hh[gate*H+j][k] for ALL k on MFMA, with the member's
//            slice of W_hh (the same 192 rows as in the forward) resident in registers as 60 transposed fragments per wave;
//   exchange the 64-column block of `partial` that belongs to member p is sent to p (granule pairs, 16-byte sc1 stores);
//            every member adds the five blocks of its own columns in member order (fixed => deterministic):
//            carry' = dh*z + sum_src partial_src.   The carry never leaves the owning thread's registers.
constexpr int LDG = 3 * 64 + 4;      // LDS row stride of the own d gh tile [16][3][64]
constexpr int LDP = 64 + 4;

__global__ __launch_bounds__(NT, 1) void gru_bwd_cluster_kernel(const float* __restrict__ dy,      // [B][T][2H]
                                                                const float* __restrict__ y,       // [B][T][2H]
                                                                const float* __restrict__ rs,      // [B][T][2][4][H]
                                                                const float* __restrict__ wpt,     // packed bwd images, 2 dirs
                                                                float* __restrict__ dg,            // [B][T][2][4H]
                                                                u64* __restrict__ xch, int* __restrict__ err, int B, int T,
                                                                int tile0, int nclusters, int dbg) {
    __shared__ __attribute__((aligned(16))) float sg[16 * LDG];
    __shared__ __attribute__((aligned(16))) float sp[16 * LDP];
    const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
    const int q = r % G, c = (r / G) * 8 + xcd;
    if (c >= nclusters) return;
    const int dir = c & 1, b0 = (tile0 + (c >> 1)) * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const int k0 = q * 64;                                  // first own unit
    // phase-1 / gather ownership: thread -> (batch row bb, units k0 + jl4 .. +3)
    const int bb = tid >> 4, jl4 = (tid & 15) * 4;
    const int jo = k0 + jl4, bo = b0 + bb;
    const bool own_ok = jo < H && bo < B;

    // ---- resident transposed W_hh fragments: wave w serves k-tiles w, w+4, ... ; own j-tiles jt0..jt0+3 ----
    constexpr int NKW = (NJT + TPW - 1) / TPW;             // k-tiles per wave (5)
    float4 wf[NKW * 3 * TPW];
    {
        const float4* base = reinterpret_cast<const float4*>(wpt) + (long)dir * (NJT * 3 * NJT) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < NKW; ++kk)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int jl = 0; jl < TPW; ++jl) {
                    const int kt = wave + kk * TPW, jt = q * TPW + jl;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kt < NJT && jt < NJT) v = base[((long)(kt * 3 + gate) * NJT + jt) * 64];
                    wf[(kk * 3 + gate) * TPW + jl] = v;
                }
    }
    for (int i = tid; i < 16 * LDG; i += NT) sg[i] = 0.f;
    for (int i = tid; i < 16 * LDP; i += NT) sp[i] = 0.f;
    u64* xc = xch + (long)c * 2 * G * G * 16 * 64;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xc, 0, 2 * G * G * 16 * 64 * 8, 0x00020000);
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;
        const int tp = dir ? t + 1 : t - 1;
        const bool has_prev = tp >= 0 && tp < T;
        // ---- phase 1 ----
        float4 dar = make_float4(0.f, 0.f, 0.f, 0.f), daz = dar, dghn = dar, dhz = dar;
        if (own_ok) {
            const long bt = (long)bo * T + t;
            const float4 dy4 = *reinterpret_cast<const float4*>(dy + bt * 2 * H + dir * H + jo);
            const float* rp = rs + (bt * 2 + dir) * 4 * H + jo;
            const float4 r4 = *reinterpret_cast<const float4*>(rp);
            const float4 z4 = *reinterpret_cast<const float4*>(rp + H);
            const float4 n4 = *reinterpret_cast<const float4*>(rp + 2 * H);
            const float4 q4 = *reinterpret_cast<const float4*>(rp + 3 * H);
            float4 hp4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_prev) hp4 = *reinterpret_cast<const float4*>(y + ((long)bo * T + tp) * 2 * H + dir * H + jo);
            float4 dan;
            const float* pdy = &dy4.x; const float* pr = &r4.x; const float* pz = &z4.x; const float* pn = &n4.x;
            const float* pq = &q4.x; const float* php = &hp4.x; const float* pc = &carry.x;
            float* o_r = &dar.x; float* o_z = &daz.x; float* o_n = &dan.x; float* o_q = &dghn.x; float* o_c = &dhz.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dh = pdy[u] + pc[u];
                const float dn = dh * (1.f - pz[u]);
                const float dz = dh * (php[u] - pn[u]);
                const float a_n = dn * (1.f - pn[u] * pn[u]);
                o_n[u] = a_n;
                o_z[u] = dz * pz[u] * (1.f - pz[u]);
                o_r[u] = a_n * pq[u] * pr[u] * (1.f - pr[u]);
                o_q[u] = a_n * pr[u];
                o_c[u] = dh * pz[u];
            }
            float* gp = dg + (bt * 2 + dir) * 4 * H + jo;
            *reinterpret_cast<float4*>(gp) = dar;
            *reinterpret_cast<float4*>(gp + H) = daz;
            *reinterpret_cast<float4*>(gp + 2 * H) = dan;
            *reinterpret_cast<float4*>(gp + 3 * H) = dghn;
        }
        if (s + 1 == T) break;                              // the carry out of the last step is never used (h0 is constant)
        *reinterpret_cast<float4*>(&sg[bb * LDG + jl4]) = dar;
        *reinterpret_cast<float4*>(&sg[bb * LDG + 64 + jl4]) = daz;
        *reinterpret_cast<float4*>(&sg[bb * LDG + 128 + jl4]) = dghn;
        __syncthreads();
        // ---- phase 2: partial sums for all k from the own units ----
        float4 bop[3 * TPW];
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int jl = 0; jl < TPW; ++jl) bop[gate * TPW + jl] = *reinterpret_cast<const float4*>(&sg[lb * LDG + gate * 64 + 16 * jl + 4 * g]);
        const unsigned tag = (unsigned)(s + 1);
#pragma unroll
        for (int kk = 0; kk < NKW; ++kk) {
            const int kt = wave + kk * TPW;
            if (kt >= NJT) continue;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
            for (int jl = 0; jl < TPW; ++jl) {
                const float* w0 = &wf[(kk * 3 + 0) * TPW + jl].x; const float* w1 = &wf[(kk * 3 + 1) * TPW + jl].x;
                const float* w2 = &wf[(kk * 3 + 2) * TPW + jl].x;
                const float* d0 = &bop[jl].x; const float* d1 = &bop[TPW + jl].x; const float* d2 = &bop[2 * TPW + jl].x;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[u], d0[u], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[u], d1[u], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[u], d2[u], a2, 0, 0, 0);
                }
            }
            const float p0 = a0[0] + a1[0] + a2[0], p1 = a0[1] + a1[1] + a2[1], p2 = a0[2] + a1[2] + a2[2], p3 = a0[3] + a1[3] + a2[3];
            const int dst = kt / TPW, kl = 16 * (kt % TPW) + 4 * g;      // owner of these columns, column within its block
            if (dst == q) {
                *reinterpret_cast<float4*>(&sp[lb * LDP + kl]) = make_float4(p0, p1, p2, p3);
            } else if (dbg != 2) {
                const int go = (((((s & 1) * G + dst) * G + q) * 16 + lb) * 64 + kl) * 8;
                store_granule_pair(xr, go, tag, p0, p1);
                store_granule_pair(xr, go + 16, tag, p2, p3);
            }
        }
        __syncthreads();
        // ---- gather: carry' = dh*z + sum over members (ascending) of their partial for my 4 units ----
        float4 part[G];                                      // statically indexed only (member q's slot stays unused)
        const float4 own_part = *reinterpret_cast<const float4*>(&sp[bb * LDP + jl4]);
        if (dbg != 2) {
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int src = 0; src < G; ++src) {
                    // columns >= H (the padding of the last member's block) are never published: do not wait for them
                    if (src == q || jo >= H) { part[src] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
                    const int go = (((((s & 1) * G + q) * G + src) * 16 + bb) * 64 + jl4) * 8;
                    const u32x4 x0 = load_granule_pair(xr, go), x1 = load_granule_pair(xr, go + 16);
                    part[src] = make_float4(__uint_as_float(x0[0]), __uint_as_float(x0[2]), __uint_as_float(x1[0]), __uint_as_float(x1[2]));
                    ok = ok && x0[1] == tag && x0[3] == tag && x1[1] == tag && x1[3] == tag;
                }
                if (__all(ok) || dbg == 1) break;
                if (spins > SPIN_LIMIT) { if (lane == 0) atomicExch(err, 1); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        } else {
#pragma unroll
            for (int src = 0; src < G; ++src) part[src] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        carry = dhz;
#pragma unroll
        for (int src = 0; src < G; ++src) {
            const float4 p = (src == q) ? own_part : part[src];
            carry.x += p.x; carry.y += p.y; carry.z += p.z; carry.w += p.w;
        }
        if (!own_ok) carry = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace

static int g_dbg = 0;
extern "C" {

void ha2g_gru_cluster_debug(int m) { g_dbg = m; }
long ha2g_gru_cluster_workspace_bytes(void) { return (long)MAX_TILES * 2 * 2 * G * G * 16 * 64 * 8 + 64; }   /* sized for the backward's per-pair slots */
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

// BPTT counterpart (same contract as ha2g_gru_layer_bwd, H = 300 only); wpt = packed backward images of both directions.
int ha2g_gru_layer_bwd_cluster(const float* dy, const float* y, const float* rs, const float* wpt, float* dg, void* xch, int* err,
                               int B, int T, int H_, void* stream) {
    HA2G_REQUIRE(H_ == H, "gru cluster kernel: H=%d not instantiated (300)", H_);
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || T == 0) return 0;
    const int tiles = ceil_div(B, 16);
    for (int t0 = 0; t0 < tiles; t0 += MAX_TILES) {
        const int nt = tiles - t0 < MAX_TILES ? tiles - t0 : MAX_TILES;
        const int nclusters = nt * 2;
        hipError_t e = hipMemsetAsync(xch, 0, (size_t)nclusters * 2 * G * G * 16 * 64 * 8, st);
        if (e != hipSuccess) return ha2g_set_error(-2, "gru cluster: memset failed: %s", hipGetErrorString(e));
        const int grid = ceil_div(nclusters, 8) * 8 * G;
        hipLaunchKernelGGL(gru_bwd_cluster_kernel, dim3(grid), dim3(NT), 0, st, dy, y, rs, wpt, dg, (u64*)xch, err, B, T, t0, nclusters, g_dbg);
        HA2G_CHECK_LAUNCH("gru_layer_bwd_cluster");
    }
    return 0;
}

}  // extern "C"
