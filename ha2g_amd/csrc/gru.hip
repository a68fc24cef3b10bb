// Bidirectional GRU layer: persistent recurrent kernels (forward and BPTT) for gfx950.
//
// torch.nn.GRU semantics (reference: scripts/model/hierarchy_net.py:87-88,144 / :213,232):
//   gi = x W_ih^T + b_ih (done by ha2g_gemm_f32 for all T at once), gh = h W_hh^T + b_hh,
//   r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1-z) n + z h, h0 = 0.
//
// Decomposition: grid = (ceil(B/16), 2 directions).  A workgroup owns 16 batch rows of one direction for
// all T steps, so there is NO cross-workgroup synchronisation inside the recurrence.  Per step it computes
// gh^T[3H x 16] = W_hh[3H x H] * h^T[H x 16] with v_mfma_f32_16x16x4_f32 (exact fp32):
//   * A operand = W_hh, pre-packed once per call into MFMA fragment order (1 KiB contiguous per wave load,
//     streamed from L2 every step; 1.08 MB at H=300 stays L2-resident),
//   * B operand = h^T, 16 x H, exchanged between the 4 waves through LDS and then held in VGPRs,
//   * the three gate accumulators of a hidden unit land in the same lane, so the gate math is in-register;
//     h_prev for the blend is the lane's own B-operand register (same (batch, unit) mapping).
// K order inside a 16-wide k block is permuted (lane group g takes k = 16m + 4g + u) so both operands are
// float4 loads; a sum is order-independent up to rounding.
#include "common.h"

namespace {

constexpr int NT = 512;        // 8 waves = 2 per SIMD: one wave's W_hh loads overlap its partner's MFMAs
constexpr int NW = NT / 64;
constexpr int PF = 8;          // W_hh fragments (1 KiB per wave each) kept in flight per wave

template <int H> struct GruCfg {
    static constexpr int NJT = (H + 15) / 16;      // 16-wide tiles over hidden units (and over k)
    static constexpr int HP = NJT * 16;            // padded hidden size
    static constexpr int LDH = HP + 4;             // LDS row stride of the 16 x HP hidden-state tile
};

// ---- weight packing ---------------------------------------------------------------------------------
// fwd fragment (jt, gate, m):  lane (i = l&15, g = l>>4), u:  W_hh[gate*H + 16jt + i][16m + 4g + u]
// bwd fragment (kt, gate, jt): lane (i, g), u:               W_hh[gate*H + 16jt + 4g + u][16kt + i]
template <int H>
__global__ void gru_pack_kernel(const float* __restrict__ whh, float* __restrict__ pf, float* __restrict__ pb) {
    constexpr int NJT = GruCfg<H>::NJT;
    const int frag = blockIdx.x;                     // (a*3 + gate)*NJT + c
    const int c = frag % NJT, gate = (frag / NJT) % 3, a = frag / (3 * NJT);
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    float4 f, b;
    float* fp = &f.x; float* bp = &b.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int row = 16 * a + i, k = 16 * c + 4 * g + u;                 // fwd: a = jt, c = m
        fp[u] = (row < H && k < H) ? whh[(long)(gate * H + row) * H + k] : 0.f;
        int j = 16 * c + 4 * g + u, kk = 16 * a + i;                  // bwd: a = kt, c = jt
        bp[u] = (j < H && kk < H) ? whh[(long)(gate * H + j) * H + kk] : 0.f;
    }
    reinterpret_cast<float4*>(pf)[(long)frag * 64 + lane] = f;
    reinterpret_cast<float4*>(pb)[(long)frag * 64 + lane] = b;
}

// the same for up to 16 weight matrices in one launch (blockIdx.y = matrix): all layers and directions of a stacked GRU are packed by one launch
// at the start of its forward instead of two per layer on the chain between the layers' recurrences
struct PackBatch { const float* whh[16]; float* pf[16]; float* pb[16]; };
template <int H>
__global__ void gru_pack_multi_kernel(PackBatch pbt) {
    constexpr int NJT = GruCfg<H>::NJT;
    const float* __restrict__ whh = pbt.whh[blockIdx.y];
    float* __restrict__ pf = pbt.pf[blockIdx.y];
    float* __restrict__ pb = pbt.pb[blockIdx.y];
    const int frag = blockIdx.x;
    const int c = frag % NJT, gate = (frag / NJT) % 3, a = frag / (3 * NJT);
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    float4 f, b;
    float* fp = &f.x; float* bp = &b.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int row = 16 * a + i, k = 16 * c + 4 * g + u;
        fp[u] = (row < H && k < H) ? whh[(long)(gate * H + row) * H + k] : 0.f;
        int j = 16 * c + 4 * g + u, kk = 16 * a + i;
        bp[u] = (j < H && kk < H) ? whh[(long)(gate * H + j) * H + kk] : 0.f;
    }
    reinterpret_cast<float4*>(pf)[(long)frag * 64 + lane] = f;
    reinterpret_cast<float4*>(pb)[(long)frag * 64 + lane] = b;
}

// ---- forward ------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(NT) void gru_fwd_kernel(const float* __restrict__ gi,      // [B][T][2][3H]
                                                      const float* __restrict__ wp,      // [2][NJT*3*NJT][64][4]
                                                      const float* __restrict__ bhh0, const float* __restrict__ bhh1,
                                                      float* __restrict__ y,             // [B][T][2H]
                                                      float* __restrict__ rs,            // [B][T][2][4][H] or null
                                                      int B, int T) {
    using C = GruCfg<H>;
    constexpr int NJT = C::NJT, LDH = C::LDH;
    __shared__ __attribute__((aligned(16))) float hs[2][16 * LDH];
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const int b = b0 + lb;
    const bool bok = b < B;
    const float4* wpd = reinterpret_cast<const float4*>(wp) + (long)dir * (NJT * 3 * NJT) * 64 + lane;
    const float* bhh = dir ? bhh1 : bhh0;

    for (int i = tid; i < 16 * LDH; i += NT) hs[0][i] = 0.f;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        const int cur = s & 1;
        for (int jt = wave; jt < NJT; jt += NW) {
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
            const float4* w = wpd + (long)(jt * 3) * NJT * 64;
            const int j = 16 * jt + 4 * g;
            const bool ok = bok && j < H;
            // issue the step's gi loads early; they are consumed after the MFMA chain
            float4 gir = make_float4(0.f, 0.f, 0.f, 0.f), giz = gir, gin = gir;
            if (ok) {
                const float* gp = gi + ((long)(b * T + t) * 2 + dir) * 3 * H + j;
                gir = *reinterpret_cast<const float4*>(gp);
                giz = *reinterpret_cast<const float4*>(gp + H);
                gin = *reinterpret_cast<const float4*>(gp + 2 * H);
            }
            // k-block order of this j-tile: rotated to start at the blocks of the 4-tile group the tile belongs to,
            // m_i = (4 (jt / 4) + i) mod NJT -- the order in which the cluster kernel (gru_cluster.hip) consumes the hidden
            // state (own member's columns first, then the other members' as they arrive), so that both kernels produce
            // the same bits.  (NJT <= 4: the identity.)
            const int rot = 4 * (jt / 4);
            // W_hh fragment stream of this j-tile, order f = 3*i + gate, kept PF fragments ahead of the MFMAs
            // (one wave alone cannot hide the ~500-cycle L2 latency otherwise; hipcc keeps <=2 loads in flight).
            constexpr int NF = 3 * NJT;
            float4 ring[PF];
#pragma unroll
            for (int f = 0; f < PF && f < NF; ++f) ring[f] = w[((f % 3) * NJT + (rot + f / 3) % NJT) * 64];
            float4 hbv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float4 wv = ring[f % PF];
                if (f + PF < NF) ring[f % PF] = w[(((f + PF) % 3) * NJT + (rot + (f + PF) / 3) % NJT) * 64];
                if (f % 3 == 0) hbv = *reinterpret_cast<const float4*>(&hs[cur][lb * LDH + 16 * ((rot + f / 3) % NJT) + 4 * g]);   // B operand: h[b][16m + 4g + u]
                __builtin_amdgcn_sched_barrier(0);      // keep the refill load ahead of the MFMAs (hipcc sinks it otherwise)
                const float* pw = &wv.x; const float* ph = &hbv.x;
                if (f % 3 == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ar = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], ph[u], ar, 0, 0, 0);
                } else if (f % 3 == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) az = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], ph[u], az, 0, 0, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) an = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], ph[u], an, 0, 0, 0);
                }
            }
            // C/D layout 16x16: col (batch) = lane&15, row (unit within tile) = 4*(lane>>4) + reg
            float4 hn4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float4 br = *reinterpret_cast<const float4*>(bhh + j);
                const float4 bz = *reinterpret_cast<const float4*>(bhh + H + j);
                const float4 bn = *reinterpret_cast<const float4*>(bhh + 2 * H + j);
                const float4 hprev = *reinterpret_cast<const float4*>(&hs[cur][lb * LDH + j]);   // == hb[jt], re-read (static indexing)
                const float* hp = &hprev.x;
                float4 r4, z4, n4, q4;
                float* pr = &r4.x; float* pz = &z4.x; float* pn = &n4.x; float* pq = &q4.x; float* ph = &hn4.x;
                const float* gr = &gir.x; const float* gz = &giz.x; const float* gn = &gin.x;
                const float* cbr = &br.x; const float* cbz = &bz.x; const float* cbn = &bn.x;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float r = sigmoidf_(gr[u] + ar[u] + cbr[u]);
                    const float z = sigmoidf_(gz[u] + az[u] + cbz[u]);
                    const float q = an[u] + cbn[u];                  // W_hn h + b_hn
                    const float n = tanhf_(gn[u] + r * q);
                    pr[u] = r; pz[u] = z; pn[u] = n; pq[u] = q;
                    ph[u] = (1.f - z) * n + z * hp[u];
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
            *reinterpret_cast<float4*>(&hs[cur ^ 1][lb * LDH + j]) = hn4;    // zeros in the padding
        }
        __syncthreads();
    }
}

// ---- backward (BPTT) ----------------------------------------------------------------------------------
// Per step (reverse of the forward order):  dh = dy_t + carry;
//   dn = dh (1-z), dz = dh (h_prev - n), da_n = dn (1-n^2), da_z = dz z (1-z), da_r = da_n hn r (1-r)
//   dg[b][t][dir] = [da_r | da_z | da_n | da_n r]   (first three = d gi, [0,1,3] = d gh)
//   carry' = dh z + [da_r, da_z, da_n r] * W_hh      (MFMA, K = 3H)
template <int H>
__global__ __launch_bounds__(NT) void gru_bwd_kernel(const float* __restrict__ dy,      // [B][T][2H]
                                                      const float* __restrict__ y,       // [B][T][2H]
                                                      const float* __restrict__ rs,      // [B][T][2][4][H]
                                                      const float* __restrict__ wpt,     // [2][NJT*3*NJT][64][4]
                                                      float* __restrict__ dg,            // [B][T][2][4H]
                                                      float* __restrict__ hpo,           // [B][T][2H] h_prev per step (nullable)
                                                      int B, int T) {
    using C = GruCfg<H>;
    constexpr int NJT = C::NJT, HP = C::HP, LDH = C::LDH, LDG = 3 * HP + 4;
    __shared__ __attribute__((aligned(16))) float sg[16 * LDG];      // d gh tile  [16][3*HP]
    __shared__ __attribute__((aligned(16))) float sc[16 * LDH];      // dh carry   [16][HP]
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const float4* wpd = reinterpret_cast<const float4*>(wpt) + (long)dir * (NJT * 3 * NJT) * 64 + lane;

    for (int i = tid; i < 16 * LDH; i += NT) sc[i] = 0.f;
    for (int i = tid; i < 16 * LDG; i += NT) sg[i] = 0.f;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;
        const int tp = dir ? t + 1 : t - 1;              // time index of h_prev
        const bool has_prev = tp >= 0 && tp < T;
        // ---- phase 1: gate gradients (elementwise over 16 x H) ----
        for (int idx = tid; idx < 16 * (H / 4); idx += NT) {
            const int bb = idx / (H / 4), j = (idx % (H / 4)) * 4;
            const int b = b0 + bb;
            float4 dar = make_float4(0.f, 0.f, 0.f, 0.f), daz = dar, dghn = dar, dhz = dar;
            if (b < B) {
                const long bt = (long)b * T + t;
                const float4 dy4 = *reinterpret_cast<const float4*>(dy + bt * 2 * H + dir * H + j);
                const float* rp = rs + (bt * 2 + dir) * 4 * H + j;
                const float4 r4 = *reinterpret_cast<const float4*>(rp);
                const float4 z4 = *reinterpret_cast<const float4*>(rp + H);
                const float4 n4 = *reinterpret_cast<const float4*>(rp + 2 * H);
                const float4 q4 = *reinterpret_cast<const float4*>(rp + 3 * H);
                float4 hp4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has_prev) hp4 = *reinterpret_cast<const float4*>(y + ((long)b * T + tp) * 2 * H + dir * H + j);
                const float4 c4 = *reinterpret_cast<const float4*>(&sc[bb * LDH + j]);
                float4 dan;
                const float* pdy = &dy4.x; const float* pr = &r4.x; const float* pz = &z4.x; const float* pn = &n4.x;
                const float* pq = &q4.x; const float* php = &hp4.x; const float* pc = &c4.x;
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
                float* gp = dg + bt * 8 * H + dir * 3 * H + j;              // dg row: [d gi (r z n) fwd][d gi (r z n) rev][d gh_n fwd][d gh_n rev]
                *reinterpret_cast<float4*>(gp) = dar;
                *reinterpret_cast<float4*>(gp + H) = daz;
                *reinterpret_cast<float4*>(gp + 2 * H) = dan;
                *reinterpret_cast<float4*>(dg + bt * 8 * H + 6 * H + dir * H + j) = dghn;
                if (hpo) *reinterpret_cast<float4*>(hpo + bt * 2 * H + dir * H + j) = hp4;
            }
            *reinterpret_cast<float4*>(&sg[bb * LDG + j]) = dar;
            *reinterpret_cast<float4*>(&sg[bb * LDG + HP + j]) = daz;
            *reinterpret_cast<float4*>(&sg[bb * LDG + 2 * HP + j]) = dghn;
            *reinterpret_cast<float4*>(&sc[bb * LDH + j]) = dhz;
        }
        __syncthreads();
        // ---- phase 2: carry[b][k] += sum_{gate,j} dgh[b][gate,j] * W_hh[gate*H + j][k] ----
        if (s + 1 < T) {
            for (int kt = wave; kt < NJT; kt += NW) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
                const float4* w = wpd + (long)(kt * 3) * NJT * 64;
                constexpr int NF = 3 * NJT;       // f = 3*jt + gate
                float4 ring[PF];
#pragma unroll
                for (int f = 0; f < PF && f < NF; ++f) ring[f] = w[((f % 3) * NJT + f / 3) * 64];
                float4 dnext = *reinterpret_cast<const float4*>(&sg[lb * LDG + 4 * g]);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float4 wv = ring[f % PF];
                    const float4 dv = dnext;
                    if (f + PF < NF) ring[f % PF] = w[(((f + PF) % 3) * NJT + (f + PF) / 3) * 64];
                    if (f + 1 < NF) dnext = *reinterpret_cast<const float4*>(&sg[lb * LDG + ((f + 1) % 3) * HP + 16 * ((f + 1) / 3) + 4 * g]);
                    __builtin_amdgcn_sched_barrier(0);
                    const float* pw = &wv.x; const float* pd = &dv.x;
                    if (f % 3 == 0) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], pd[u], a0, 0, 0, 0);
                    } else if (f % 3 == 1) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], pd[u], a1, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[u], pd[u], a2, 0, 0, 0);
                    }
                }
                float* cp = &sc[lb * LDH + 16 * kt + 4 * g];
                float4 c4 = *reinterpret_cast<float4*>(cp);
                c4.x += a0[0] + a1[0] + a2[0];
                c4.y += a0[1] + a1[1] + a2[1];
                c4.z += a0[2] + a1[2] + a2[2];
                c4.w += a0[3] + a1[3] + a2[3];
                *reinterpret_cast<float4*>(cp) = c4;
            }
        }
        __syncthreads();
    }
}

// ---- small hidden sizes (H <= 64: the discriminator's GRU(8->64, 4 layers), model/hierarchy_net.py:213,232) --------------------
// The generic kernels above re-stream W_hh from L2 every step and load the step's operands from HBM when they need them: fine when a
// step carries 15 us of MFMA work (H = 300), but at H = 64 a step is 48 MFMAs per wave (0.64 us) and the launch was pure latency
// (3.3 us per step).  Here one wave owns one 16-unit tile: its 3 x NJT weight fragments stay in registers for the whole sequence
// (12 float4 at H = 64), the step's gi / gate operands are prefetched two steps ahead, barriers wait for LDS only, and the hidden
// state ping-pongs between two LDS tiles (one barrier per step).  Same MFMA order per accumulator as the generic kernel => same bits.
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // vmcnt(63) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int H>
__global__ __launch_bounds__(64 * GruCfg<H>::NJT) void gru_fwd_small_kernel(const float* __restrict__ gi, const float* __restrict__ wp,
                                                                            const float* __restrict__ bhh0, const float* __restrict__ bhh1,
                                                                            float* __restrict__ y, float* __restrict__ rs, int B, int T) {
    using C = GruCfg<H>;
    constexpr int NJT = C::NJT, LDH = C::LDH, NTS = 64 * NJT;
    __shared__ __attribute__((aligned(16))) float hs[2][16 * LDH];
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, jt = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const int b = b0 + lb;
    const int j = 16 * jt + 4 * g;
    const bool ok = b < B && j < H;
    const float* bhh = dir ? bhh1 : bhh0;
    float4 wf[3 * NJT];                                   // [gate][m]
    {
        const float4* w = reinterpret_cast<const float4*>(wp) + ((long)dir * (NJT * 3 * NJT) + (long)jt * 3 * NJT) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 3 * NJT; ++f) wf[f] = w[f * 64];
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 br = zero4, bz = zero4, bn = zero4;
    if (ok) {
        br = *reinterpret_cast<const float4*>(bhh + j);
        bz = *reinterpret_cast<const float4*>(bhh + H + j);
        bn = *reinterpret_cast<const float4*>(bhh + 2 * H + j);
    }
    for (int i = tid; i < 2 * 16 * LDH; i += NTS) (&hs[0][0])[i] = 0.f;
    // gi of the current step in registers; the next step's is requested right AFTER this step's stores: on gfx9 a wave that has loads
    // and stores in flight can only wait with vmcnt(0), so a load issued before the gates would be waited for by the gates themselves
    const long tstep = dir ? -1 : 1;
    const float* gp = gi + ((long)(b * (long)T + (dir ? T - 1 : 0)) * 2 + dir) * 3 * H + j;
    float4 c_r = zero4, c_z = zero4, c_n = zero4;
    if (ok) { c_r = *reinterpret_cast<const float4*>(gp); c_z = *reinterpret_cast<const float4*>(gp + H); c_n = *reinterpret_cast<const float4*>(gp + 2 * H); }
    float* yp = y + ((long)b * T + (dir ? T - 1 : 0)) * 2 * H + dir * H + j;
    float* rp = rs ? rs + (((long)b * T + (dir ? T - 1 : 0)) * 2 + dir) * 4 * H + j : nullptr;
    lds_only_barrier();

    for (int s = 0; s < T; ++s) {
        const int cur = s & 1;
        float4 hb[NJT];
#pragma unroll
        for (int m = 0; m < NJT; ++m) hb[m] = *reinterpret_cast<const float4*>(&hs[cur][lb * LDH + 16 * m + 4 * g]);
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
        if (s > 0) {                                      // h_0 = 0: nothing to multiply in the first step
#pragma unroll
            for (int m = 0; m < NJT; ++m) {
                const float* ph = &hb[m].x;
                const float* pr = &wf[m].x; const float* pz = &wf[NJT + m].x; const float* pn = &wf[2 * NJT + m].x;
#pragma unroll
                for (int u = 0; u < 4; ++u) ar = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[u], ph[u], ar, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) az = __builtin_amdgcn_mfma_f32_16x16x4f32(pz[u], ph[u], az, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) an = __builtin_amdgcn_mfma_f32_16x16x4f32(pn[u], ph[u], an, 0, 0, 0);
            }
        }
        float4 hn4 = zero4;
        if (ok) {
            const float* hp = &hb[0].x + 0;               // h_prev of this lane's units = hb[jt] (statically indexed below)
            float4 hprev = hb[0];
#pragma unroll
            for (int m = 1; m < NJT; ++m) if (m == jt) hprev = hb[m];
            (void)hp;
            const float* hpp = &hprev.x;
            float4 r4, z4, n4, q4;
            float* pr = &r4.x; float* pz = &z4.x; float* pn = &n4.x; float* pq = &q4.x; float* ph = &hn4.x;
            const float* gr = &c_r.x; const float* gz = &c_z.x; const float* gn = &c_n.x;
            const float* cbr = &br.x; const float* cbz = &bz.x; const float* cbn = &bn.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float r = sigmoidf_(gr[u] + ar[u] + cbr[u]);
                const float z = sigmoidf_(gz[u] + az[u] + cbz[u]);
                const float q = an[u] + cbn[u];
                const float n = tanhf_(gn[u] + r * q);
                pr[u] = r; pz[u] = z; pn[u] = n; pq[u] = q;
                ph[u] = (1.f - z) * n + z * hpp[u];
            }
            *reinterpret_cast<float4*>(yp) = hn4;
            if (rs) {
                *reinterpret_cast<float4*>(rp) = r4;
                *reinterpret_cast<float4*>(rp + H) = z4;
                *reinterpret_cast<float4*>(rp + 2 * H) = n4;
                *reinterpret_cast<float4*>(rp + 3 * H) = q4;
            }
        }
        *reinterpret_cast<float4*>(&hs[cur ^ 1][lb * LDH + j]) = hn4;    // zeros in the padding
        gp += tstep * 6 * H; yp += tstep * 2 * H;
        if (ok && s + 1 < T) { c_r = *reinterpret_cast<const float4*>(gp); c_z = *reinterpret_cast<const float4*>(gp + H); c_n = *reinterpret_cast<const float4*>(gp + 2 * H); }
        if (rs) rp += tstep * 8 * H;
        lds_only_barrier();
    }
}

// BPTT twin: thread -> (batch row, 4 units) for the gate gradients (operands prefetched one step ahead), wave kt -> k-tile of
// carry[b][k] += dgh * W_hh with its 3 x NJT transposed fragments resident; the carry lives in LDS (the two phases own it differently).
template <int H>
__global__ __launch_bounds__(64 * GruCfg<H>::NJT) void gru_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                            const float* __restrict__ rs, const float* __restrict__ wpt,
                                                                            float* __restrict__ dg, float* __restrict__ hpo, int B, int T) {
    using C = GruCfg<H>;
    constexpr int NJT = C::NJT, HP = C::HP, LDH = C::LDH, LDG = 3 * HP + 4, NTS = 64 * NJT;
    static_assert(16 * (H / 4) <= NTS, "one (row, 4-unit) item per thread");
    __shared__ __attribute__((aligned(16))) float sg[16 * LDG];
    __shared__ __attribute__((aligned(16))) float sc[16 * LDH];
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, kt = tid >> 6;
    const int lb = lane & 15, g = lane >> 4;
    const int bb = tid / (H / 4), j = (tid % (H / 4)) * 4;
    const int b = b0 + bb;
    const bool item = tid < 16 * (H / 4), own = item && b < B;
    float4 wf[3 * NJT];                                   // [gate][jt] of k-tile kt
    {
        const float4* w = reinterpret_cast<const float4*>(wpt) + ((long)dir * (NJT * 3 * NJT) + (long)(kt * 3) * NJT) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 3 * NJT; ++f) wf[f] = w[f * 64];
    }
    for (int i = tid; i < 16 * LDH; i += NTS) sc[i] = 0.f;
    for (int i = tid; i < 16 * LDG; i += NTS) sg[i] = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n_dy = zero4, n_r = zero4, n_z = zero4, n_n = zero4, n_q = zero4, n_hp = zero4;
#define HA2G_SMALL_BWD_LOAD(S)                                                                                    \
    if (own && (S) < T) {                                                                                         \
        const int t_ = dir ? (S) : T - 1 - (S);                                                                   \
        const int tp_ = dir ? t_ + 1 : t_ - 1;                                                                    \
        const long bt_ = (long)b * T + t_;                                                                        \
        n_dy = *reinterpret_cast<const float4*>(dy + bt_ * 2 * H + dir * H + j);                                  \
        const float* rp_ = rs + (bt_ * 2 + dir) * 4 * H + j;                                                      \
        n_r = *reinterpret_cast<const float4*>(rp_);                                                              \
        n_z = *reinterpret_cast<const float4*>(rp_ + H);                                                          \
        n_n = *reinterpret_cast<const float4*>(rp_ + 2 * H);                                                      \
        n_q = *reinterpret_cast<const float4*>(rp_ + 3 * H);                                                      \
        n_hp = (tp_ >= 0 && tp_ < T) ? *reinterpret_cast<const float4*>(y + ((long)b * T + tp_) * 2 * H + dir * H + j) : zero4; \
    }
    HA2G_SMALL_BWD_LOAD(0)
    lds_only_barrier();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;
        // ---- phase 1: gate gradients ----
        if (item) {
            float4 dar = zero4, daz = zero4, dghn = zero4, dhz = zero4;
            if (own) {
                const long bt = (long)b * T + t;
                const float4 c4 = *reinterpret_cast<const float4*>(&sc[bb * LDH + j]);
                float4 dan;
                const float* pdy = &n_dy.x; const float* pr = &n_r.x; const float* pz = &n_z.x; const float* pn = &n_n.x;
                const float* pq = &n_q.x; const float* php = &n_hp.x; const float* pc = &c4.x;
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
                float* gp = dg + bt * 8 * H + dir * 3 * H + j;              // dg row: [d gi (r z n) fwd][d gi (r z n) rev][d gh_n fwd][d gh_n rev]
                *reinterpret_cast<float4*>(gp) = dar;
                *reinterpret_cast<float4*>(gp + H) = daz;
                *reinterpret_cast<float4*>(gp + 2 * H) = dan;
                *reinterpret_cast<float4*>(dg + bt * 8 * H + 6 * H + dir * H + j) = dghn;
                if (hpo) *reinterpret_cast<float4*>(hpo + bt * 2 * H + dir * H + j) = n_hp;
            }
            *reinterpret_cast<float4*>(&sg[bb * LDG + j]) = dar;
            *reinterpret_cast<float4*>(&sg[bb * LDG + HP + j]) = daz;
            *reinterpret_cast<float4*>(&sg[bb * LDG + 2 * HP + j]) = dghn;
            *reinterpret_cast<float4*>(&sc[bb * LDH + j]) = dhz;
        }
        if (s + 1 == T) break;
        lds_only_barrier();
        HA2G_SMALL_BWD_LOAD(s + 1)
        // ---- phase 2: carry[b][16 kt ..] += sum_{gate, jt} dgh[b][gate, jt] * W_hh[gate*H + 16 jt ..][16 kt ..] ----
        {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
            for (int jq = 0; jq < NJT; ++jq) {
                const float4 d0 = *reinterpret_cast<const float4*>(&sg[lb * LDG + 0 * HP + 16 * jq + 4 * g]);
                const float4 d1 = *reinterpret_cast<const float4*>(&sg[lb * LDG + 1 * HP + 16 * jq + 4 * g]);
                const float4 d2 = *reinterpret_cast<const float4*>(&sg[lb * LDG + 2 * HP + 16 * jq + 4 * g]);
                const float* w0 = &wf[0 * NJT + jq].x; const float* w1 = &wf[1 * NJT + jq].x; const float* w2 = &wf[2 * NJT + jq].x;
                const float* p0 = &d0.x; const float* p1 = &d1.x; const float* p2 = &d2.x;
#pragma unroll
                for (int u = 0; u < 4; ++u) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[u], p0[u], a0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[u], p1[u], a1, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[u], p2[u], a2, 0, 0, 0);
            }
            float* cp = &sc[lb * LDH + 16 * kt + 4 * g];
            float4 c4 = *reinterpret_cast<float4*>(cp);
            c4.x += a0[0] + a1[0] + a2[0];
            c4.y += a0[1] + a1[1] + a2[1];
            c4.z += a0[2] + a1[2] + a2[2];
            c4.w += a0[3] + a1[3] + a2[3];
            *reinterpret_cast<float4*>(cp) = c4;
        }
        lds_only_barrier();
    }
#undef HA2G_SMALL_BWD_LOAD
}

template <int H>
int run_pack(const float* whh, float* pf, float* pb, hipStream_t st) {
    constexpr int NJT = GruCfg<H>::NJT;
    hipLaunchKernelGGL(gru_pack_kernel<H>, dim3(NJT * 3 * NJT), dim3(64), 0, st, whh, pf, pb);
    HA2G_CHECK_LAUNCH("gru_pack");
    return 0;
}

template <int H>
int run_pack_multi(const PackBatch& b, int n, hipStream_t st) {
    constexpr int NJT = GruCfg<H>::NJT;
    hipLaunchKernelGGL(gru_pack_multi_kernel<H>, dim3(NJT * 3 * NJT, n), dim3(64), 0, st, b);
    HA2G_CHECK_LAUNCH("gru_pack_multi");
    return 0;
}

}  // namespace

namespace {
// Bias gradients of one bidirectional layer from ONE column sum of the gate-gradient buffer dg [rows][(r z n_i) fwd | (r z n_i) rev | n_h fwd | n_h rev]:
//   d b_ih = (r, z, n_i)      d b_hh = (r, z, n_h)      (nn.GRU keeps b_ih and b_hh separate; r and z gradients coincide)
__global__ void gru_bias_grads_kernel(const float* __restrict__ cs, float* __restrict__ bih0, float* __restrict__ bhh0,
                                      float* __restrict__ bih1, float* __restrict__ bhh1, int H, float beta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * H) return;
    const int d = i / (3 * H), j = i % (3 * H);
    const float* t = cs + 3 * H * d;
    float* bi = d ? bih1 : bih0; float* bh = d ? bhh1 : bhh0;
    const float vi = t[j], vh = j < 2 * H ? t[j] : cs[6 * H + H * d + (j - 2 * H)];
    bi[j] = (beta != 0.f ? beta * bi[j] : 0.f) + vi;
    bh[j] = (beta != 0.f ? beta * bh[j] : 0.f) + vh;
}
}  // namespace

extern "C" {

// floats per direction of one packed W_hh image (forward or backward form)
long ha2g_gru_packed_floats(int H) {
    long njt = (H + 15) / 16;
    return njt * 3 * njt * 64 * 4;
}

int ha2g_gru_supported_hidden(int H) { return H == 300 || H == 64 || H == 32; }

// Pack one direction's W_hh [3H][H] into the forward (pf) and BPTT (pb) fragment images.
int ha2g_gru_pack_whh(const float* whh, float* pf, float* pb, int H, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 300: return run_pack<300>(whh, pf, pb, st);
        case 64: return run_pack<64>(whh, pf, pb, st);
        case 32: return run_pack<32>(whh, pf, pb, st);
    }
    return ha2g_set_error(-1, "gru: hidden size %d not instantiated (300, 64, 32)", H);
}

// ha2g_gru_pack_whh for n <= 16 matrices in one launch; whh / pf / pb are HOST arrays of n device pointers
int ha2g_gru_pack_whh_multi(const void* const* whh, void* const* pf, void* const* pb, int n, int H, void* stream) {
    HA2G_REQUIRE(n >= 0 && n <= 16, "gru_pack_whh_multi: %d matrices (max 16)", n);
    if (n == 0) return 0;
    PackBatch b{};
    for (int i = 0; i < n; ++i) { b.whh[i] = (const float*)whh[i]; b.pf[i] = (float*)pf[i]; b.pb[i] = (float*)pb[i]; }
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 300: return run_pack_multi<300>(b, n, st);
        case 64: return run_pack_multi<64>(b, n, st);
        case 32: return run_pack_multi<32>(b, n, st);
    }
    return ha2g_set_error(-1, "gru: hidden size %d not instantiated (300, 64, 32)", H);
}

// One bidirectional layer, all T steps.  gi [B][T][2][3H] already holds x W_ih^T + b_ih of both directions;
// wp = packed forward images of (fwd dir, reverse dir) back to back; y [B][T][2H]; rs (optional reserve for
// backward) [B][T][2][4][H] = r, z, n, (W_hn h + b_hn).
int ha2g_gru_layer_fwd(const float* gi, const float* wp, const float* bhh_fwd, const float* bhh_rev, float* y, float* rs,
                       int B, int T, int H, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || T == 0) return 0;
    dim3 grid(ceil_div(B, 16), 2), block(NT);
    switch (H) {
        case 300: hipLaunchKernelGGL(gru_fwd_kernel<300>, grid, block, 0, st, gi, wp, bhh_fwd, bhh_rev, y, rs, B, T); break;
        case 64: hipLaunchKernelGGL(gru_fwd_small_kernel<64>, grid, dim3(64 * GruCfg<64>::NJT), 0, st, gi, wp, bhh_fwd, bhh_rev, y, rs, B, T); break;
        case 32: hipLaunchKernelGGL(gru_fwd_small_kernel<32>, grid, dim3(64 * GruCfg<32>::NJT), 0, st, gi, wp, bhh_fwd, bhh_rev, y, rs, B, T); break;
        default: return ha2g_set_error(-1, "gru: hidden size %d not instantiated (300, 64, 32)", H);
    }
    HA2G_CHECK_LAUNCH("gru_layer_fwd");
    return 0;
}

// BPTT of one bidirectional layer.  dy/y [B][T][2H], rs from the forward, wpt = packed backward images of both
// directions; writes dg [B][T][2][4H] = (d gi_r, d gi_z, d gi_n, d gh_n).  Weight/bias/input gradients are
// batched GEMMs / column sums over dg done by the caller.
int ha2g_gru_layer_bwd(const float* dy, const float* y, const float* rs, const float* wpt, float* dg, float* hp, int B, int T, int H,
                       void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || T == 0) return 0;
    dim3 grid(ceil_div(B, 16), 2), block(NT);
    switch (H) {
        case 300: hipLaunchKernelGGL(gru_bwd_kernel<300>, grid, block, 0, st, dy, y, rs, wpt, dg, hp, B, T); break;
        case 64: hipLaunchKernelGGL(gru_bwd_small_kernel<64>, grid, dim3(64 * GruCfg<64>::NJT), 0, st, dy, y, rs, wpt, dg, hp, B, T); break;
        case 32: hipLaunchKernelGGL(gru_bwd_small_kernel<32>, grid, dim3(64 * GruCfg<32>::NJT), 0, st, dy, y, rs, wpt, dg, hp, B, T); break;
        default: return ha2g_set_error(-1, "gru: hidden size %d not instantiated (300, 64, 32)", H);
    }
    HA2G_CHECK_LAUNCH("gru_layer_bwd");
    return 0;
}


// colsums [8H] = column sums of dg [rows][8H]; writes (beta = 0) or accumulates (beta = 1) the four bias gradients of the layer
int ha2g_gru_bias_grads_f32(const float* colsums, float* dbih_fwd, float* dbhh_fwd, float* dbih_rev, float* dbhh_rev, int H, float beta,
                            void* stream) {
    hipLaunchKernelGGL(gru_bias_grads_kernel, dim3((6 * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, colsums, dbih_fwd, dbhh_fwd,
                       dbih_rev, dbhh_rev, H, beta);
    HA2G_CHECK_LAUNCH("gru_bias_grads");
    return 0;
}
}  // extern "C"
