// On-GPU log-mel front-end: replaces the offline librosa step of the reference (scripts/utils/data_utils.py:34-38,
// dataset_script/script/make_ted_dataset.py:121-123):
//     melspectrogram(y, sr=16000, n_fft=1024, hop_length=512, power=2) -> power_to_db(ref=max over the clip) -> float16
// The STFT is a GEMM: rows = frames of the centre-padded clip (row t starts at sample 512 t of the padded signal, i.e. the
// A operand is the padded signal itself with lda = 512 -- overlapping rows, nothing is materialised), columns = the
// Hann-windowed DFT basis [cos | sin] (1026 x 1024, built once).  |X|^2 and the mel projection (128 x 513 Slaney filters,
// K padded to 516) follow as an elementwise kernel and a second GEMM; a per-clip max and one last kernel produce dB values
// laid out [clip][mel][frame] like the reference's arrays.  Both GEMMs run on the exact fp32 MFMA path (ha2g_gemm_f32).
#include "common.h"
#include "../../include/ha2g_hip.h"

namespace {

constexpr int NFFT = 1024, HOP = 512, NBIN = 513, NBINP = 516, NMEL = 128;
constexpr double PI = 3.14159265358979323846;

// basis[k][n] = hann[n] * cos(2 pi k n / 1024)  (k < 513),  basis[513 + k][n] = -hann[n] * sin(2 pi k n / 1024)
__global__ void dft_basis_kernel(float* __restrict__ basis) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * NBIN * NFFT) return;
    const int row = i / NFFT, n = i % NFFT;
    const int k = row < NBIN ? row : row - NBIN;
    const double w = 0.5 - 0.5 * cos(2.0 * PI * n / NFFT);
    const int ph = (int)(((long)k * n) % NFFT);                       // exact phase reduction
    const double a = 2.0 * PI * ph / NFFT;
    basis[i] = (float)(row < NBIN ? w * cos(a) : -w * sin(a));
}

__device__ double hz_to_mel_d(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_hz / f_sp + log(f / min_log_hz) / logstep : f / f_sp;
}
__device__ double mel_to_hz_d(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = log(6.4) / 27.0, min_log_mel = min_log_hz / f_sp;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}
// fb[m][k], k < 516 (columns 513..515 zero): Slaney-normalised triangles, htk = False, fmin = 0, fmax = sr / 2
__global__ void mel_fb_kernel(float* __restrict__ fb, int sr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NMEL * NBINP) return;
    const int m = i / NBINP, k = i % NBINP;
    float v = 0.f;
    if (k < NBIN) {
        const double mmax = hz_to_mel_d(sr / 2.0);
        const double f0 = mel_to_hz_d(mmax * m / (NMEL + 1)), f1 = mel_to_hz_d(mmax * (m + 1) / (NMEL + 1)),
                     f2 = mel_to_hz_d(mmax * (m + 2) / (NMEL + 1));
        const double f = (sr / 2.0) * k / (NBIN - 1);
        const double lower = (f - f0) / (f1 - f0), upper = (f2 - f) / (f2 - f1);
        const double t = fmin(lower, upper);
        v = (float)(fmax(0.0, t) * (2.0 / (f2 - f0)));
    }
    fb[i] = v;
}

// ypad[b][j] = y[b][reflect(j - 512)] (or 0 outside with pad_reflect = 0), j < S; S = clip stride (multiple of 512)
__global__ void pad_kernel(const float* __restrict__ y, long n, long S, int reflect, float* __restrict__ ypad, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / S, j = i % S;
    long s = j - NFFT / 2;
    float v = 0.f;
    if (j < n + NFFT) {
        if (s < 0) { if (reflect) { s = -s; v = s < n ? y[b * n + s] : 0.f; } }
        else if (s >= n) { if (reflect) { s = 2 * (n - 1) - s; v = s >= 0 ? y[b * n + s] : 0.f; } }
        else v = y[b * n + s];
    }
    ypad[i] = v;
}

// P[r][k] = re^2 + im^2 for k < 513, 0 for the 3 padding columns
__global__ void power_kernel(const float* __restrict__ X, long rows, float* __restrict__ P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * NBINP) return;
    const long r = i / NBINP; const int k = (int)(i % NBINP);
    float v = 0.f;
    if (k < NBIN) { const float re = X[r * (2 * NBIN) + k], im = X[r * (2 * NBIN) + NBIN + k]; v = re * re + im * im; }
    P[i] = v;
}

// per clip: max over its T valid frames x 128 mels (fixed-order tree => deterministic)
__global__ __launch_bounds__(256) void clip_max_kernel(const float* __restrict__ M, int RPC, int T, float* __restrict__ mx) {
    __shared__ float sh[256];
    const float* base = M + (long)blockIdx.x * RPC * NMEL;
    float m = 0.f;                                                     // powers are >= 0
    for (int i = threadIdx.x; i < T * NMEL; i += 256) m = fmaxf(m, base[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) mx[blockIdx.x] = sh[0];
}

// out[b][m][t] = max(10 log10(max(S, amin)) - 10 log10(max(amin, ref_b)), -top_db)   (the maximum of the left side is 0)
__global__ void to_db_kernel(const float* __restrict__ M, const float* __restrict__ mx, int RPC, int T, long B,
                             int round_f16, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * NMEL * T) return;
    const int t = (int)(i % T); const int m = (int)((i / T) % NMEL); const long b = i / ((long)T * NMEL);
    const float amin = 1e-10f;
    const float s = M[((long)b * RPC + t) * NMEL + m];
    float v = 10.f * log10f(fmaxf(amin, s)) - 10.f * log10f(fmaxf(amin, mx[b]));
    v = fmaxf(v, -80.f);
    if (round_f16) v = (float)(_Float16)v;                             // the reference stores float16
    out[i] = v;
}

inline long clip_stride(long n) { return (n + NFFT + HOP - 1) / HOP * HOP; }

}  // namespace

extern "C" {

long ha2g_logmel_tables_floats(void) { return (long)2 * NBIN * NFFT + (long)NMEL * NBINP; }
int ha2g_logmel_frames(long n_samples) { return (int)(1 + n_samples / HOP); }

// tables: ha2g_logmel_tables_floats() floats, filled once (windowed DFT basis, then the mel filters)
int ha2g_logmel_init_f32(float* tables, int sr, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dft_basis_kernel, dim3(ceil_div(2 * NBIN * NFFT, 256)), dim3(256), 0, st, tables);
    hipLaunchKernelGGL(mel_fb_kernel, dim3(ceil_div(NMEL * NBINP, 256)), dim3(256), 0, st, tables + (long)2 * NBIN * NFFT, sr);
    HA2G_CHECK_LAUNCH("logmel_init");
    return 0;
}

long ha2g_logmel_workspace_floats(int B, long n) {
    const long S = clip_stride(n), R = (long)B * (S / HOP);
    return (long)B * S + NFFT + R * (2 * NBIN) + R * NBINP + R * NMEL + B + 4096 + (64L << 20) / 4;
}

// y [B][n] fp32 clips -> out [B][128][T], T = 1 + n / 512, dB relative to each clip's maximum, floored at -80.
int ha2g_logmel_f32(const float* y, int B, long n, int pad_reflect, const float* tables, int round_f16, float* out, float* ws,
                    void* stream) {
    HA2G_REQUIRE(B > 0 && n >= 1, "logmel: empty input");
    HA2G_REQUIRE(!pad_reflect || n > NFFT / 2, "logmel: reflect padding needs more than %d samples", NFFT / 2);
    hipStream_t st = (hipStream_t)stream;
    const long S = clip_stride(n);
    const int RPC = (int)(S / HOP), T = ha2g_logmel_frames(n);
    const long R = (long)B * RPC;
    float* ypad = ws;                                   // [B][S] + one frame of slack (the last clip's junk rows read past S)
    float* X = ypad + (long)B * S + NFFT;               // [R][1026]  re | im
    float* P = X + R * (2 * NBIN);                      // [R][516]
    float* M = P + R * NBINP;                           // [R][128]
    float* mx = M + R * NMEL;                           // [B]
    float* gws = mx + ((B + 3) / 4) * 4 + 4;
    gws += (4 - ((gws - ws) & 3)) & 3;
    const long total = (long)B * S;
    hipLaunchKernelGGL(pad_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, y, n, S, pad_reflect, ypad, total);
    if (hipMemsetAsync(ypad + total, 0, NFFT * sizeof(float), st) != hipSuccess) return ha2g_set_error(-2, "logmel: memset failed");
    HA2G_CHECK_LAUNCH("logmel_pad");
    const float* basis = tables; const float* fb = tables + (long)2 * NBIN * NFFT;
    int rc = ha2g_gemm_f32(0, 1, (int)R, 2 * NBIN, NFFT, 1.f, ypad, HOP, basis, NFFT, 0.f, X, 2 * NBIN, nullptr, 0, gws, 64L << 20, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(power_kernel, dim3(ceil_div(R * NBINP, 256)), dim3(256), 0, st, X, R, P);
    rc = ha2g_gemm_f32(0, 1, (int)R, NMEL, NBINP, 1.f, P, NBINP, fb, NBINP, 0.f, M, NMEL, nullptr, 0, gws, 64L << 20, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(clip_max_kernel, dim3(B), dim3(256), 0, st, M, RPC, T, mx);
    hipLaunchKernelGGL(to_db_kernel, dim3(ceil_div((long)B * NMEL * T, 256)), dim3(256), 0, st, M, mx, RPC, T, (long)B, round_f16, out);
    HA2G_CHECK_LAUNCH("logmel");
    return 0;
}

}  // extern "C"
