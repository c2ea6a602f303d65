// mfcc.hip -- feature front-end: the step right before the hot path.
//
// Replaces analyze_mfccs (src/sound.rs:215-242): 1024-sample Hanning windows hopped by 256
// (src/lib.rs:24-25, src/sound.rs:228-229), per window a 12-coefficient MFCC between 100 Hz and
// 8 kHz (src/sound.rs:218), all frames of a sound back to back (frame-major, the layout
// Sound::mfccs() hands to the matcher, src/sound.rs:189-193).
//
// PARITY UNPINNED: the reference's arithmetic lives in two un-vendored crates (`vox_box` at git
// HEAD for the MFCC, `sample` 0.9.1 for the Windower) and no reference test holds a number for it.
// This is therefore a self-consistent extractor with its definition written down here (and
// restated on the CPU by the test oracle, which the parity tests compare against):
//   frames   T = (n - 1024) / 256 + 1 full windows (0 when n < 1024); with SSYM_MFCC_PAD_TAIL
//            T = n / 256 windows, samples past the end read as 0
//   window   w[i] = 0.5 - 0.5 cos(2 pi i / 1024)
//   spectrum X = FFT_1024(w x) (radix-2 decimation in time, f64), P[k] = re^2 + im^2, k = 0..512
//   mel      NF = 2 n_coeffs + 2 triangular filters, equally spaced on mel(f) = 1127 ln(1 + f/700)
//            between f_lo and min(f_hi, rate/2), evaluated at the bin centres k rate / 1024
//   E[m]     = sum_k W[m][k] P[k] (k ascending);  L[m] = ln(max(E[m], 1e-30))
//   c[j]     = sum_m L[m] cos(pi j (m + 1/2) / NF), j = 1..n_coeffs (m ascending; c0 is not kept)
// Window, twiddles, filter weights and the DCT matrix are tabulated on the host in f64 and uploaded,
// so device and oracle share every constant; FFT, spectrum and filter sums use the same operation
// order on both sides (bit-identical), the natural log is each side's libm.
//
// Mapping: one frame per 256-thread workgroup, the 1024-point FFT in LDS (16 KB), two butterflies
// per thread per stage; filter m and coefficient j are each one thread's sequential sum.
#include "ssym_internal.hpp"

#include <cmath>
#include <vector>

namespace ssym {

constexpr int kBin = SSYM_MFCC_BIN, kHop = SSYM_MFCC_HOP, kSpec = kBin / 2 + 1;
constexpr int kMaxFilters = 130;      // n_coeffs <= 64

struct MfccTables {
    const double *win;      // [1024]
    const double *twRe;     // [512]  cos(-2 pi k / 1024)
    const double *twIm;     // [512]  sin(-2 pi k / 1024)
    const double *weights;  // [nf][513]
    const int *lo, *hi;     // [nf] first / one-past-last bin with a non-zero weight
    const double *dct;      // [n_coeffs][nf]
};

__device__ __forceinline__ uint32_t bitrev10(uint32_t i) { return __brev(i) >> 22; }

__global__ __launch_bounds__(256) void mfcc_kernel(const double *__restrict__ samples, uint64_t nSamples,
                                                   uint64_t nFrames, MfccTables tb, int nf, int nCoeffs,
                                                   double *__restrict__ out)
{
    __shared__ double re[kBin], im[kBin];
    __shared__ double logE[kMaxFilters];
    const int tid = threadIdx.x;
    for (uint64_t t = blockIdx.x; t < nFrames; t += gridDim.x) {
        const uint64_t base = t * kHop;
#pragma unroll
        for (int q = 0; q < kBin / 256; ++q) {
            const int i = tid + 256 * q;
            const uint64_t g = base + i;
            const double v = g < nSamples ? samples[g] : 0.0;
            const uint32_t r = bitrev10((uint32_t)i);
            re[r] = __dmul_rn(v, tb.win[i]);
            im[r] = 0.0;
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 1; s <= 10; ++s) {
            const int half = 1 << (s - 1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int b = tid + 256 * q;
                const int j = b & (half - 1);
                const int i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + half;
                const int k = j << (10 - s);
                const double wr = tb.twRe[k], wi = tb.twIm[k];
                const double xr = re[i1], xi = im[i1];
                const double tr = __dsub_rn(__dmul_rn(wr, xr), __dmul_rn(wi, xi));
                const double ti = __dadd_rn(__dmul_rn(wr, xi), __dmul_rn(wi, xr));
                const double ar = re[i0], ai = im[i0];
                re[i1] = __dsub_rn(ar, tr);
                im[i1] = __dsub_rn(ai, ti);
                re[i0] = __dadd_rn(ar, tr);
                im[i0] = __dadd_rn(ai, ti);
            }
            __syncthreads();
        }
        // power spectrum into re[0..512] (each thread reads and writes its own bins only)
        for (int k = tid; k < kSpec; k += 256) {
            const double a = re[k], b = im[k];
            re[k] = __dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b));
        }
        __syncthreads();
        if (tid < nf) {
            const double *w = tb.weights + (size_t)tid * kSpec;
            double e = 0.0;
            for (int k = tb.lo[tid]; k < tb.hi[tid]; ++k)
                e = __dadd_rn(e, __dmul_rn(w[k], re[k]));
            logE[tid] = log(fmax(e, 1e-30));
        }
        __syncthreads();
        if (tid < nCoeffs) {
            const double *d = tb.dct + (size_t)tid * nf;
            double c = 0.0;
            for (int m = 0; m < nf; ++m)
                c = __dadd_rn(c, __dmul_rn(logE[m], d[m]));
            out[t * nCoeffs + tid] = c;
        }
        __syncthreads();   // re / im / logE are reused by the next frame
    }
}

static double mel_of(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }
static double hz_of(double m) { return 700.0 * (std::exp(m / 1127.0) - 1.0); }

}  // namespace ssym

using namespace ssym;

extern "C" {

int32_t ssym_mfcc_num_frames(uint64_t n_samples, uint32_t flags, uint64_t *out_frames)
{
    if (!out_frames)
        return SSYM_E_INVALID;
    if (flags & SSYM_MFCC_PAD_TAIL)
        *out_frames = n_samples / kHop;
    else
        *out_frames = n_samples >= (uint64_t)kBin ? (n_samples - kBin) / kHop + 1 : 0;
    return SSYM_OK;
}

int32_t ssym_mfcc(ssym_ctx *ctx, const double *samples, uint64_t n_samples, double sample_rate,
                  uint32_t n_coeffs, double f_lo, double f_hi, uint32_t flags, double *out_mfccs,
                  double *out_mean)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (n_coeffs == 0 || n_coeffs > 64 || !(sample_rate > 0.0) || !(f_lo >= 0.0) || !(f_hi > f_lo)) {
        ctx->err = "ssym_mfcc: need 1 <= n_coeffs <= 64, sample_rate > 0, 0 <= f_lo < f_hi";
        return SSYM_E_INVALID;
    }
    uint64_t T = 0;
    ssym_mfcc_num_frames(n_samples, flags, &T);
    if (out_mean)
        for (uint32_t j = 0; j < n_coeffs; ++j)
            out_mean[j] = 0.0;
    if (T == 0)
        return SSYM_OK;
    if (!samples || !out_mfccs) {
        ctx->err = "ssym_mfcc: NULL buffer";
        return SSYM_E_INVALID;
    }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool outDev = (flags & SSYM_OUT_DEVICE) != 0;

    // tables (host f64; the oracle tabulates the same expressions)
    const int nf = 2 * (int)n_coeffs + 2;
    const double PI = 3.14159265358979323846;
    std::vector<double> tab;
    const size_t oWin = 0, oTwRe = oWin + kBin, oTwIm = oTwRe + kBin / 2, oW = oTwIm + kBin / 2,
                 oDct = oW + (size_t)nf * kSpec, nTab = oDct + (size_t)n_coeffs * nf;
    tab.assign(nTab, 0.0);
    for (int i = 0; i < kBin; ++i)
        tab[oWin + i] = 0.5 - 0.5 * std::cos(2.0 * PI * (double)i / (double)kBin);
    for (int k = 0; k < kBin / 2; ++k) {
        tab[oTwRe + k] = std::cos(-2.0 * PI * (double)k / (double)kBin);
        tab[oTwIm + k] = std::sin(-2.0 * PI * (double)k / (double)kBin);
    }
    std::vector<int> range(2 * (size_t)nf, 0);
    {
        const double top = std::min(f_hi, 0.5 * sample_rate);
        const double m0 = mel_of(f_lo), m1 = mel_of(top);
        for (int m = 0; m < nf; ++m) {
            const double h0 = hz_of(m0 + (m1 - m0) * (double)m / (double)(nf + 1));
            const double h1 = hz_of(m0 + (m1 - m0) * (double)(m + 1) / (double)(nf + 1));
            const double h2 = hz_of(m0 + (m1 - m0) * (double)(m + 2) / (double)(nf + 1));
            int lo = kSpec, hi = 0;
            for (int k = 0; k < kSpec; ++k) {
                const double f = (double)k * sample_rate / (double)kBin;
                double w = 0.0;
                if (f > h0 && f <= h1)
                    w = (f - h0) / (h1 - h0);
                else if (f > h1 && f < h2)
                    w = (h2 - f) / (h2 - h1);
                tab[oW + (size_t)m * kSpec + k] = w;
                if (w != 0.0) {
                    lo = std::min(lo, k);
                    hi = std::max(hi, k + 1);
                }
            }
            range[m] = lo < hi ? lo : 0;
            range[nf + m] = lo < hi ? hi : 0;
        }
    }
    for (uint32_t j = 0; j < n_coeffs; ++j)
        for (int m = 0; m < nf; ++m)
            tab[oDct + (size_t)j * nf + m] = std::cos(PI * (double)(j + 1) * ((double)m + 0.5) / (double)nf);

    double *dTab = nullptr, *dSmp = nullptr, *dOut = nullptr;
    int *dRange = nullptr;
    int32_t rc = dev_alloc(ctx, (void **)&dTab, nTab * sizeof(double));
    if (rc == SSYM_OK)
        rc = dev_alloc(ctx, (void **)&dRange, range.size() * sizeof(int));
    if (rc == SSYM_OK)
        rc = dev_alloc(ctx, (void **)&dSmp, n_samples * sizeof(double));
    if (rc == SSYM_OK && !outDev)
        rc = dev_alloc(ctx, (void **)&dOut, T * n_coeffs * sizeof(double));
    hipError_t e = hipSuccess;
    if (rc == SSYM_OK) {
        if (outDev)
            dOut = out_mfccs;
        e = hipMemcpyAsync(dTab, tab.data(), nTab * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dRange, range.data(), range.size() * sizeof(int), hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dSmp, samples, n_samples * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            MfccTables tb{dTab + oWin, dTab + oTwRe, dTab + oTwIm, dTab + oW, dRange, dRange + nf, dTab + oDct};
            const unsigned grid = (unsigned)std::min<uint64_t>(T, (uint64_t)ctx->num_cus * 16);
            mfcc_kernel<<<grid, 256, 0, st>>>(dSmp, n_samples, T, tb, nf, (int)n_coeffs, dOut);
            e = hipGetLastError();
        }
        std::vector<double> hostOut;
        double *res = out_mfccs;
        if (e == hipSuccess && outDev && out_mean) {
            hostOut.resize(T * n_coeffs);
            res = hostOut.data();
        }
        if (e == hipSuccess && (!outDev || out_mean))
            e = hipMemcpyAsync(res, dOut, T * n_coeffs * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e == hipSuccess && out_mean) {
            // analyze_mean_mfccs (src/sound.rs:271-286): per-coefficient sum over frames, then / T
            for (uint64_t t = 0; t < T; ++t)
                for (uint32_t j = 0; j < n_coeffs; ++j)
                    out_mean[j] += res[t * n_coeffs + j];
            for (uint32_t j = 0; j < n_coeffs; ++j)
                out_mean[j] = out_mean[j] / (double)T;
        }
    }
    dev_free(ctx, dTab);
    dev_free(ctx, dRange);
    dev_free(ctx, dSmp);
    if (!outDev)
        dev_free(ctx, dOut);
    if (rc != SSYM_OK)
        return rc;
    if (e != hipSuccess) {
        ctx->err = std::string("ssym_mfcc: ") + hipGetErrorString(e);
        return SSYM_E_HIP;
    }
    return SSYM_OK;
    });
}

}  // extern "C"
