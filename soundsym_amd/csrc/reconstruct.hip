// reconstruct.hip -- the step right after the matching path.
//
// Replaces, for a whole batch, the per-target tail of SoundSequence::clone_from_dictionary
// (src/sound.rs:456-465: zero-pad the matched sound's samples up to the target's sample count, or
// truncate them down to it) and SoundSequence::to_sound's concatenation (src/sound.rs:475-480),
// plus Sound::write_file's sample conversion (src/sound.rs:139).  Pure gather: bit-exact by
// construction; the dictionary's samples stay resident so only indices go in and samples come out.
#include "ssym_internal.hpp"

#include <new>

struct ssym_samples {
    double *samples = nullptr;
    uint64_t *off = nullptr;
    uint32_t n = 0;
    uint64_t total = 0;
};

namespace ssym {

// one workgroup per (target, 4096-sample chunk)
__global__ __launch_bounds__(256) void reconstruct_kernel(const double *__restrict__ src,
                                                          const uint64_t *__restrict__ srcOff,
                                                          const uint32_t *__restrict__ idx,
                                                          const uint64_t *__restrict__ outOff, uint32_t nSounds,
                                                          double *__restrict__ out, int32_t *__restrict__ pcm)
{
    const uint32_t t = blockIdx.y;
    const uint64_t o0 = outOff[t], n = outOff[t + 1] - o0;
    const uint32_t s = idx[t];
    uint64_t sBase = 0, sLen = 0;
    if (s < nSounds) {
        sBase = srcOff[s];
        sLen = srcOff[s + 1] - sBase;
    }
    // 16 samples per thread, four loads in flight at a time (a plain one-load-per-iteration loop
    // reached 4.4 TB/s; the copy is bound by how many bytes each wave keeps outstanding)
    const uint64_t k0 = (uint64_t)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
        double v[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t k = k0 + (uint64_t)(u + w) * 256;
            v[w] = (k < n && k < sLen) ? src[sBase + k] : 0.0;     // :457-462
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t k = k0 + (uint64_t)(u + w) * 256;
            if (k >= n)
                continue;
            if (out)
                out[o0 + k] = v[w];
            if (pcm) {
                // (i32::max_value() as f64 * sample) as i32: truncate toward zero, saturate, NaN -> 0
                const double x = __dmul_rn(2147483647.0, v[w]);
                int32_t q;
                if (x != x) q = 0;
                else if (x >= 2147483647.0) q = 2147483647;
                else if (x <= -2147483648.0) q = (int32_t)0x80000000;
                else q = (int32_t)x;
                pcm[o0 + k] = q;
            }
        }
    }
}

}  // namespace ssym

using namespace ssym;

extern "C" {

int32_t ssym_samples_create(ssym_ctx *ctx, const double *samples, const uint64_t *sample_offsets,
                            uint32_t n_sounds, ssym_samples **out)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx || !out)
        return SSYM_E_INVALID;
    *out = nullptr;
    if (!sample_offsets || sample_offsets[0] != 0) {
        ctx->err = "sample_offsets must be non-NULL and start at 0";
        return SSYM_E_INVALID;
    }
    for (uint32_t i = 0; i < n_sounds; ++i)
        if (sample_offsets[i + 1] < sample_offsets[i]) {
            ctx->err = "sample_offsets must be non-decreasing";
            return SSYM_E_INVALID;
        }
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ssym_samples *h = new (std::nothrow) ssym_samples();
    if (!h)
        return SSYM_E_NOMEM;
    h->n = n_sounds;
    h->total = sample_offsets[n_sounds];
    if (h->total && !samples) {
        delete h;
        ctx->err = "samples is NULL";
        return SSYM_E_INVALID;
    }
    hipError_t e = hipMalloc((void **)&h->samples, (h->total ? h->total : 1) * sizeof(double));
    if (e == hipSuccess)
        e = hipMalloc((void **)&h->off, (n_sounds + 1) * sizeof(uint64_t));
    if (e == hipSuccess && h->total)
        e = hipMemcpyAsync(h->samples, samples, h->total * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(h->off, sample_offsets, (n_sounds + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                           ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->err = std::string("ssym_samples_create: ") + hipGetErrorString(e);
        if (h->samples) (void)hipFree(h->samples);
        if (h->off) (void)hipFree(h->off);
        delete h;
        return SSYM_E_HIP;
    }
    *out = h;
    return SSYM_OK;
    });
}

int32_t ssym_samples_destroy(ssym_ctx *ctx, ssym_samples *s)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!s)
        return SSYM_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (s->samples) (void)hipFree(s->samples);
    if (s->off) (void)hipFree(s->off);
    delete s;
    return SSYM_OK;
    });
}

int32_t ssym_reconstruct(ssym_ctx *ctx, const ssym_samples *s, const uint32_t *idx, const uint64_t *out_offsets,
                         uint32_t n_targets, double *out_samples, int32_t *out_pcm32)
{
    return guarded(ctx, [&]() -> int32_t {
    if (!ctx)
        return SSYM_E_INVALID;
    if (!s || !idx || !out_offsets || out_offsets[0] != 0) {
        ctx->err = "ssym_reconstruct: bad arguments";
        return SSYM_E_INVALID;
    }
    if (s->n == 0) {
        ctx->err = "empty dictionary";
        return SSYM_E_EMPTY_DICT;
    }
    uint64_t maxLen = 0;
    for (uint32_t t = 0; t < n_targets; ++t) {
        if (out_offsets[t + 1] < out_offsets[t] || idx[t] >= s->n) {
            ctx->err = "ssym_reconstruct: offsets not monotonic or index out of range";
            return SSYM_E_INVALID;
        }
        maxLen = std::max<uint64_t>(maxLen, out_offsets[t + 1] - out_offsets[t]);
    }
    const uint64_t total = out_offsets[n_targets];
    if (n_targets == 0 || total == 0 || (!out_samples && !out_pcm32))
        return SSYM_OK;
    SSYM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // staging: idx + offsets, then outputs
    const size_t inBytes = sizeof(uint32_t) * n_targets + sizeof(uint64_t) * (n_targets + 1) + 16;
    int32_t rc = ensure(ctx, ctx->best, inBytes);
    if (rc != SSYM_OK)
        return rc;
    uint64_t *dOff = (uint64_t *)ctx->best.ptr;
    uint32_t *dIdx = (uint32_t *)(dOff + n_targets + 1);
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(dOff, out_offsets, sizeof(uint64_t) * (n_targets + 1), hipMemcpyHostToDevice, st));
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(dIdx, idx, sizeof(uint32_t) * n_targets, hipMemcpyHostToDevice, st));
    rc = ensure(ctx, ctx->part, total * (sizeof(double) + sizeof(int32_t)));
    if (rc != SSYM_OK)
        return rc;
    double *dOut = out_samples ? (double *)ctx->part.ptr : nullptr;
    int32_t *dPcm = out_pcm32 ? (int32_t *)((double *)ctx->part.ptr + total) : nullptr;
    dim3 grid((unsigned)((maxLen + 4095) / 4096), n_targets);
    SSYM_HIP_CHECK(ctx, hipEventRecord(ctx->ev[0], st));
    reconstruct_kernel<<<grid, 256, 0, st>>>(s->samples, s->off, dIdx, dOff, s->n, dOut, dPcm);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    SSYM_HIP_CHECK(ctx, hipEventRecord(ctx->ev[1], st));
    if (out_samples)
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync(out_samples, dOut, total * sizeof(double), hipMemcpyDeviceToHost, st));
    if (out_pcm32)
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync(out_pcm32, dPcm, total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    ssym_timings tm{};
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess)
        tm.main_ms = tm.total_ms = ms;       // the gather kernel alone (ssym_get_timings)
    tm.main_launches = 1;
    ctx->timings = tm;
    return SSYM_OK;
    });
}

}  // extern "C"
