// pack.hip -- turns the caller's array-of-segments (Sound::mfccs() per segment, frame-major,
// reference layout src/sound.rs:189-193, 330-343) into the device-resident forms the kernels read:
//   raw   f64 [total_frames][dim]        exact copy / exact widening of the input
//   norm  f64 [n]                        refcos: norm(me) per segment, src/sound.rs:35-38 order
//   rec   f32 [n_pad][frames_pad][2*KSP] dtw filter: MFMA operand records (ssym_internal.hpp)
//   len, max_sqnorm                      per segment
#include "ssym_internal.hpp"

#include <algorithm>
#include <cstring>

namespace ssym {

int32_t ensure(ssym_ctx *ctx, DeviceBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.ptr)
        return SSYM_OK;
    if (b.ptr && ctx->stream_only) {
        // inside a sharded step the stream may hold collectives that only complete when every rank has arrived: waiting
        // for it here would be a wait without the step's deadline.  The old block is released after the step's one
        // (deadline-guarded) synchronisation instead (release_deferred, called by comm.hip).
        ctx->deferred_free.push_back(b.ptr);
        b.ptr = nullptr;
        b.bytes = 0;
    }
    if (b.ptr) {
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        SSYM_HIP_CHECK(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.bytes = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    SSYM_HIP_CHECK(ctx, hipMalloc(&b.ptr, want));
    b.bytes = want;
    return SSYM_OK;
}

// Small device blocks on the matching path are zeroed by a kernel of this library: hipMemsetAsync runs as a runtime
// blit kernel that costs 5-10 us of idle stream on either side of it (profiles/r03_share512_1gpu.md), four of them per
// filter-path step.
__global__ void zero_words_kernel(uint32_t *p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = 0u;
}

int32_t zero_words(ssym_ctx *ctx, void *p, size_t bytes)
{
    const size_t n = (bytes + 3) / 4;
    if (n == 0)
        return SSYM_OK;
    zero_words_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((uint32_t *)p, n);
    SSYM_HIP_CHECK(ctx, hipGetLastError());
    return SSYM_OK;
}

void release_deferred(ssym_ctx *ctx)
{
    for (void *p : ctx->deferred_free)
        (void)hipFree(p);
    ctx->deferred_free.clear();
}

__global__ void widen_f32_kernel(const float *__restrict__ in, double *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        out[i] = (double)in[i];
}

// refcos: norm(me) = fold(0, |memo, item| item*item + memo), src/sound.rs:35-38: strictly sequential, product and sum
// rounded separately (the library is compiled with -ffp-contract=off; __dmul_rn/__dadd_rn make it explicit).
// One WAVE per segment: 64 consecutive values per load (coalesced), every lane squares its own, and the chain of sums takes
// the squares in order out of the lanes (v_readlane: every lane carries the same memo).  The next 64 values are
// requested before the chain over the current ones starts.  (One THREAD per segment with a load per trip waited out a
// memory round trip per value: 0.19 ms for 4096 segments of 1536 values -- as much as a whole search of that dictionary.)
__device__ __forceinline__ double lane_value(double v, int j)
{
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), j);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void segment_norm_kernel(const double *__restrict__ raw, const uint64_t *__restrict__ off,
                                                           uint32_t n, uint32_t dim, double *__restrict__ norm)
{
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s >= n)
        return;                                   // (wave-uniform)
    const double *p = raw + off[s] * dim;
    const size_t len = (size_t)(off[s + 1] - off[s]) * dim;
    double memo = 0.0;
    double next = lane < (int)(len < 64 ? len : 64) ? p[lane] : 0.0;
    for (size_t base = 0; base < len; base += 64) {
        const double v = next;
        const size_t nb = base + 64 + lane;
        next = nb < len ? p[nb] : 0.0;
        const double sq = __dmul_rn(v, v);
        if (len - base >= 64) {
#pragma unroll
            for (int j = 0; j < 64; ++j)
                memo = __dadd_rn(lane_value(sq, j), memo);
        } else {
            const int cnt = (int)(len - base);
            for (int j = 0; j < cnt; ++j)
                memo = __dadd_rn(lane_value(sq, j), memo);
        }
    }
    if (lane == 0) {
        norm[s] = memo;
        // what the matrix-pipe filters of the refcos search need per segment (refcos_mfma.hip, refcos_q8.hip), once
        // instead of per pair: an upper bound of sqrt(norm) and the correctly rounded 1 / norm
        norm[n + s] = sqrt(memo) * (1.0 + 4.5e-16);
        norm[2 * (size_t)n + s] = 1.0 / memo;
    }
}

// dtw: per-segment frame count and max squared frame norm, and the set-wide max |value|
// (bits of non-negative floats order like unsigned integers; a non-finite value poisons the max).
// Outputs are indexed by record SLOT (blockIdx.y); the slot's segment is perm[slot].
__global__ void segment_stats_kernel(const double *__restrict__ raw, const uint64_t *__restrict__ off,
                                     const uint32_t *__restrict__ perm, uint32_t n, uint32_t dim,
                                     int32_t *__restrict__ len, unsigned *__restrict__ max_sqnorm_bits,
                                     unsigned *__restrict__ max_abs_bits)
{
    const uint32_t slot = blockIdx.y;
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = perm[slot];
    if (s >= n)
        return;                                   // (block-uniform: a block is one wave of one slot)
    const uint32_t nf = (uint32_t)(off[s + 1] - off[s]);
    if (f == 0)
        len[slot] = (int32_t)nf;
    float sqf = 0.f, maf = 0.f;
    if (f < nf) {
        const double *p = raw + (off[s] + f) * dim;
        double sq = 0.0, ma = 0.0;
        for (uint32_t e = 0; e < dim; ++e) {
            const double v = p[e];
            sq += v * v;
            const double av = fabs(v);
            ma = (av > ma || av != av) ? av : ma;     // NaN sticks
        }
        sqf = (float)sq * 1.000001f;              // rounded up
        maf = (float)ma * 1.000001f;
        if (!(sqf == sqf) || !(maf == maf)) { sqf = __builtin_inff(); maf = __builtin_inff(); }
    }
    // one atomic per wave and per slot, none on a set-wide word: half a million atomics on ONE word (the set-wide
    // maximum) were what this kernel's 0.6 ms at 4096 segments of 128 frames consisted of
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sqf = fmaxf(sqf, __shfl_xor(sqf, o));
        maf = fmaxf(maf, __shfl_xor(maf, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&max_sqnorm_bits[slot], __float_as_uint(sqf));
        atomicMax(&max_abs_bits[slot], __float_as_uint(maf));      // per slot: the host takes the set-wide maximum
    }
}

// ---- cached device blocks ----------------------------------------------------------------------
// Blocks up to 512 MB are kept (at most 4 GB per context, of 288): a batch of host targets (ssym_match_batch, the
// drop-in caller's pattern) allocates and frees ~250 MB of values, records and staging per call, and going to the
// driver for them cost more than a millisecond of a 39 ms call.  Larger blocks go straight to the driver.
constexpr size_t kCacheBlockMax = (size_t)512 << 20;
constexpr size_t kCacheTotalMax = (size_t)4 << 30;

static size_t round_block(size_t bytes)
{
    size_t r = 256;
    while (r < bytes)
        r <<= 1;
    return r;
}

int32_t dev_alloc(ssym_ctx *ctx, void **p, size_t bytes)
{
    const size_t want = bytes > kCacheBlockMax ? bytes : round_block(std::max<size_t>(bytes, 1));
    auto it = ctx->free_blocks.find(want);
    if (it != ctx->free_blocks.end()) {
        *p = it->second;
        ctx->free_blocks.erase(it);
        ctx->free_bytes -= want;
    } else {
        SSYM_HIP_CHECK(ctx, hipMalloc(p, want));
    }
    ctx->live_blocks[*p] = want;
    return SSYM_OK;
}

void dev_free(ssym_ctx *ctx, void *p)
{
    if (!p)
        return;
    if (ctx) {
        auto it = ctx->live_blocks.find(p);
        if (it != ctx->live_blocks.end()) {
            const size_t sz = it->second;
            ctx->live_blocks.erase(it);
            if (sz <= kCacheBlockMax && ctx->free_bytes + sz <= kCacheTotalMax) {
                ctx->free_blocks.emplace(sz, p);
                ctx->free_bytes += sz;
                return;
            }
        }
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(p);
}

void dev_cache_release(ssym_ctx *ctx)
{
    for (auto &kv : ctx->free_blocks)
        (void)hipFree(kv.second);
    ctx->free_blocks.clear();
    ctx->free_bytes = 0;
}

// ---- pinned staging ------------------------------------------------------------------------------
constexpr size_t kStageBytes = (size_t)1 << 20, kStageMaxCopy = (size_t)256 << 10;

char *stage_take(ssym_ctx *ctx, size_t bytes)
{
    if (bytes > kStageMaxCopy)
        return nullptr;
    if (!ctx->stage) {
        if (hipHostMalloc((void **)&ctx->stage, kStageBytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stage = nullptr;
            return nullptr;
        }
        ctx->stage_cap = kStageBytes;
    }
    const size_t at = (ctx->stage_cur + 63) & ~(size_t)63;
    if (at + bytes > ctx->stage_cap)
        return nullptr;
    ctx->stage_cur = at + bytes;
    return ctx->stage + at;
}

int32_t stage_h2d(ssym_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0)
        return SSYM_OK;
    char *p = ctx->api_depth > 0 ? stage_take(ctx, bytes) : nullptr;
    if (p) {
        memcpy(p, src_host, bytes);
        src_host = p;
    }
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SSYM_OK;
}

int32_t stage_d2h(ssym_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0)
        return SSYM_OK;
    char *p = ctx->api_depth > 0 ? stage_take(ctx, bytes) : nullptr;
    if (p) {
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync(p, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ctx->pending_d2h.push_back({dst_host, p, bytes});
        return SSYM_OK;
    }
    SSYM_HIP_CHECK(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return SSYM_OK;
}

void stage_finish(ssym_ctx *ctx)
{
    for (const auto &d : ctx->pending_d2h)
        memcpy(d.user, d.pinned, d.bytes);
    ctx->pending_d2h.clear();
}

void free_segments(ssym_ctx *ctx, SegmentSet &set)
{
    dev_free(ctx, set.raw);
    dev_free(ctx, set.off);
    dev_free(ctx, set.norm);
    dev_free(ctx, set.rec);
    set.rec = nullptr;
    dev_free(ctx, set.len);
    dev_free(ctx, set.max_sqnorm);
    dev_free(ctx, set.perm);
    dev_free(ctx, set.centroid);
    dev_free(ctx, set.len_order);
    refcos_q8_release(ctx, set);
    set = SegmentSet{};
}

// (Re)build everything derived from set.raw / set.h_off.
static int32_t build_derived(ssym_ctx *ctx, SegmentSet &set)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = set.n, dim = set.dim;
    dev_free(ctx, set.off); set.off = nullptr;
    dev_free(ctx, set.norm); set.norm = nullptr;
    dev_free(ctx, set.len); set.len = nullptr;
    dev_free(ctx, set.max_sqnorm); set.max_sqnorm = nullptr;
    dev_free(ctx, set.perm); set.perm = nullptr;
    dev_free(ctx, set.centroid); set.centroid = nullptr; set.centroid_n = 0;
    dev_free(ctx, set.len_order); set.len_order = nullptr; set.len_order_n = 0;
    refcos_q8_release(ctx, set);

    set.max_frames = 0;
    for (uint32_t i = 0; i < n; ++i)
        set.max_frames = std::max<uint32_t>(set.max_frames, (uint32_t)(set.h_off[i + 1] - set.h_off[i]));

    { int32_t rca = dev_alloc(ctx, (void **)&set.off, sizeof(uint64_t) * (n + 1)); if (rca != SSYM_OK) return rca; }
    { int32_t rcs = stage_h2d(ctx, set.off, set.h_off.data(), sizeof(uint64_t) * (n + 1)); if (rcs != SSYM_OK) return rcs; }
    if (n == 0) {
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSYM_OK;
    }

    if (ctx->metric == SSYM_METRIC_REFCOS) {
        { int32_t rca = dev_alloc(ctx, (void **)&set.norm, sizeof(double) * 3 * (size_t)n); if (rca != SSYM_OK) return rca; }
        segment_norm_kernel<<<(n + 3) / 4, 256, 0, st>>>(set.raw, set.off, n, dim, set.norm);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
    } else if (ctx->pack_light && !set.is_source) {
        // a handful of short queries that will be scored by the exact kernel on every pair (capi.hip, kFlagFewTargets):
        // no slot order, no statistics, no synchronisation for them
        set.light = true;
    } else {
        // sources: 8 per workgroup; targets: groups of 32, 8 groups per workgroup in the banded kernel
        const uint32_t quantum = (!set.is_source && ctx->band >= 0) ? 256u : 32u;
        set.n_pad = (n + quantum - 1) / quantum * quantum;
        uint32_t mf = std::max<uint32_t>(set.max_frames, 1);
        FilterShape shape = filter_shape((int)mf);
        set.frames_pad = set.is_source ? (shape.nt ? (uint32_t)shape.rows() : mf) : mf;
        // slots ordered by length (stable: equal lengths keep the caller's order)
        set.h_perm.assign(set.n_pad, 0xffffffffu);
        for (uint32_t i = 0; i < n; ++i)
            set.h_perm[i] = i;
        std::stable_sort(set.h_perm.begin(), set.h_perm.begin() + n, [&](uint32_t a, uint32_t b) {
            return set.h_off[a + 1] - set.h_off[a] < set.h_off[b + 1] - set.h_off[b];
        });
        { int32_t rca = dev_alloc(ctx, (void **)&set.perm, sizeof(uint32_t) * set.n_pad); if (rca != SSYM_OK) return rca; }
        { int32_t rcs = stage_h2d(ctx, set.perm, set.h_perm.data(), sizeof(uint32_t) * set.n_pad); if (rcs != SSYM_OK) return rcs; }
        { int32_t rca = dev_alloc(ctx, (void **)&set.len, sizeof(int32_t) * set.n_pad); if (rca != SSYM_OK) return rca; }
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(set.len, 0, sizeof(int32_t) * set.n_pad, st));
        // [n_pad] max squared frame norm per slot (read by the kernels), then [n_pad] max |value| per slot (host only), then
        // [n_pad] the largest distance between a frame of the slot and the frame its filter record represents (written when
        // the records are built, dtw_filter.hip; read by the selection's margin, dtw_margin.hpp)
        { int32_t rca = dev_alloc(ctx, (void **)&set.max_sqnorm, sizeof(float) * 3 * (size_t)set.n_pad); if (rca != SSYM_OK) return rca; }
        SSYM_HIP_CHECK(ctx, hipMemsetAsync(set.max_sqnorm, 0, sizeof(float) * 3 * (size_t)set.n_pad, st));
        dim3 grid((mf + 63) / 64, n);
        segment_stats_kernel<<<grid, 64, 0, st>>>(set.raw, set.off, set.perm, n, dim, set.len,
                                                  (unsigned *)set.max_sqnorm,
                                                  (unsigned *)set.max_sqnorm + set.n_pad);
        SSYM_HIP_CHECK(ctx, hipGetLastError());
        std::vector<float> h(2 * (size_t)set.n_pad);
        SSYM_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), set.max_sqnorm, sizeof(float) * 2 * (size_t)set.n_pad,
                                           hipMemcpyDeviceToHost, st));
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        float m = 0.f, ma = 0.f;
        for (uint32_t i = 0; i < set.n_pad; ++i) {
            m = std::max(m, h[i]);
            ma = std::max(ma, h[set.n_pad + i]);
        }
        set.max_sqnorm_all = (double)m;
        set.max_abs = (double)ma;
        dev_free(ctx, set.rec);
        set.rec = nullptr;
        set.rec_scale = 0.0;
        set.rec_bytes = 0;
    }
    if (!ctx->defer_sync)
        SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    return SSYM_OK;
}

static int32_t validate_offsets(ssym_ctx *ctx, const uint64_t *off, uint32_t n)
{
    if (!off) {
        ctx->err = "frame_offsets is NULL";
        return SSYM_E_INVALID;
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) {
            ctx->err = "frame_offsets must be non-decreasing";
            return SSYM_E_INVALID;
        }
        if (off[i + 1] - off[i] > 0x7fffffffull) {
            ctx->err = "segment too long";
            return SSYM_E_INVALID;
        }
    }
    return SSYM_OK;
}

// Copy `count_vals` feature values (ctx dtype) starting at `feats` into set.raw + dst_val_offset.
static int32_t upload_values(ssym_ctx *ctx, SegmentSet &set, size_t dst_val_offset, const void *feats,
                             bool on_device, size_t count_vals)
{
    if (count_vals == 0)
        return SSYM_OK;
    hipStream_t st = ctx->stream;
    double *dst = set.raw + dst_val_offset;
    if (ctx->dtype == SSYM_DTYPE_F64) {
        if (on_device) {
            SSYM_HIP_CHECK(ctx, hipMemcpyAsync(dst, feats, count_vals * sizeof(double), hipMemcpyDeviceToDevice, st));
        } else {
            int32_t rcs = stage_h2d(ctx, dst, feats, count_vals * sizeof(double));
            if (rcs != SSYM_OK)
                return rcs;
        }
        if (!ctx->defer_sync)      // the caller's buffer is free to go when this returns
            SSYM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSYM_OK;
    }
    const float *src_dev = (const float *)feats;
    float *tmp = nullptr;
    if (!on_device) {
        { int32_t rca = dev_alloc(ctx, (void **)&tmp, count_vals * sizeof(float)); if (rca != SSYM_OK) return rca; }
        int32_t rcs = stage_h2d(ctx, tmp, feats, count_vals * sizeof(float));
        if (rcs != SSYM_OK) {
            dev_free(ctx, tmp);
            return rcs;
        }
        src_dev = tmp;
    }
    unsigned blocks = (unsigned)std::min<size_t>((count_vals + 255) / 256, 4096);
    widen_f32_kernel<<<blocks, 256, 0, st>>>(src_dev, dst, count_vals);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !ctx->defer_sync)
        e = hipStreamSynchronize(st);
    if (tmp)
        dev_free(ctx, tmp);
    if (e != hipSuccess) {
        ctx->err = std::string("widen_f32_kernel: ") + hipGetErrorString(e);
        return SSYM_E_HIP;
    }
    return SSYM_OK;
}

int32_t pack_segments(ssym_ctx *ctx, SegmentSet &set, const void *feats, bool feats_on_device,
                      const uint64_t *frame_offsets, uint32_t n, uint32_t dim, bool is_source)
{
    if (dim == 0 || dim > 4096) {
        ctx->err = "dim must be in [1, 4096]";
        return SSYM_E_INVALID;
    }
    if (n > 0) {
        int32_t rc = validate_offsets(ctx, frame_offsets, n);
        if (rc != SSYM_OK)
            return rc;
    }
    set.n = n;
    set.dim = dim;
    set.is_source = is_source;
    set.h_off.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i)
        set.h_off[i + 1] = frame_offsets[i + 1] - frame_offsets[0];
    set.total_frames = set.h_off[n];
    size_t vals = (size_t)set.total_frames * dim;
    if (vals > 0 && !feats) {
        ctx->err = "feats is NULL";
        return SSYM_E_INVALID;
    }
    set.raw_capacity_vals = std::max<size_t>(vals, 1);
    { int32_t rca = dev_alloc(ctx, (void **)&set.raw, (set.raw_capacity_vals + kRawTailPad) * sizeof(double)); if (rca != SSYM_OK) return rca; }
    if (vals > 0) {
        size_t esz = ctx->dtype == SSYM_DTYPE_F64 ? sizeof(double) : sizeof(float);
        const char *base = (const char *)feats + (size_t)frame_offsets[0] * dim * esz;
        int32_t rc = upload_values(ctx, set, 0, base, feats_on_device, vals);
        if (rc != SSYM_OK)
            return rc;
    }
    return build_derived(ctx, set);
}

int32_t append_segments(ssym_ctx *ctx, SegmentSet &set, const void *feats,
                        const uint64_t *frame_offsets, uint32_t n)
{
    if (n == 0)
        return SSYM_OK;
    int32_t rc = validate_offsets(ctx, frame_offsets, n);
    if (rc != SSYM_OK)
        return rc;
    const uint32_t dim = set.dim;
    uint64_t add_frames = frame_offsets[n] - frame_offsets[0];
    size_t old_vals = (size_t)set.total_frames * dim;
    size_t add_vals = (size_t)add_frames * dim;
    if (add_vals > 0 && !feats) {
        ctx->err = "feats is NULL";
        return SSYM_E_INVALID;
    }
    if (old_vals + add_vals > set.raw_capacity_vals) {
        size_t cap = std::max(old_vals + add_vals, set.raw_capacity_vals * 2);
        double *nraw = nullptr;
        { int32_t rca = dev_alloc(ctx, (void **)&nraw, (cap + kRawTailPad) * sizeof(double)); if (rca != SSYM_OK) return rca; }
        if (old_vals)
            SSYM_HIP_CHECK(ctx, hipMemcpyAsync(nraw, set.raw, old_vals * sizeof(double),
                                               hipMemcpyDeviceToDevice, ctx->stream));
        dev_free(ctx, set.raw);        // stream-ordered: the copy above is queued before any reuse
        set.raw = nraw;
        set.raw_capacity_vals = cap;
    }
    if (add_vals > 0) {
        size_t esz = ctx->dtype == SSYM_DTYPE_F64 ? sizeof(double) : sizeof(float);
        const char *base = (const char *)feats + (size_t)frame_offsets[0] * dim * esz;
        rc = upload_values(ctx, set, old_vals, base, false, add_vals);
        if (rc != SSYM_OK)
            return rc;
    }
    uint64_t base_frames = set.total_frames;
    for (uint32_t i = 0; i < n; ++i)
        set.h_off.push_back(base_frames + (frame_offsets[i + 1] - frame_offsets[0]));
    set.n += n;
    set.total_frames += add_frames;
    return build_derived(ctx, set);
}

}  // namespace ssym
