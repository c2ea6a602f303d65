/* tests/asan/asan_host.c -- host-side sanitizer run ("ASan/UBSan on host + CPU-backend code").
 *
 * Built by tests/asan/Makefile with -fsanitize=address,undefined together with oracle/ssym_oracle.c (the CPU oracle:
 * test infrastructure, this is one of the places allowed to link it).  Two parts:
 *   1. the oracle's edge cases -- empty segments, zero-length dictionaries, one-value segments, NaN / inf values,
 *      ssym_oracle_topk with k > n, bands that cut every path, the reconstruction's zero-length pieces, the MFCC of a
 *      sound shorter than one window -- every allocation and index under the sanitizers;
 *   2. the C ABI's paths that need no GPU (argv[1] = path of libsoundsym_amd.so, loaded with dlopen: the product library
 *      itself is device code and is not instrumented; what is checked is that its no-device / bad-argument / struct-size
 *      paths return their status codes without touching memory they should not, as seen from an instrumented caller).
 * CPU box only: never run on the GPU box (GPU sanitizers are not available on this pool).
 * Exit code 0 = everything as expected; a sanitizer report aborts with its own non-zero code. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "soundsym_amd.h"

double ssym_oracle_norm(const double *me, size_t n);
double ssym_oracle_dot(const double *xs, const double *ys, size_t len);
double ssym_oracle_cosine_sim(const double *me, size_t nme, const double *you, size_t nyou);
int64_t ssym_oracle_at_distance(const double *feats, const uint64_t *off, uint32_t n_src, uint32_t dim, double distance,
                                const double *you, uint64_t you_frames, double *out_min);
int ssym_oracle_refcos_match_all(const double *src, const uint64_t *src_off, uint32_t n_src, const double *tgt,
                                 const uint64_t *tgt_off, uint32_t n_tgt, uint32_t dim, const double *distance,
                                 int64_t *out_idx, double *out_val);
void ssym_oracle_reconstruct(const double *src_samples, const uint64_t *src_off, const int64_t *idx,
                             const uint64_t *out_off, uint32_t n_tgt, double *out);
int32_t ssym_oracle_pcm32(double sample);
double ssym_oracle_dtw(const double *a, uint64_t fa, const double *b, uint64_t fb, uint32_t dim, int64_t band, int squared);
int ssym_oracle_dtw_match_all(const double *src, const uint64_t *src_off, uint32_t n_src, const double *tgt,
                              const uint64_t *tgt_off, uint32_t n_tgt, uint32_t dim, int64_t band, int squared,
                              int nthreads, int64_t *out_idx, double *out_cost, double *cost_matrix);
void ssym_oracle_refcos_matrix(const double *src, const uint64_t *src_off, uint32_t n_src, const double *tgt,
                               const uint64_t *tgt_off, uint32_t n_tgt, uint32_t dim, double *out);
int ssym_oracle_topk(const double *values, uint32_t n_src, uint32_t n_tgt, const double *distance,
                     double default_distance, double fold_start, uint32_t k, int64_t *out_idx, double *out_key);
uint64_t ssym_oracle_mfcc_num_frames(uint64_t n, int pad_tail);
int ssym_oracle_mfcc(const double *samples, uint64_t n, double rate, uint32_t nc, int pad_tail, double *out);

static int failures = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            fprintf(stderr, "asan_host: %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
            ++failures;                                                                  \
        }                                                                                \
    } while (0)

/* exactly-sized heap blocks, so that one element too far is a report */
static double *dvec(size_t n, double fill)
{
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));
    for (size_t i = 0; i < n; ++i)
        p[i] = fill + 0.25 * (double)(i % 7);
    return p;
}

static void oracle_edges(void)
{
    enum { DIM = 12 };
    /* five segments of 0, 1, 3, 0, 9 frames: empty ones first, in the middle, and a dot of 9 * 12 = 108 values
       (13 blocks of eight + a tail of 4: rulinalg's unrolled dot and its tail, src/sound.rs:31) */
    const uint64_t off[6] = {0, 0, 1, 4, 4, 13};
    double *src = dvec(13 * DIM, 0.1);
    const uint64_t toff[4] = {0, 2, 2, 7};                 /* targets of 2, 0, 5 frames */
    double *tgt = dvec(7 * DIM, -0.3);
    int64_t idx[3];
    double val[3];
    CHECK(ssym_oracle_refcos_match_all(src, off, 5, tgt, toff, 3, DIM, NULL, idx, val) == 0);
    CHECK(idx[1] == 0);                                    /* an empty target: every key is NaN, NaN never wins, index 0 (src/sound.rs:361-367) */
    CHECK(ssym_oracle_refcos_match_all(src, off, 0, tgt, toff, 3, DIM, NULL, idx, val) == -1);   /* empty dictionary */
    const double dist[3] = {0.0, NAN, 1.9};
    CHECK(ssym_oracle_refcos_match_all(src, off, 5, tgt, toff, 3, DIM, dist, idx, val) == 0);
    CHECK(ssym_oracle_norm(src, 0) == 0.0 && ssym_oracle_dot(src, tgt, 0) == 0.0);
    CHECK(ssym_oracle_cosine_sim(src, 0, tgt, 0) != ssym_oracle_cosine_sim(src, 0, tgt, 0));    /* 0 / 0 = NaN, no guard (:32) */
    double mn = 0.0;
    CHECK(ssym_oracle_at_distance(src, off, 0, DIM, 1.0, tgt, 2, &mn) == -1);
    CHECK(ssym_oracle_at_distance(src, off, 5, DIM, 1.0, tgt, 0, &mn) == 0);

    /* dtw: empty sides, a band that cuts every path, squared cost, one frame against many */
    CHECK(isinf(ssym_oracle_dtw(src, 0, tgt, 5, DIM, -1, 0)) && isinf(ssym_oracle_dtw(src, 9, tgt, 0, DIM, -1, 0)));
    CHECK(isinf(ssym_oracle_dtw(src + 4 * DIM, 9, tgt, 2, DIM, 3, 0)));          /* |9 - 2| > 3: unreachable */
    CHECK(isfinite(ssym_oracle_dtw(src + 4 * DIM, 9, tgt, 2, DIM, 7, 1)));
    CHECK(isfinite(ssym_oracle_dtw(src + 1 * DIM, 1, tgt + 2 * DIM, 5, DIM, -1, 0)));
    double *mat = dvec(5 * 3, 0.0);
    double cost[3];
    for (int threads = 1; threads <= 3; threads += 2) {
        CHECK(ssym_oracle_dtw_match_all(src, off, 5, tgt, toff, 3, DIM, -1, 0, threads, idx, cost, mat) == 0);
        CHECK(isinf(mat[0 * 3 + 0]) && isinf(cost[1]) && idx[1] == 0);           /* empty source row; empty target: fold start */
    }
    CHECK(ssym_oracle_dtw_match_all(src, off, 5, tgt, toff, 3, DIM, 0, 1, 2, idx, cost, NULL) == 0);
    CHECK(ssym_oracle_dtw_match_all(src, off, 0, tgt, toff, 3, DIM, -1, 0, 1, idx, cost, NULL) == -1);

    /* top-k with k > n, k = 0 rows, NaN values, per-target distances */
    ssym_oracle_refcos_matrix(src, off, 5, tgt, toff, 3, DIM, mat);
    enum { K = 8 };
    int64_t *kidx = (int64_t *)malloc(sizeof(int64_t) * 3 * K);
    double *kkey = dvec(3 * K, 0.0);
    CHECK(ssym_oracle_topk(mat, 5, 3, NULL, 1.0, 2.0, K, kidx, kkey) == 0);      /* k = 8 > n = 5: the rows' tails are the fold start */
    CHECK(ssym_oracle_topk(mat, 5, 3, dist, 1.0, INFINITY, 1, kidx, kkey) == 0);
    CHECK(ssym_oracle_topk(mat, 0, 3, NULL, 1.0, 2.0, 2, kidx, kkey) == 0);      /* no sources at all */
    free(kidx);
    free(kkey);

    /* reconstruction: zero-length pieces on either side, pad and truncate (src/sound.rs:456-465) */
    const uint64_t soff[4] = {0, 0, 5, 12};                 /* sounds of 0, 5, 7 samples */
    double *smp = dvec(12, 0.5);
    const int64_t pick[4] = {0, 2, 1, 1};
    const uint64_t ooff[5] = {0, 3, 3, 13, 15};             /* pieces of 3 (from nothing: zeros), 0, 10 (pad), 2 (truncate) */
    double *out = dvec(15, 9.0);
    ssym_oracle_reconstruct(smp, soff, pick, ooff, 4, out);
    CHECK(out[0] == 0.0 && out[2] == 0.0 && out[3] == smp[0] && out[12] == 0.0 && out[14] == smp[1]);
    CHECK(ssym_oracle_pcm32(NAN) == 0 && ssym_oracle_pcm32(2.0) == INT32_MAX && ssym_oracle_pcm32(-2.0) == INT32_MIN);

    /* MFCC front-end: shorter than a window, exactly one hop, padded tail */
    double *wave = dvec(1300, -0.2);
    for (int pad = 0; pad <= 1; ++pad) {
        const uint64_t lens[4] = {0, 255, 1024, 1300};
        for (int k = 0; k < 4; ++k) {
            const uint64_t nf = ssym_oracle_mfcc_num_frames(lens[k], pad);
            double *m = dvec((size_t)nf * 12, 0.0);
            CHECK(ssym_oracle_mfcc(wave, lens[k], 44100.0, 12, pad, m) == 0);
            free(m);
        }
    }
    free(wave);
    free(out);
    free(smp);
    free(mat);
    free(tgt);
    free(src);
}

/* ---- the C ABI without a GPU -------------------------------------------------------------------------------------- */
typedef int32_t (*fn_version)(void);
typedef int32_t (*fn_ctx_create)(const ssym_config *, ssym_ctx **);
typedef int32_t (*fn_ctx_destroy)(ssym_ctx *);
typedef const char *(*fn_last_error)(const ssym_ctx *);
typedef int32_t (*fn_get_timings)(const ssym_ctx *, ssym_timings *);
typedef int32_t (*fn_dict_destroy)(ssym_ctx *, ssym_dict *);

static void abi_no_device(const char *libpath)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        fprintf(stderr, "asan_host: dlopen(%s): %s\n", libpath, dlerror());
        ++failures;
        return;
    }
    fn_version version = (fn_version)dlsym(h, "ssym_abi_version");
    fn_ctx_create create = (fn_ctx_create)dlsym(h, "ssym_ctx_create");
    fn_ctx_destroy destroy = (fn_ctx_destroy)dlsym(h, "ssym_ctx_destroy");
    fn_last_error last_error = (fn_last_error)dlsym(h, "ssym_last_error");
    fn_get_timings get_timings = (fn_get_timings)dlsym(h, "ssym_get_timings");
    fn_dict_destroy dict_destroy = (fn_dict_destroy)dlsym(h, "ssym_dict_destroy");
    CHECK(version && create && destroy && last_error && get_timings && dict_destroy);
    if (!(version && create && destroy && last_error && get_timings && dict_destroy))
        return;
    CHECK(version() == SSYM_ABI_VERSION);
    /* exactly-sized heap copies of the structs: a library that read or wrote past sizeof(struct) would be reported */
    ssym_config *cfg = (ssym_config *)calloc(1, sizeof(ssym_config));
    ssym_ctx **out = (ssym_ctx **)calloc(1, sizeof(ssym_ctx *));
    cfg->struct_size = (uint32_t)sizeof(ssym_config);
    cfg->metric = SSYM_METRIC_DTW;
    cfg->dtype = SSYM_DTYPE_F32;
    cfg->band = -1;
    int32_t rc = create(cfg, out);
    /* this program only runs on the CPU box: no device; a GPU box would return SSYM_OK (then destroy) */
    CHECK(rc == SSYM_E_NO_DEVICE || rc == SSYM_OK);
    if (rc == SSYM_OK) {
        CHECK(destroy(*out) == SSYM_OK);
    } else {
        CHECK(*out == NULL);
        const char *msg = last_error(NULL);
        CHECK(msg != NULL && strlen(msg) > 0);
    }
    /* bad arguments: NULL pointers, a struct size from another ABI, unknown enums */
    CHECK(create(NULL, out) == SSYM_E_INVALID);
    CHECK(create(cfg, NULL) == SSYM_E_INVALID);
    cfg->struct_size = 4;
    CHECK(create(cfg, out) == SSYM_E_INVALID && *out == NULL);
    cfg->struct_size = (uint32_t)sizeof(ssym_config);
    cfg->metric = 77;
    CHECK(create(cfg, out) == SSYM_E_INVALID && *out == NULL);
    cfg->metric = SSYM_METRIC_REFCOS;
    cfg->dtype = 99;
    CHECK(create(cfg, out) == SSYM_E_INVALID && *out == NULL);
    ssym_timings *tm = (ssym_timings *)calloc(1, sizeof(ssym_timings));
    CHECK(get_timings(NULL, tm) == SSYM_E_INVALID);
    CHECK(destroy(NULL) == SSYM_OK);
    CHECK(dict_destroy(NULL, NULL) == SSYM_OK);
    free(tm);
    free(out);
    free(cfg);
    /* (the library stays loaded: unloading a HIP runtime is not something this test is about) */
}

int main(int argc, char **argv)
{
    oracle_edges();
    if (argc > 1)
        abi_no_device(argv[1]);
    if (failures) {
        fprintf(stderr, "asan_host: %d check(s) failed\n", failures);
        return 1;
    }
    printf("asan_host ok (oracle edge cases%s)\n", argc > 1 ? " + C ABI without a device" : "");
    return 0;
}
