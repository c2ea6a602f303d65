/*
 * ssym_oracle.c -- CPU ORACLE for the soundsym segment-distance matching path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's `cpu_baseline` leg may load this library.  The product
 * (soundsym_amd/, libsoundsym_amd.so) never links, imports or calls anything in oracle/.
 *
 * What it restates (all citations relative to /root/reference):
 *   refcos  : src/sound.rs:22-33   cosine_sim      (prefix dot / product of SQUARED norms)
 *             src/sound.rs:35-38   norm            (left fold, item*item + memo, no sqrt)
 *             src/sound.rs:345-348 match_sound     (= at_distance(1.0, other))
 *             src/sound.rs:351-370 at_distance     (|sim - distance|, fold (0, 2.0), strict <)
 *             src/sound.rs:451-455 clone_from_dictionary (one match_sound per target, in order)
 *             src/sound.rs:456-465 length fit of the matched samples (zero-pad / truncate)
 *             src/sound.rs:475-480 to_sound concatenation, :139 write_file sample conversion
 *   third-party arithmetic on the path: `rulinalg::utils::dot`, crate rulinalg = "0.4.2"
 *             (Cargo.toml:15; call site src/sound.rs:31).  Its source is NOT under
 *             /root/reference; the published 0.4.2 algorithm is restated in ssym_oracle_dot below.
 *   dtw     : NOT IN THE REFERENCE.  The definition is this
 *             repository's own (DESIGN.md "DTW metric"), stated in ssym_oracle_dtw below.
 *
 * PARITY STATUS: **parity unpinned** for match indices and similarity values.  The reference
 * cannot be built here (no rustc/cargo; un-vendored, partly unpinned dependencies) and its own
 * tests hold no golden vector with numbers for this path.  What pins this file:
 *   - the source text cited above, followed operation by operation;
 *   - the arithmetic the reference's test_angular_distance (src/sound.rs:611-615) implies:
 *     cosine_sim(m, m) = 0.85 / 0.7225 > 1 for m = [0.1,0.4,0.2,0.8,0,...] (squared norms);
 *   - hand-derived known answers in tests/golden/refcos_kat.json.
 *
 * Build: strict IEEE, no FMA contraction -- see oracle/Makefile (-O2 -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ssym_rulinalg.h"   /* ../include: the one constant shared with the product (the dot's association) */

#ifdef _OPENMP
#include <omp.h>
#endif

#define SSYM_ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * refcos
 * ---------------------------------------------------------------------------------------- */

/* src/sound.rs:35-38   me.iter().fold(0., |memo, item| item * item + memo) */
SSYM_ORACLE_API double ssym_oracle_norm(const double *me, size_t n)
{
    double memo = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double sq = me[i] * me[i];
        memo = sq + memo;
    }
    return memo;
}

/* rulinalg 0.4.2 `utils::dot` (called at src/sound.rs:31), restated from the published crate:
 * eight independent accumulators over blocks of eight, combined as
 *   s = s + (p0+p4); s = s + (p1+p5); s = s + (p2+p6); s = s + (p3+p7);       (SSYM_RULINALG_COMBINE 0, the default)
 * or left-associated, s = s + p0 + p4; ...                                    (SSYM_RULINALG_COMBINE 1)
 * -- which of the two the crate uses could not be checked in this image (its source is not under /root/reference);
 * the choice is ONE constant shared with the product, include/ssym_rulinalg.h --,
 * then the (len mod 8) tail is added to s one product at a time.  Products and sums are
 * separately rounded (no fused multiply-add under default rustc codegen). */
#define SSYM_ORACLE_ADD(x, y) ((x) + (y))     /* one rounded f64 addition (-ffp-contract=off) */
SSYM_ORACLE_API double ssym_oracle_dot(const double *xs, const double *ys, size_t len)
{
    double s = 0.0;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0, p5 = 0.0, p6 = 0.0, p7 = 0.0;
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        p0 = p0 + xs[i + 0] * ys[i + 0];
        p1 = p1 + xs[i + 1] * ys[i + 1];
        p2 = p2 + xs[i + 2] * ys[i + 2];
        p3 = p3 + xs[i + 3] * ys[i + 3];
        p4 = p4 + xs[i + 4] * ys[i + 4];
        p5 = p5 + xs[i + 5] * ys[i + 5];
        p6 = p6 + xs[i + 6] * ys[i + 6];
        p7 = p7 + xs[i + 7] * ys[i + 7];
    }
    s = SSYM_RULINALG_STEP(SSYM_ORACLE_ADD, s, p0, p4);     /* association: include/ssym_rulinalg.h */
    s = SSYM_RULINALG_STEP(SSYM_ORACLE_ADD, s, p1, p5);
    s = SSYM_RULINALG_STEP(SSYM_ORACLE_ADD, s, p2, p6);
    s = SSYM_RULINALG_STEP(SSYM_ORACLE_ADD, s, p3, p7);
    for (; i < len; ++i)
        s = s + xs[i] * ys[i];
    return s;
}

/* which association this build of the oracle uses (tests print it; tools/rulinalg_variants.sh checks it) */
SSYM_ORACLE_API int ssym_oracle_rulinalg_combine(void) { return SSYM_RULINALG_COMBINE; }

/* src/sound.rs:22-33 */
SSYM_ORACLE_API double ssym_oracle_cosine_sim(const double *me, size_t nme,
                                              const double *you, size_t nyou)
{
    size_t len = nme < nyou ? nme : nyou;               /* :24-28 */
    double nrm = ssym_oracle_norm(me, nme) * ssym_oracle_norm(you, nyou); /* :30 full vectors */
    double dot = ssym_oracle_dot(me, you, len);          /* :31 common prefix */
    return dot / nrm;                                    /* :32 no zero guard */
}

/* src/sound.rs:351-370.  Dictionary = n_src segments stored back to back in `feats`;
 * segment i occupies values [off[i]*dim, off[i+1]*dim) (offsets are in FRAMES, src/sound.rs:335
 * gives each segment seg/HOP frames of NCOEFFS values).
 * Returns the chosen index, or -1 for an empty dictionary (the reference panics there, :369).
 * *out_min receives the winning |sim - distance| (the reference computes and discards it). */
SSYM_ORACLE_API int64_t ssym_oracle_at_distance(const double *feats, const uint64_t *off,
                                                uint32_t n_src, uint32_t dim, double distance,
                                                const double *you, uint64_t you_frames,
                                                double *out_min)
{
    if (n_src == 0)
        return -1;
    size_t min_idx = 0;          /* fold init (0usize, 2f64), :361 */
    double min_distance = 2.0;
    for (uint32_t i = 0; i < n_src; ++i) {
        const double *me = feats + off[i] * dim;
        size_t nme = (size_t)(off[i + 1] - off[i]) * dim;
        double sim = ssym_oracle_cosine_sim(me, nme, you, (size_t)you_frames * dim); /* :354 */
        double v = fabs(sim - distance);                 /* :359 */
        if (v < min_distance) {                          /* :362 strict <, NaN never wins */
            min_idx = i;
            min_distance = v;
        }
    }
    if (out_min)
        *out_min = min_distance;
    return (int64_t)min_idx;
}

/* The loop of src/sound.rs:451-455 (clone_from_dictionary) / :440-446 (morph_to):
 * one at_distance per target, in target order.  distance == NULL means 1.0 for every target
 * (match_sound, :346-348). */
SSYM_ORACLE_API int ssym_oracle_refcos_match_all(const double *src, const uint64_t *src_off,
                                                 uint32_t n_src, const double *tgt,
                                                 const uint64_t *tgt_off, uint32_t n_tgt,
                                                 uint32_t dim, const double *distance,
                                                 int64_t *out_idx, double *out_val)
{
    if (n_src == 0)
        return -1;
    for (uint32_t t = 0; t < n_tgt; ++t) {
        double v;
        out_idx[t] = ssym_oracle_at_distance(src, src_off, n_src, dim,
                                             distance ? distance[t] : 1.0,
                                             tgt + tgt_off[t] * dim,
                                             tgt_off[t + 1] - tgt_off[t], &v);
        if (out_val)
            out_val[t] = v;
    }
    return 0;
}

/* src/sound.rs:456-465: fit the matched sound's samples to the target's sample count --
 * zero-pad when the target is longer (:457-459), truncate when shorter (:460-462), copy when
 * equal (:463-464 shares the Arc).  `out` has room for n_target samples. */
SSYM_ORACLE_API void ssym_oracle_length_fit(const double *matched, uint64_t n_matched,
                                            uint64_t n_target, double *out)
{
    uint64_t ncopy = n_matched < n_target ? n_matched : n_target;
    memcpy(out, matched, (size_t)ncopy * sizeof(double));
    for (uint64_t i = ncopy; i < n_target; ++i)
        out[i] = 0.0;
}

/* The tail of clone_from_dictionary + to_sound (src/sound.rs:456-465, 475-480): every matched
 * sound's samples fitted to its target's sample count, concatenated in target order.
 * src_off / out_off are SAMPLE offsets (n+1 entries each). */
SSYM_ORACLE_API void ssym_oracle_reconstruct(const double *src_samples, const uint64_t *src_off,
                                             const int64_t *idx, const uint64_t *out_off,
                                             uint32_t n_tgt, double *out)
{
    for (uint32_t t = 0; t < n_tgt; ++t) {
        const uint64_t s = (uint64_t)idx[t];
        ssym_oracle_length_fit(src_samples + src_off[s], src_off[s + 1] - src_off[s],
                               out_off[t + 1] - out_off[t], out + out_off[t]);
    }
}

/* Sound::write_file's sample conversion, src/sound.rs:139: `(i32::max_value() as f64 * sample) as
 * i32` -- Rust's float-to-int `as` truncates toward zero, saturates, and maps NaN to 0. */
SSYM_ORACLE_API int32_t ssym_oracle_pcm32(double sample)
{
    double v = 2147483647.0 * sample;
    if (v != v)
        return 0;
    if (v >= 2147483647.0)
        return INT32_MAX;
    if (v <= -2147483648.0)
        return INT32_MIN;
    return (int32_t)v;
}

/* ------------------------------------------------------------------------------------------
 * dtw  (this repository's definition; DESIGN.md "DTW metric")
 *
 *   c(i,j)  = sqrt( sum_{k=0}^{d-1} (a[i][k] - b[j][k])^2 )   (k ascending; sub, mul, add each
 *             rounded; f64)            -- or the un-rooted sum when squared != 0
 *   D(0,0)  = c(0,0)
 *   D(i,j)  = c(i,j) + min( D(i-1,j), D(i,j-1), D(i-1,j-1) )   (out of range = +inf)
 *   band    : band >= 0 restricts to |i-j| <= band (Sakoe-Chiba); cells outside are +inf
 *   cost    = D(Fa-1, Fb-1);   +inf when unreachable or either segment is empty
 * ---------------------------------------------------------------------------------------- */
SSYM_ORACLE_API double ssym_oracle_dtw(const double *a, uint64_t fa, const double *b, uint64_t fb,
                                       uint32_t dim, int64_t band, int squared)
{
    if (fa == 0 || fb == 0)
        return INFINITY;
    /* prev[j+1] = D(i-1, j); cur[j+1] = D(i, j); index 0 is the j = -1 boundary */
    double *prev = (double *)malloc((size_t)(fb + 1) * sizeof(double));
    double *cur = (double *)malloc((size_t)(fb + 1) * sizeof(double));
    for (uint64_t j = 0; j <= fb; ++j)
        prev[j] = INFINITY;
    prev[0] = 0.0; /* virtual D(-1,-1) = 0 so that D(0,0) = c(0,0) */
    for (uint64_t i = 0; i < fa; ++i) {
        cur[0] = INFINITY;
        const double *ai = a + i * dim;
        for (uint64_t j = 0; j < fb; ++j) {
            int64_t dij = (int64_t)i - (int64_t)j;
            if (band >= 0 && (dij > band || -dij > band)) {
                cur[j + 1] = INFINITY;
                continue;
            }
            const double *bj = b + j * dim;
            double acc = 0.0;
            for (uint32_t k = 0; k < dim; ++k) {
                double df = ai[k] - bj[k];
                double sq = df * df;
                acc = acc + sq;
            }
            double c = squared ? acc : sqrt(acc);
            double best = prev[j + 1];          /* D(i-1, j)   */
            if (cur[j] < best) best = cur[j];   /* D(i,   j-1) */
            if (prev[j] < best) best = prev[j]; /* D(i-1, j-1) */
            cur[j + 1] = c + best;
        }
        double *tmp = prev; prev = cur; cur = tmp;
        prev[0] = INFINITY; /* D(i, -1) for the next row's diagonal */
    }
    double r = prev[fb];
    free(prev);
    free(cur);
    return r;
}

/* idx[t] = argmin_s cost(s, t), strict < from (0, +inf): ties -> lowest s, mirrors
 * src/sound.rs:361-367.  cost_matrix (nullable) is [n_src][n_tgt].  nthreads <= 1 is serial. */
SSYM_ORACLE_API int ssym_oracle_dtw_match_all(const double *src, const uint64_t *src_off,
                                              uint32_t n_src, const double *tgt,
                                              const uint64_t *tgt_off, uint32_t n_tgt,
                                              uint32_t dim, int64_t band, int squared,
                                              int nthreads, int64_t *out_idx, double *out_cost,
                                              double *cost_matrix)
{
    if (n_src == 0)
        return -1;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t t = 0; t < (int64_t)n_tgt; ++t) {
        size_t min_idx = 0;
        double min_cost = INFINITY;
        const double *b = tgt + tgt_off[t] * dim;
        uint64_t fb = tgt_off[t + 1] - tgt_off[t];
        for (uint32_t s = 0; s < n_src; ++s) {
            double c = ssym_oracle_dtw(src + src_off[s] * dim, src_off[s + 1] - src_off[s],
                                       b, fb, dim, band, squared);
            if (cost_matrix)
                cost_matrix[(size_t)s * n_tgt + (size_t)t] = c;
            if (c < min_cost) {
                min_idx = s;
                min_cost = c;
            }
        }
        out_idx[t] = (int64_t)min_idx;
        if (out_cost)
            out_cost[t] = min_cost;
    }
    return 0;
}

/* Every cosine_sim(source s, target t) (src/sound.rs:22-33), out[s * n_tgt + t]. */
SSYM_ORACLE_API void ssym_oracle_refcos_matrix(const double *src, const uint64_t *src_off,
                                               uint32_t n_src, const double *tgt,
                                               const uint64_t *tgt_off, uint32_t n_tgt,
                                               uint32_t dim, double *out)
{
    for (uint32_t s = 0; s < n_src; ++s)
        for (uint32_t t = 0; t < n_tgt; ++t)
            out[(size_t)s * n_tgt + t] =
                ssym_oracle_cosine_sim(src + src_off[s] * dim, (size_t)(src_off[s + 1] - src_off[s]) * dim,
                                       tgt + tgt_off[t] * dim, (size_t)(tgt_off[t + 1] - tgt_off[t]) * dim);
}

/* Top-k candidates per target (the reference only takes the first,
 * src/sound.rs:351-370): the entries a repeated at_distance would return if each winner were
 * removed from the dictionary -- i.e. the sources ordered by (|value - distance|, index), keeping
 * those whose key is below the fold start (strict '<', so NaN keys never enter, :362).
 * Restated as a plain sort, independently of the GPU's round formulation.
 * values[s * n_tgt + t]; out_idx / out_key are [n_tgt][k]; missing entries: index -1, key NaN. */
typedef struct { double key; int64_t idx; } ssym_oracle_cand;
static int ssym_oracle_cand_cmp(const void *a, const void *b)
{
    const ssym_oracle_cand *x = (const ssym_oracle_cand *)a, *y = (const ssym_oracle_cand *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
SSYM_ORACLE_API int ssym_oracle_topk(const double *values, uint32_t n_src, uint32_t n_tgt,
                                     const double *distance, double default_distance,
                                     double fold_start, uint32_t k, int64_t *out_idx, double *out_key)
{
    ssym_oracle_cand *c = (ssym_oracle_cand *)malloc(sizeof(ssym_oracle_cand) * (n_src ? n_src : 1));
    if (!c)
        return -1;
    for (uint32_t t = 0; t < n_tgt; ++t) {
        const double d = distance ? distance[t] : default_distance;
        uint32_t n = 0;
        for (uint32_t s = 0; s < n_src; ++s) {
            const double key = fabs(values[(size_t)s * n_tgt + t] - d);
            if (key < fold_start) {
                c[n].key = key;
                c[n].idx = (int64_t)s;
                ++n;
            }
        }
        qsort(c, n, sizeof(ssym_oracle_cand), ssym_oracle_cand_cmp);
        for (uint32_t r = 0; r < k; ++r) {
            out_idx[(size_t)t * k + r] = r < n ? c[r].idx : -1;
            out_key[(size_t)t * k + r] = r < n ? c[r].key : NAN;
        }
    }
    free(c);
    return 0;
}

/* ---- feature front-end ---------------------------------------------
 * CPU restatement of soundsym_amd/csrc/mfcc.hip's definition.  PARITY UNPINNED against the
 * reference: analyze_mfccs (src/sound.rs:215-242) delegates to vox_box's MFCC (git HEAD,
 * Cargo.toml:9) and sample 0.9.1's Windower, neither of which is under /root/reference, and no
 * reference test pins a value.  What the reference does fix is followed: 1024-sample Hanning
 * windows hopped by 256 (src/lib.rs:24-25, src/sound.rs:228-229), 12 coefficients between 100 and
 * 8000 Hz (src/sound.rs:218), frame-major output (src/sound.rs:236-240).
 *   T = (n - 1024) / 256 + 1 full windows (pad_tail: n / 256, zeros past the end);
 *   w[i] = 0.5 - 0.5 cos(2 pi i / 1024); radix-2 DIT FFT; P[k] = re^2 + im^2;
 *   NF = 2 nc + 2 triangular filters equally spaced in mel = 1127 ln(1 + f / 700) between f_lo and
 *   min(f_hi, rate / 2); E[m] = sum_k W[m][k] P[k]; L[m] = ln(max(E[m], 1e-30));
 *   c[j] = sum_m L[m] cos(pi j (m + 1/2) / NF), j = 1..nc. */
static double ssym_mel_of(double f) { return 1127.0 * log(1.0 + f / 700.0); }
static double ssym_hz_of(double m) { return 700.0 * (exp(m / 1127.0) - 1.0); }

SSYM_ORACLE_API uint64_t ssym_oracle_mfcc_num_frames(uint64_t n, int pad_tail)
{
    if (pad_tail)
        return n / 256;
    return n >= 1024 ? (n - 1024) / 256 + 1 : 0;
}

SSYM_ORACLE_API int ssym_oracle_mfcc(const double *samples, uint64_t n, double rate, uint32_t nc,
                                     double f_lo, double f_hi, int pad_tail, double *out)
{
    enum { BIN = 1024, HOP = 256, SPEC = 513 };
    const double PI = 3.14159265358979323846;
    const int nf = 2 * (int)nc + 2;
    const uint64_t T = ssym_oracle_mfcc_num_frames(n, pad_tail);
    double *win = (double *)malloc(sizeof(double) * (BIN + BIN + (size_t)nf * SPEC + (size_t)nc * nf + 2 * BIN + nf));
    if (!win)
        return -1;
    double *twr = win + BIN, *twi = twr + BIN / 2, *W = twi + BIN / 2, *D = W + (size_t)nf * SPEC;
    double *re = D + (size_t)nc * nf, *im = re + BIN, *L = im + BIN;
    for (int i = 0; i < BIN; ++i)
        win[i] = 0.5 - 0.5 * cos(2.0 * PI * (double)i / (double)BIN);
    for (int k = 0; k < BIN / 2; ++k) {
        twr[k] = cos(-2.0 * PI * (double)k / (double)BIN);
        twi[k] = sin(-2.0 * PI * (double)k / (double)BIN);
    }
    {
        const double top = f_hi < 0.5 * rate ? f_hi : 0.5 * rate;
        const double m0 = ssym_mel_of(f_lo), m1 = ssym_mel_of(top);
        for (int m = 0; m < nf; ++m) {
            const double h0 = ssym_hz_of(m0 + (m1 - m0) * (double)m / (double)(nf + 1));
            const double h1 = ssym_hz_of(m0 + (m1 - m0) * (double)(m + 1) / (double)(nf + 1));
            const double h2 = ssym_hz_of(m0 + (m1 - m0) * (double)(m + 2) / (double)(nf + 1));
            for (int k = 0; k < SPEC; ++k) {
                const double f = (double)k * rate / (double)BIN;
                double w = 0.0;
                if (f > h0 && f <= h1)
                    w = (f - h0) / (h1 - h0);
                else if (f > h1 && f < h2)
                    w = (h2 - f) / (h2 - h1);
                W[(size_t)m * SPEC + k] = w;
            }
        }
    }
    for (uint32_t j = 0; j < nc; ++j)
        for (int m = 0; m < nf; ++m)
            D[(size_t)j * nf + m] = cos(PI * (double)(j + 1) * ((double)m + 0.5) / (double)nf);

    for (uint64_t t = 0; t < T; ++t) {
        for (int i = 0; i < BIN; ++i) {
            uint32_t r = 0;
            for (int b = 0; b < 10; ++b)
                r |= (uint32_t)((i >> b) & 1) << (9 - b);
            const uint64_t g = t * HOP + (uint64_t)i;
            re[r] = (g < n ? samples[g] : 0.0) * win[i];
            im[r] = 0.0;
        }
        for (int s = 1; s <= 10; ++s) {
            const int half = 1 << (s - 1);
            for (int b = 0; b < BIN / 2; ++b) {
                const int j = b & (half - 1);
                const int i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + half;
                const int k = j << (10 - s);
                const double tr = twr[k] * re[i1] - twi[k] * im[i1];
                const double ti = twr[k] * im[i1] + twi[k] * re[i1];
                const double ar = re[i0], ai = im[i0];
                re[i1] = ar - tr;
                im[i1] = ai - ti;
                re[i0] = ar + tr;
                im[i0] = ai + ti;
            }
        }
        for (int k = 0; k < SPEC; ++k)
            re[k] = re[k] * re[k] + im[k] * im[k];
        for (int m = 0; m < nf; ++m) {
            double e = 0.0;
            for (int k = 0; k < SPEC; ++k)
                if (W[(size_t)m * SPEC + k] != 0.0)
                    e = e + W[(size_t)m * SPEC + k] * re[k];
            L[m] = log(e > 1e-30 ? e : 1e-30);
        }
        for (uint32_t j = 0; j < nc; ++j) {
            double c = 0.0;
            for (int m = 0; m < nf; ++m)
                c = c + L[m] * D[(size_t)j * nf + m];
            out[t * nc + j] = c;
        }
    }
    free(win);
    return 0;
}

SSYM_ORACLE_API int ssym_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
