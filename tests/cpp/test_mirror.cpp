// tests/cpp/test_mirror.cpp -- the reference's own matching tests, restated against the C++ mirror
// (include/soundsym.hpp) and checked against the CPU oracle (oracle/ is test infrastructure).
//
//   test_sound_should_match_itself     src/sound.rs:600-609
//   clone_from_dictionary / morph_to / from_distances / to_sound    src/sound.rs:405-483
//   empty dictionary                    the panic at src/sound.rs:369
//
// Build: tests/cpp/Makefile.  Exit code 0 = all checks passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "soundsym.hpp"

extern "C" {
int64_t ssym_oracle_at_distance(const double *feats, const uint64_t *off, uint32_t n_src, uint32_t dim,
                                double distance, const double *you, uint64_t you_frames, double *out_min);
int ssym_oracle_dtw_match_all(const double *src, const uint64_t *src_off, uint32_t n_src, const double *tgt,
                              const uint64_t *tgt_off, uint32_t n_tgt, uint32_t dim, int64_t band, int squared,
                              int nthreads, int64_t *out_idx, double *out_cost, double *cost_matrix);
void ssym_oracle_length_fit(const double *matched, uint64_t n_matched, uint64_t n_target, double *out);
int ssym_oracle_mfcc(const double *samples, uint64_t n, double rate, uint32_t nc, double f_lo, double f_hi,
                     int pad_tail, double *out);
}

using namespace soundsym;

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)

static uint64_t g_state = 0x5EED0500;
static double rnd()   // splitmix64 -> uniform in (-0.5, 0.5)
{
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0 - 0.5;
}

static Sound parent(std::size_t frames)
{
    std::vector<double> samples(frames * HOP), mfccs(frames * NCOEFFS);
    for (auto &v : samples) v = 0.2 * rnd();
    for (auto &v : mfccs) v = 0.2 * rnd();
    return Sound::from_samples(std::move(samples), 44100.0, std::move(mfccs), std::nullopt);
}

static int64_t oracle_at_distance(const SoundDictionary &d, double distance, const Sound &q)
{
    std::vector<double> flat;
    std::vector<uint64_t> off;
    pack_features(d.sounds, flat, off);
    double v;
    return ssym_oracle_at_distance(flat.data(), off.data(), (uint32_t)d.sounds.size(), NCOEFFS, distance,
                                   q.mfccs().data(), q.num_frames(), &v);
}

int main()
{
    auto ctx = std::make_shared<Context>(SSYM_METRIC_REFCOS);
    std::vector<std::size_t> segs;
    for (int k : {3, 5, 2, 7, 4, 6, 3, 8, 5, 4, 6, 2})
        segs.push_back(HOP * k);
    Sound src = parent(64);
    auto dict = SoundDictionary::from_segments(ctx, src, segs);
    CHECK(dict->sounds.size() == segs.size());
    CHECK(dict->sounds[3]->num_frames() == 7 && dict->sounds[3]->samples().size() == 7 * HOP);

    // test_sound_should_match_itself (src/sound.rs:600-609); the oracle decides what "match" is
    {
        const auto &sound = dict->sounds[4];
        ArcSound got = dict->match_sound(*sound);
        CHECK(got == dict->sounds[oracle_at_distance(*dict, 1.0, *sound)]);
    }
    // clone_from_dictionary: indices from the oracle, samples fitted as src/sound.rs:456-465
    Sound tsrc = parent(40);
    auto tdict = SoundDictionary::from_segments(ctx, tsrc, {HOP * 5, HOP * 5, HOP * 7, HOP * 3, HOP * 6, HOP * 4});
    SoundSequence seq = SoundSequence::new_(tdict->sounds);
    SoundSequence out = seq.clone_from_dictionary(*dict);
    CHECK(out.sounds().size() == tdict->sounds.size());
    std::size_t total = 0;
    for (std::size_t k = 0; k < out.sounds().size(); ++k) {
        const Sound &t = *tdict->sounds[k];
        const Sound &w = *dict->sounds[oracle_at_distance(*dict, 1.0, t)];
        std::vector<double> want(t.samples().size());
        ssym_oracle_length_fit(w.samples().data(), w.samples().size(), want.size(), want.data());
        CHECK(out.sounds()[k]->samples() == want);
        total += want.size();
    }
    CHECK(out.to_sound().samples().size() == total);
    // morph_to: per-target distance
    std::vector<double> dist{0.0, 0.01, 0.02, 0.03, 0.04, 0.05};
    SoundSequence morphed = seq.morph_to(dist, *dict);
    for (std::size_t k = 0; k < dist.size(); ++k)
        CHECK(morphed.sounds()[k] == dict->sounds[oracle_at_distance(*dict, dist[k], *tdict->sounds[k])]);
    // from_distances: greedy chain
    SoundSequence chain = SoundSequence::from_distances({0.01, 0.02, 0.0}, dict->sounds[0], *dict);
    ArcSound cur = dict->sounds[0];
    for (std::size_t k = 0; k < 3; ++k) {
        ArcSound want = dict->sounds[oracle_at_distance(*dict, std::vector<double>{0.01, 0.02, 0.0}[k], *cur)];
        CHECK(chain.sounds()[k + 1] == want);
        cur = want;
    }
    // candidates: the first of the k best is at_distance's answer, no sound appears twice
    {
        auto cands = dict->candidates({tdict->sounds[0], tdict->sounds[2]}, 3);
        CHECK(cands.size() == 2 && cands[0].size() == 3);
        CHECK(cands[0][0] == dict->sounds[oracle_at_distance(*dict, 1.0, *tdict->sounds[0])]);
        CHECK(cands[1][0] == dict->sounds[oracle_at_distance(*dict, 1.0, *tdict->sounds[2])]);
        CHECK(cands[0][0] != cands[0][1] && cands[0][1] != cands[0][2] && cands[0][0] != cands[0][2]);
    }
    // from_samples without features analyses on the GPU: 1024-sample windows hopped by 256
    {
        std::vector<double> smp(4096);
        for (std::size_t i = 0; i < smp.size(); ++i)
            smp[i] = std::sin(0.05 * (double)i);
        Sound analysed = Sound::from_samples(*ctx, smp, 44100.0);
        CHECK(analysed.num_frames() == (4096 - 1024) / 256 + 1);
        std::vector<double> want(analysed.mfccs().size());
        ssym_oracle_mfcc(smp.data(), smp.size(), 44100.0, NCOEFFS, 100.0, 8000.0, 0, want.data());
        double worst = 0.0;
        for (std::size_t i = 0; i < want.size(); ++i)
            worst = std::max(worst, std::fabs(want[i] - analysed.mfccs()[i]) / (1.0 + std::fabs(want[i])));
        CHECK(worst <= 1e-12);
    }
    // add_segments invalidates the resident copy; indices continue
    dict->add_segments(tsrc, {HOP * 9});
    CHECK(dict->match_sound(*dict->sounds.back()) == dict->sounds[oracle_at_distance(*dict, 1.0, *dict->sounds.back())]);
    // empty dictionary: the reference panics (src/sound.rs:369)
    bool threw = false;
    try {
        SoundDictionary::new_(ctx)->match_sound(*dict->sounds[0]);
    } catch (const EmptyDictionary &) {
        threw = true;
    }
    CHECK(threw);

    // the same containers on the dtw metric
    {
        auto dctx = std::make_shared<Context>(SSYM_METRIC_DTW);
        auto dd = SoundDictionary::from_segments(dctx, src, segs);
        CHECK(dd->match_sound(*dd->sounds[6]) == dd->sounds[6]);   // DTW of a segment with itself is 0
        Sound probe = parent(6);
        std::vector<double> flat, pf = probe.mfccs();
        std::vector<uint64_t> off, po{0, probe.num_frames()};
        pack_features(dd->sounds, flat, off);
        int64_t want;
        double cost;
        ssym_oracle_dtw_match_all(flat.data(), off.data(), (uint32_t)dd->sounds.size(), pf.data(), po.data(), 1,
                                  NCOEFFS, -1, 0, 1, &want, &cost, nullptr);
        CHECK(dd->match_sound(probe) == dd->sounds[want]);

        // the same dictionary behind a context with early abandoning as its default (ssym_config.dtw_prune):
        // batches of 64 targets and more are pruned, and answer exactly as the plain context does
        auto pctx = std::make_shared<Context>(SSYM_METRIC_DTW, 0, -1, false, true);
        auto pd = SoundDictionary::from_segments(pctx, src, segs);
        std::vector<ArcSound> many;
        for (int r = 0; r < 6; ++r)
            for (const auto &snd : dd->sounds)
                many.push_back(snd);
        CHECK(many.size() >= 64);
        const std::vector<uint32_t> plain = dd->match_indices(many, nullptr), pruned = pd->match_indices(many, nullptr);
        CHECK(plain == pruned);
        ssym_timings tm{};
        CHECK(ssym_get_timings(pctx->get(), &tm) == SSYM_OK && tm.pruned == 1);
        CHECK(ssym_get_timings(dctx->get(), &tm) == SSYM_OK && tm.pruned == 0);
    }
    // an entry replaced in place (same length): the resident copy follows the content
    {
        auto d2 = SoundDictionary::from_segments(ctx, src, segs);
        ArcSound first = d2->match_sound(*d2->sounds[2]);
        const std::size_t at = (std::size_t)(std::find(d2->sounds.begin(), d2->sounds.end(), first) - d2->sounds.begin());
        d2->sounds[at] = tdict->sounds[1];
        ArcSound again = d2->match_sound(*d2->sounds[2]);
        CHECK(again == d2->sounds[oracle_at_distance(*d2, 1.0, *d2->sounds[2])]);
    }
    // a dictionary as ONE rank of a source-sharded run: RCCL communicator of world 1 behind the C ABI, global indices
    {
        auto dctx = std::make_shared<Context>(SSYM_METRIC_DTW);
        auto whole = SoundDictionary::from_segments(dctx, src, segs);
        std::vector<ArcSound> targets(tdict->sounds.begin(), tdict->sounds.end());
        const std::vector<uint32_t> want = whole->match_indices(targets, nullptr);
        ShardedDictionary rank0(dctx, ShardedDictionary::unique_id(), 0, 1, whole->sounds, 100);
        std::vector<uint32_t> got = rank0.match_indices(targets, nullptr);
        CHECK(got.size() == want.size());
        for (std::size_t i = 0; i < got.size() && i < want.size(); ++i)
            CHECK(got[i] == want[i] + 100);
        ssym_timings tm{};
        CHECK(ssym_get_timings(dctx->get(), &tm) == SSYM_OK && tm.attempts == 1);
    }
    std::printf(g_fail ? "%d checks FAILED\n" : "all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
