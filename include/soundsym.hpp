// soundsym.hpp -- C++ host mirror of the reference's matching interface over the C ABI.
//
// The reference is a compiled (Rust) crate and this image has no Rust toolchain, so the host side
// above `include/soundsym_amd.h` is C++ with the reference's names, argument meaning and error
// behaviour for the hot path (upstream src/sound.rs):
//
//   Sound::from_samples / samples / sample_rate / mfccs / num_frames     :92, :181-212
//   SoundDictionary::new_ / from_segments / add_segments                 :296, :323, :330
//   SoundDictionary::match_sound / at_distance                           :346, :351
//   SoundSequence::new_ / sounds / from_distances / morph_to /
//                  clone_from_dictionary / to_sound                      :392-483
//
// `Arc<Sound>` is `std::shared_ptr<const Sound>`; `Option<Arc<Sound>>` is returned as the pointer
// itself (the reference never returns None, :369); `Result<_, String>` becomes an exception of
// type soundsym::Error; the reference's panic on an empty dictionary (:369) becomes
// soundsym::EmptyDictionary.  Feature extraction (:215-242) is out of scope: a Sound is built from
// samples plus ready-made features (the `Some(mfccs)` form of from_samples, :92-94).
//
// Header-only; link libsoundsym_amd.so and one HIP runtime.
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "soundsym_amd.h"

namespace soundsym {

constexpr std::size_t NCOEFFS = 12;   // src/lib.rs:22
constexpr std::size_t HOP = 256;      // src/lib.rs:24
constexpr std::size_t BIN = 1024;     // src/lib.rs:25

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct EmptyDictionary : Error {
    EmptyDictionary() : Error(SSYM_E_EMPTY_DICT, "empty dictionary (the reference panics here, src/sound.rs:369)") {}
};

// One GPU context; metric 0 = refcos (the crate's own cosine_sim), 1 = dtw.
class Context {
  public:
    // prune: dtw nearest-neighbour searches of 64 targets and more abandon pairs early (same results,
    // data-dependent time)
    explicit Context(int metric = SSYM_METRIC_REFCOS, int device = 0, int band = -1, bool squared = false,
                     bool prune = false)
    {
        ssym_config cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.device = device;
        cfg.metric = metric;
        cfg.dtype = SSYM_DTYPE_F64;   // Sound::mfccs() is Vec<f64>
        cfg.band = band;
        cfg.dtw_squared = squared ? 1 : 0;
        cfg.stream = nullptr;
        cfg.dtw_prune = prune ? 1 : 0;
        int rc = ssym_ctx_create(&cfg, &ctx_);
        if (rc != SSYM_OK)
            throw Error(rc, ssym_last_error(nullptr));
        metric_ = metric;
    }
    ~Context() { ssym_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ssym_ctx *get() const { return ctx_; }
    int metric() const { return metric_; }
    void check(int rc) const
    {
        if (rc == SSYM_E_EMPTY_DICT)
            throw EmptyDictionary();
        if (rc != SSYM_OK)
            throw Error(rc, ssym_last_error(ctx_));
    }

  private:
    ssym_ctx *ctx_ = nullptr;
    int metric_ = 0;
};

class Sound {
  public:
    std::optional<std::string> name;

    // Sound::from_samples(samples, sample_rate, mfccs, name), src/sound.rs:92
    static Sound from_samples(std::vector<double> samples, double sample_rate,
                              std::optional<std::vector<double>> mfccs, std::optional<std::string> name)
    {
        Sound s;
        s.samples_ = std::move(samples);
        s.sample_rate_ = sample_rate;
        if (mfccs) {
            if (mfccs->size() % NCOEFFS)
                throw Error(SSYM_E_INVALID, "mfccs must hold whole frames of NCOEFFS values");
            s.mfccs_ = std::move(*mfccs);
            s.has_mfccs_ = true;
        }
        s.name = std::move(name);
        return s;
    }
    // The `None` form of from_samples (src/sound.rs:97-101): run the MFCC analysis (analyze_mfccs,
    // :215-242) -- on the GPU, ssym_mfcc; its arithmetic is this library's own definition (parity
    // with the un-vendored vox_box MFCC is unpinned).
    template <class Ctx>
    static Sound from_samples(Ctx &ctx, std::vector<double> samples, double sample_rate,
                              std::optional<std::string> name = std::nullopt)
    {
        uint64_t frames = 0;
        ssym_mfcc_num_frames(samples.size(), 0, &frames);
        std::vector<double> m(frames * NCOEFFS);
        ctx.check(ssym_mfcc(ctx.get(), samples.data(), samples.size(), sample_rate, (uint32_t)NCOEFFS, 100.0,
                            8000.0, 0, m.data(), nullptr));
        return from_samples(std::move(samples), sample_rate, std::move(m), std::move(name));
    }
    const std::vector<double> &samples() const { return samples_; }   // :181
    double sample_rate() const { return sample_rate_; }                // :185
    const std::vector<double> &mfccs() const                           // :191
    {
        if (!has_mfccs_)
            throw Error(SSYM_E_INVALID, "this Sound carries no features (use the from_samples overload that analyses)");
        return mfccs_;
    }
    bool has_mfccs() const { return has_mfccs_; }
    std::size_t num_frames() const { return mfccs().size() / NCOEFFS; }   // :210

  private:
    std::vector<double> samples_, mfccs_;
    double sample_rate_ = 44100.0;
    bool has_mfccs_ = false;
};

using ArcSound = std::shared_ptr<const Sound>;

inline void pack_features(const std::vector<ArcSound> &sounds, std::vector<double> &flat,
                          std::vector<uint64_t> &off)
{
    off.assign(1, 0);
    flat.clear();
    for (const auto &s : sounds) {
        const auto &m = s->mfccs();
        flat.insert(flat.end(), m.begin(), m.end());
        off.push_back(off.back() + m.size() / NCOEFFS);
    }
}

class SoundDictionary {
  public:
    std::vector<ArcSound> sounds;   // `pub sounds: Vec<Arc<Sound>>`, src/sound.rs:291

    explicit SoundDictionary(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {}
    ~SoundDictionary() { release(); }
    SoundDictionary(const SoundDictionary &) = delete;
    SoundDictionary &operator=(const SoundDictionary &) = delete;

    static std::unique_ptr<SoundDictionary> new_(std::shared_ptr<Context> ctx)   // :296
    {
        return std::make_unique<SoundDictionary>(std::move(ctx));
    }
    static std::unique_ptr<SoundDictionary> from_segments(std::shared_ptr<Context> ctx, const Sound &sound,
                                                          const std::vector<std::size_t> &segments)   // :323
    {
        auto d = new_(std::move(ctx));
        d->add_segments(sound, segments);
        return d;
    }
    // src/sound.rs:330-343: `seg` samples and `seg / HOP * NCOEFFS` feature values per segment,
    // consumed in order from the parent sound (`take` on an exhausted iterator yields what is left)
    void add_segments(const Sound &sound, const std::vector<std::size_t> &segments)
    {
        const auto &samples = sound.samples();
        const auto &mfccs = sound.mfccs();
        std::size_t spos = 0, mpos = 0;
        for (std::size_t seg : segments) {
            std::size_t se = std::min(spos + seg, samples.size());
            std::size_t nm = seg / HOP * NCOEFFS, me = std::min(mpos + nm, mfccs.size());
            std::vector<double> samp(samples.begin() + spos, samples.begin() + se);
            std::vector<double> mf(mfccs.begin() + mpos, mfccs.begin() + me);
            spos = se;
            mpos = me;
            sounds.push_back(std::make_shared<const Sound>(
                Sound::from_samples(std::move(samp), sound.sample_rate(), std::move(mf), std::nullopt)));
        }
    }

    ArcSound match_sound(const Sound &other) const   // :346  at_distance(1., other)
    {
        return at_distance(ctx_->metric() == SSYM_METRIC_REFCOS ? 1.0 : 0.0, other);
    }
    ArcSound at_distance(double distance, const Sound &other) const   // :351
    {
        if (sounds.empty())
            throw EmptyDictionary();
        uint32_t idx = 0;
        double val = 0.0;
        ctx_->check(ssym_match_one(ctx_->get(), resident(), other.mfccs().data(), other.num_frames(), distance,
                                   &idx, &val));
        return sounds[idx];   // Some(self.sounds[min_idx].clone())
    }
    // batched form of the loops at :442-446 and :453-454
    std::vector<uint32_t> match_indices(const std::vector<ArcSound> &targets, const double *distances) const
    {
        if (sounds.empty())
            throw EmptyDictionary();
        std::vector<double> flat;
        std::vector<uint64_t> off;
        pack_features(targets, flat, off);
        std::vector<uint32_t> idx(targets.size());
        ctx_->check(ssym_match_batch(ctx_->get(), resident(), flat.data(), off.data(), (uint32_t)targets.size(),
                                     distances, idx.data(), nullptr));
        return idx;
    }

    // the k best sounds per target, best first (ssym_match_topk; the crate has no counterpart:
    // at_distance keeps only the first minimum, :361-367)
    std::vector<std::vector<ArcSound>> candidates(const std::vector<ArcSound> &targets, uint32_t k,
                                                  const double *distances = nullptr) const
    {
        if (sounds.empty())
            throw EmptyDictionary();
        std::vector<double> flat;
        std::vector<uint64_t> off;
        pack_features(targets, flat, off);
        ssym_queries *q = nullptr;
        ctx_->check(ssym_queries_create(ctx_->get(), flat.data(), off.data(), (uint32_t)targets.size(),
                                        (uint32_t)NCOEFFS, &q));
        std::vector<uint32_t> idx(targets.size() * k);
        int32_t rc = ssym_match_topk(ctx_->get(), resident(), q, distances, k, 0, idx.data(), nullptr, 0);
        ssym_queries_destroy(ctx_->get(), q);
        ctx_->check(rc);
        std::vector<std::vector<ArcSound>> out(targets.size());
        for (std::size_t t = 0; t < targets.size(); ++t)
            for (uint32_t r = 0; r < k; ++r)
                if (idx[t * k + r] != SSYM_NO_MATCH)
                    out[t].push_back(sounds[idx[t * k + r]]);
        return out;
    }
    // from_distances' chain of at_distance calls, on the device in one call (ssym_chain)
    std::vector<uint32_t> chain_indices(const Sound &start, const std::vector<double> &distances) const
    {
        if (sounds.empty())
            throw EmptyDictionary();
        std::vector<uint32_t> idx(distances.size());
        ctx_->check(ssym_chain(ctx_->get(), const_cast<ssym_dict *>(resident()), start.mfccs().data(),
                               start.num_frames(), distances.data(), (uint32_t)distances.size(), idx.data(),
                               nullptr));
        return idx;
    }

  private:
    // the dictionary's features are packed once per content change, not per query
    const ssym_dict *resident() const
    {
        // `sounds` is public and mutable (`pub sounds`): entries may have been replaced or reordered in place, so the
        // packed copy is compared by content (the Arcs themselves), not by length; the copy keeps them alive
        if (!dict_ || packed_ != sounds) {
            release();
            std::vector<double> flat;
            std::vector<uint64_t> off;
            pack_features(sounds, flat, off);
            ctx_->check(ssym_dict_create(ctx_->get(), flat.data(), off.data(), (uint32_t)sounds.size(),
                                         (uint32_t)NCOEFFS, &dict_));
            packed_ = sounds;
        }
        return dict_;
    }
    void release() const
    {
        if (dict_)
            ssym_dict_destroy(ctx_->get(), dict_);
        dict_ = nullptr;
    }
    std::shared_ptr<Context> ctx_;
    mutable ssym_dict *dict_ = nullptr;
    mutable std::vector<ArcSound> packed_;
};

// One rank of a SoundDictionary split over the GPUs of one node (source-axis shards, one rank -- thread or process --
// per GPU): rank g holds sounds [lo, lo + shard.size()) of the whole dictionary and is handed the same targets as every
// other rank; match_indices returns GLOBAL indices, the same complete answer on every rank, bit for bit the unsharded
// one.  The collectives (RCCL all-reduce(MIN) of the per-target bounds, all-gather of (cost, index)) run inside the
// library on the context's stream (ssym_match_sharded): the loop of clone_from_dictionary / morph_to,
// src/sound.rs:451-455 / 440-446, for a dictionary too large or too slow for one GPU.
class ShardedDictionary {
  public:
    using Id = std::vector<unsigned char>;
    // rank 0 draws the id (ncclGetUniqueId) and hands it to the other ranks by whatever means the host has
    static Id unique_id()
    {
        Id id(SSYM_COMM_ID_BYTES);
        int rc = ssym_comm_unique_id(id.data());
        if (rc != SSYM_OK)
            throw Error(rc, "ssym_comm_unique_id failed (is RCCL available?)");
        return id;
    }
    // collective over all `world` ranks (ncclCommInitRank)
    ShardedDictionary(std::shared_ptr<Context> ctx, const Id &id, int rank, int world, std::vector<ArcSound> shard,
                      uint32_t lo)
        : ctx_(std::move(ctx)), shard_(std::move(shard)), lo_(lo)
    {
        ctx_->check(ssym_comm_create(ctx_->get(), id.data(), rank, world, &comm_));
        std::vector<double> flat;
        std::vector<uint64_t> off;
        pack_features(shard_, flat, off);
        int rc = ssym_dict_create(ctx_->get(), flat.data(), off.data(), (uint32_t)shard_.size(), (uint32_t)NCOEFFS, &dict_);
        if (rc != SSYM_OK) {
            ssym_comm_destroy(ctx_->get(), comm_);
            ctx_->check(rc);
        }
    }
    ~ShardedDictionary()
    {
        ssym_dict_destroy(ctx_->get(), dict_);
        ssym_comm_destroy(ctx_->get(), comm_);
    }
    ShardedDictionary(const ShardedDictionary &) = delete;
    ShardedDictionary &operator=(const ShardedDictionary &) = delete;

    std::vector<uint32_t> match_indices(const std::vector<ArcSound> &targets, const double *distances) const
    {
        std::vector<double> flat;
        std::vector<uint64_t> off;
        pack_features(targets, flat, off);
        ssym_queries *q = nullptr;
        ctx_->check(ssym_queries_create(ctx_->get(), flat.data(), off.data(), (uint32_t)targets.size(),
                                        (uint32_t)NCOEFFS, &q));
        std::vector<uint32_t> idx(targets.size());
        int32_t rc = ssym_match_sharded(ctx_->get(), comm_, dict_, q, distances, lo_, idx.data(), nullptr, 0);
        ssym_queries_destroy(ctx_->get(), q);
        ctx_->check(rc);
        return idx;
    }

  private:
    std::shared_ptr<Context> ctx_;
    std::vector<ArcSound> shard_;
    uint32_t lo_;
    ssym_comm *comm_ = nullptr;
    ssym_dict *dict_ = nullptr;
};

class SoundSequence {
  public:
    static SoundSequence new_(std::vector<ArcSound> sounds)   // :392
    {
        SoundSequence s;
        s.sounds_ = std::move(sounds);
        return s;
    }
    const std::vector<ArcSound> &sounds() const { return sounds_; }   // :432

    // :405-417 greedy chain: each step's query is the previous result; the steps run on the device
    // back to back (one call, one wait)
    static SoundSequence from_distances(const std::vector<double> &distances, ArcSound start,
                                        const SoundDictionary &dict)
    {
        std::vector<ArcSound> sounds{start};
        if (!distances.empty())
            for (uint32_t i : dict.chain_indices(*start, distances))
                sounds.push_back(dict.sounds[i]);
        return new_(std::move(sounds));
    }
    // :440-449 zip(sounds, distances) -> at_distance, as ONE batch
    SoundSequence morph_to(const std::vector<double> &distances, const SoundDictionary &dict) const
    {
        std::size_t n = std::min(sounds_.size(), distances.size());
        std::vector<ArcSound> q(sounds_.begin(), sounds_.begin() + n), out;
        if (n == 0)
            return new_({});
        for (uint32_t i : dict.match_indices(q, distances.data()))
            out.push_back(dict.sounds[i]);
        return new_(std::move(out));
    }
    // :451-472 match every sound, then fit the match to the target's length (:456-465)
    SoundSequence clone_from_dictionary(const SoundDictionary &dict) const
    {
        if (sounds_.empty())
            return new_({});
        std::vector<ArcSound> out;
        auto idx = dict.match_indices(sounds_, nullptr);
        for (std::size_t k = 0; k < sounds_.size(); ++k) {
            const ArcSound &sound = sounds_[k], &s = dict.sounds[idx[k]];
            if (sound->samples().size() == s->samples().size()) {
                out.push_back(s);   // :463-464 shares the Arc
            } else {
                std::vector<double> samps(sound->samples().size(), 0.0);   // :457-462
                std::copy_n(s->samples().begin(), std::min(samps.size(), s->samples().size()), samps.begin());
                out.push_back(std::make_shared<const Sound>(
                    Sound::from_samples(std::move(samps), sound->sample_rate(), std::nullopt, std::nullopt)));
            }
        }
        return new_(std::move(out));
    }
    Sound to_sound() const   // :475-483
    {
        std::vector<double> samples;
        for (const auto &s : sounds_)
            samples.insert(samples.end(), s->samples().begin(), s->samples().end());
        double rate = sounds_.empty() ? 44100.0 : sounds_[0]->sample_rate();
        return Sound::from_samples(std::move(samples), rate, std::nullopt, std::nullopt);
    }

  private:
    std::vector<ArcSound> sounds_;
};

}  // namespace soundsym
