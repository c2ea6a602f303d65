// tools/latency_one.cpp -- ssym_match_one from a C++ caller (no Python in the way): the latency a Rust caller would see.
//   g++ -O2 -std=c++17 -I include tools/latency_one.cpp -o tools/latency_one -L soundsym_amd -lsoundsym_amd -L/opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,$PWD/soundsym_amd -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "soundsym_amd.h"

static double rnd(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0 - 0.5;
}

int main()
{
    for (int metric = 0; metric < 2; ++metric) {
        ssym_config cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.metric = metric == 0 ? SSYM_METRIC_REFCOS : SSYM_METRIC_DTW;
        cfg.dtype = SSYM_DTYPE_F64;
        cfg.band = -1;
        ssym_ctx *ctx = nullptr;
        if (ssym_ctx_create(&cfg, &ctx) != SSYM_OK) {
            std::printf("ctx: %s\n", ssym_last_error(nullptr));
            return 1;
        }
        const uint32_t n = 1024, dim = 12;
        unsigned long long seed = 12345;
        std::vector<uint64_t> off(n + 1, 0);
        for (uint32_t i = 0; i < n; ++i)
            off[i + 1] = off[i] + 5 + (uint64_t)((rnd(seed) + 0.5) * 35);
        std::vector<double> feats(off[n] * dim);
        for (auto &v : feats)
            v = 0.2 * rnd(seed);
        ssym_dict *dict = nullptr;
        if (ssym_dict_create(ctx, feats.data(), off.data(), n, dim, &dict) != SSYM_OK)
            return 2;
        std::vector<double> q(24 * dim);
        for (auto &v : q)
            v = 0.2 * rnd(seed);
        uint32_t idx = 0;
        double val = 0;
        for (int w = 0; w < 10; ++w)
            ssym_match_one(ctx, dict, q.data(), 24, metric == 0 ? 1.0 : 0.0, &idx, &val);
        const int reps = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            ssym_match_one(ctx, dict, q.data(), 24, metric == 0 ? 1.0 : 0.0, &idx, &val);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        ssym_timings tm{};
        ssym_get_timings(ctx, &tm);
        std::printf("%s: ssym_match_one on %u entries: %.1f us per call (device %.1f us, pack %.1f us)\n",
                    metric == 0 ? "refcos" : "dtw", n, us, tm.total_ms * 1e3, tm.pack_ms * 1e3);
        ssym_dict_destroy(ctx, dict);
        ssym_ctx_destroy(ctx);
    }
    return 0;
}
