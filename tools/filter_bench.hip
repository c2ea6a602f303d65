// tools/filter_bench.hip -- standalone timing of the dtw filter kernel (development tool, not
// product): random records, HIP-event timing, optional ablation via -DSSYM_FILTER_MODE=1|2.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Isoundsym_amd/csrc \
//         tools/filter_bench.hip -o tools/filter_bench
#include "dtw_filter_kernel.hpp"

#ifndef SSYM_TOOL_SQ
#define SSYM_TOOL_SQ false
#endif
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT, int OCC>
int run(int N, int M, int F, int reps)
{
    using namespace ssym;
    constexpr int REC = kFilterRecHalfs;
    const int nPasses = (F + 16 * NT - 1) / (16 * NT);
    const int rows = 16 * NT * nPasses;
    std::vector<_Float16> hs((size_t)N * rows * REC), ht((size_t)M * F * REC);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : hs) v = (_Float16)(4.0f * rnd());
    for (auto &v : ht) v = (_Float16)(4.0f * rnd());
    std::vector<int> ls(N, F), lt(M, F);
    _Float16 *ds, *dt; float *dc, *dc2, *dh; int *dls, *dlt; unsigned *dctr;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocksPerCU = OCC;
    const int grid = prop.multiProcessorCount * blocksPerCU / 8 * 8;
    CK(hipMalloc(&ds, hs.size() * 2)); CK(hipMalloc(&dt, ht.size() * 2));
    CK(hipMalloc(&dc, (size_t)N * M * 4)); CK(hipMalloc(&dc2, (size_t)N * M * 4)); CK(hipMalloc(&dls, N * 4)); CK(hipMalloc(&dlt, M * 4));
    CK(hipMalloc(&dh, (size_t)grid * kFilterWavesPerBlock * F * 64 * 4));
    CK(hipMalloc(&dctr, 8 * kTaskCtrStride * sizeof(unsigned)));
    CK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dls, ls.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dlt, lt.data(), M * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nSrcPairs = N / 2;
    const int nTasks = nSrcPairs * (M / 32);
    int taskChunk = std::max(1, std::min(8, 8192 / (rows * F)));
    taskChunk = std::max(1, std::min(taskChunk, nTasks / (grid * kFilterWavesPerBlock * 16)));
    auto launch = [&]() {
        (void)hipMemsetAsync(dctr, 0, 8 * kTaskCtrStride * sizeof(unsigned), 0);
        dtw_filter_kernel<NT, SSYM_TOOL_SQ, OCC><<<grid, 64 * kFilterWavesPerBlock>>>(ds, dt, dls, dlt, rows, nPasses, F, M, nSrcPairs,
                                                                         nTasks, taskChunk, 1.0f, dh, dctr, dc);
    };
    for (int w = 0; w < 2; ++w) launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    double pairs = (double)N * M;
    double cells = pairs * rows * F;
    printf("mode %d sq %d NT %d passes %d F %d N %d M %d blocks/CU %d: best %.3f ms avg %.3f -> %.3e pairs/s, %.3e cells/s (padded rows %d), %.3f ns*SIMD/cell\n",
           SSYM_FILTER_MODE, (int)SSYM_TOOL_SQ, NT, nPasses, F, N, M, blocksPerCU, best, sum / reps, pairs / (best * 1e-3),
           cells / (best * 1e-3), rows, best * 1e6 * 1024.0 / cells);
    (void)hipFree(ds); (void)hipFree(dt); (void)hipFree(dc); (void)hipFree(dls); (void)hipFree(dlt); (void)hipFree(dh);
    return 0;
}

int main(int argc, char **argv)
{
    int N = argc > 1 ? atoi(argv[1]) : 2048, M = argc > 2 ? atoi(argv[2]) : 2048;
    int reps = argc > 3 ? atoi(argv[3]) : 5;
    int only = argc > 4 ? atoi(argv[4]) : -1;   // run a single variant (for rocprofv3 --pmc)
    int v = 0;
#define VARIANT(NT, F, BPC) do { if (only < 0 || only == v) { if (run<NT, BPC>(N, M, F, reps)) return 1; } ++v; } while (0)
    VARIANT(4, 128, 2);     // 0: BASELINE shape: 128 frames = 2 passes of 64 rows
    VARIANT(4, 64, 2);      // 1: one pass
    VARIANT(3, 128, 2);     // 2: 3 passes of 48 rows (144 padded)
    VARIANT(2, 128, 2);     // 3: (was three workgroups per CU; multi-pass shapes need the ring of 4)
    VARIANT(2, 128, 2);     // 4: same, 2 waves/SIMD
    VARIANT(4, 256, 2);     // 5: 256 frames = 4 passes
    VARIANT(3, 48, 2);      // 6: 48 frames, one pass
    VARIANT(2, 32, 2);      // 7: 32 frames, one pass
    VARIANT(1, 16, 2);      // 8: 16 frames, one pass
    VARIANT(2, 32, 3);      // 9: 32 frames, three waves per SIMD, ring of 2
    VARIANT(1, 16, 3);      // 10: 16 frames, three waves per SIMD
    return 0;
}
