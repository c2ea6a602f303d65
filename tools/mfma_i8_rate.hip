// tools/mfma_i8_rate.hip -- cycles per v_mfma_i32_32x32x32_i8 when a wave issues them back to back on 12 accumulators of
// 16 registers (the shape of refcos_q8_kernel's chunk: 24 MFMAs, each accumulator reused after 4 others at the least), with
// one and with two waves per SIMD, accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1) or AGPRs.
// hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] -o mfma_i8_rate tools/mfma_i8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int THREADS, bool CHAINED>
__global__ __launch_bounds__(THREADS, 1) void k(int iters, int *out, v4i a0, v4i b0)
{
    v16i acc[3][2][2];
    for (int l = 0; l < 3; ++l)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int g = 0; g < 16; ++g)
                    acc[l][a][b][g] = 0;
    v4i av[3][2], bv[3][2];
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 2; ++q) {
            av[p][q] = a0 + (int)threadIdx.x + p + q;
            bv[p][q] = b0 - (int)threadIdx.x + p - q;
        }
    for (int it = 0; it < iters; ++it) {
        if (CHAINED) {           // the order the compiler's scheduler gave the kernel: an accumulator's updates back to back
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[0][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0][a], bv[0][b], acc[0][a][b], 0, 0, 0);
                    acc[1][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0][a], bv[1][b], acc[1][a][b], 0, 0, 0);
                    acc[1][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[1][a], bv[0][b], acc[1][a][b], 0, 0, 0);
                    acc[2][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0][a], bv[2][b], acc[2][a][b], 0, 0, 0);
                    acc[2][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[1][a], bv[1][b], acc[2][a][b], 0, 0, 0);
                    acc[2][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[2][a], bv[0][b], acc[2][a][b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            continue;
        }
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb + pa < 3; ++pb)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[pa + pb][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[pa][a], bv[pb][b], acc[pa + pb][a][b], 0, 0, 0);
    }
    int x = 0;
    for (int l = 0; l < 3; ++l)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int g = 0; g < 16; ++g)
                    x ^= acc[l][a][b][g];
    if (x == 0x12345678)
        out[0] = x;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    v4i a0 = {1, 2, 3, 4}, b0 = {5, 6, 7, 8};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<256, false><<<p.multiProcessorCount, 256>>>(iters, out, a0, b0);
            if (mode == 1) k<512, false><<<p.multiProcessorCount, 512>>>(iters, out, a0, b0);
            if (mode == 2) k<256, true><<<p.multiProcessorCount, 256>>>(iters, out, a0, b0);
            if (mode == 3) k<512, true><<<p.multiProcessorCount, 512>>>(iters, out, a0, b0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double mfmaPerSimd = (double)iters * 24 * ((mode & 1) == 0 ? 1 : 2);
        printf("%s%d wave(s) per SIMD: %.3f ms, %.1f ns per MFMA and SIMD = %.1f cycles at 2.4 GHz; %.0f TOP/s over %d CUs\n", mode >= 2 ? "chained updates, " : "", (mode & 1) + 1, ms,
               ms * 1e6 / mfmaPerSimd, ms * 1e6 / mfmaPerSimd * 2.4, mfmaPerSimd * 4 * p.multiProcessorCount * 65536.0 / (ms * 1e-3) / 1e12,
               p.multiProcessorCount);
    }
    return 0;
}
