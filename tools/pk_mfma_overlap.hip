// tools/pk_mfma_overlap.hip -- does a packed f32 addition overlap the f16 matrix pipe the way a plain VALU addition does?
//
// One workgroup of 8 waves per CU = two waves per SIMD.  The EVEN waves of a SIMD pair run a loop of VALU additions
// (mode 0: v_add_f32, mode 1: v_pk_add_f32, mode 2: v_sqrt_f32), the ODD waves either idle (quiet = 1) or issue
// v_mfma_f32_32x32x16_f16 back to back (quiet = 0).  Reported: cycles per VALU instruction of the even waves
// (s_memtime, 100 MHz reference converted with the measured kernel time is avoided: clock64 = shader clock).
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_mfma_overlap tools/pk_mfma_overlap.hip && ./pk_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096, kUnroll = 16;

template <int MODE>
__global__ __launch_bounds__(512) void overlap_kernel(int quiet, unsigned long long *cycles, float *sink)
{
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    // waves w and w + 4 share a SIMD (round-robin placement): w < 4 adds, w >= 4 multiplies
    if (wave >= 4) {
        if (quiet)
            return;
        half8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)(0.001f * (lane + i));
            b[i] = (_Float16)(0.002f * (lane - i));
        }
        f32x16 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < kIters * 2; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
        if (acc0[0] + acc1[3] == 12345.0f)
            sink[threadIdx.x] = acc0[1];
        return;
    }
    f32x2 x[kUnroll];
    for (int i = 0; i < kUnroll; ++i)
        x[i] = f32x2{1.0f + 0.001f * (lane + i), 2.0f + 0.001f * (lane - i)};
    const f32x2 inc = {1.0009765625f, 0.9990234375f};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (MODE == 0) {
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i].x) : "v"(x[i].x), "v"(inc.x));
            } else if (MODE == 1) {
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(inc));
            } else {
                asm volatile("v_sqrt_f32 %0, %1" : "=v"(x[i].x) : "v"(x[i].x));
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.0f;
    for (int i = 0; i < kUnroll; ++i)
        s += x[i].x + x[i].y;
    if (s == 12345.0f)
        sink[threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0)
        cycles[wave] = t1 - t0;
}

template <int MODE>
static double run(int quiet, int blocks, unsigned long long *dc, float *ds)
{
    overlap_kernel<MODE><<<blocks, 512>>>(quiet, dc, ds);
    (void)hipDeviceSynchronize();
    overlap_kernel<MODE><<<blocks, 512>>>(quiet, dc, ds);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 4; ++i)
        c += (double)h[i];
    return c / 4 / ((double)kIters * kUnroll);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;          // one workgroup per CU
    unsigned long long *dc;
    float *ds;
    (void)hipMalloc(&dc, 8 * sizeof(unsigned long long));
    (void)hipMalloc(&ds, 512 * sizeof(float));
    printf("clock64 ticks per VALU instruction of one wave, alone on its SIMD / beside a wave issuing v_mfma_f32_32x32x16_f16:\n");
    printf("  v_add_f32     %6.2f  / %6.2f\n", run<0>(1, blocks, dc, ds), run<0>(0, blocks, dc, ds));
    printf("  v_pk_add_f32  %6.2f  / %6.2f\n", run<1>(1, blocks, dc, ds), run<1>(0, blocks, dc, ds));
    printf("  v_sqrt_f32    %6.2f  / %6.2f\n", run<2>(1, blocks, dc, ds), run<2>(0, blocks, dc, ds));
    return 0;
}
