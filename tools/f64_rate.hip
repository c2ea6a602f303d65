// tools/f64_rate.hip -- issue cost of the f64 instructions the bit-exact kernels are made of (v_mul_f64 + v_add_f64,
// no FMA), and of v_fma_f64 for comparison: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/f64_rate.hip -o tools/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, double x, double y)
{
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        a[i] = x + i + threadIdx.x;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0)
                a[i] = __dadd_rn(a[i], __dmul_rn(a[(i + 1) & 15], y));     // mul + add, rounded separately
            else if (MODE == 1)
                a[i] = __fma_rn(a[(i + 1) & 15], y, a[i]);
            else
                a[i] = __dadd_rn(a[i], y);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[3] = {"v_mul_f64 + v_add_f64", "v_fma_f64", "v_add_f64"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate_kernel<0><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 1) rate_kernel<1><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 2) rate_kernel<2><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 /*waves*/ * iters * 16 * (mode == 0 ? 2 : 1);     // wave instructions
        const double perSimd = insts / (p.multiProcessorCount * 4.0);
        printf("%-24s %.2f ms: %.2f ns per wave instruction per SIMD (%.1f cycles at 2.1 GHz)\n", names[mode], ms,
               ms * 1e6 / perSimd, ms * 1e6 / perSimd * 2.1);
    }
    return 0;
}
