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
            else if (MODE == 2)
                a[i] = __dadd_rn(a[i], y);
            else if (MODE == 3)
                a[i] = (double)(float)a[(i + 1) & 15];                      // v_cvt_f32_f64 + v_cvt_f64_f32
            else if (MODE == 4)
                a[i] = (double)(int)a[(i + 1) & 15];                        // v_cvt_i32_f64 + v_cvt_f64_i32
            else
                a[i] = sqrt(a[(i + 1) & 15] + y);                           // correctly rounded f64 square root (+ one add)
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
    const char *names[6] = {"v_mul_f64 + v_add_f64", "v_fma_f64", "v_add_f64", "cvt f64->f32->f64", "cvt f64->i32->f64", "sqrt(f64) + add"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate_kernel<0><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 1) rate_kernel<1><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 2) rate_kernel<2><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 3) rate_kernel<3><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 4) rate_kernel<4><<<blocks, 256>>>(out, iters, 1.0, 1.0000001);
            if (mode == 5) rate_kernel<5><<<blocks, 256>>>(out, iters / 8, 1.0, 1.0000001);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 /*waves*/ * (mode == 5 ? iters / 8 : iters) * 16 * ((mode == 0 || mode == 3 || mode == 4) ? 2 : 1);     // wave instructions (sqrt: per call)
        const double perSimd = insts / (p.multiProcessorCount * 4.0);
        printf("%-24s %.2f ms: %.2f ns per wave instruction per SIMD (%.1f cycles at 2.1 GHz)\n", names[mode], ms,
               ms * 1e6 / perSimd, ms * 1e6 / perSimd * 2.1);
    }
    return 0;
}
