// tools/dma_rate.hip -- how many bytes per clock does one CU take in through global_load_lds_dwordx4 (global -> LDS DMA)
// and through global_load_dwordx4 (global -> registers)?  One workgroup of 256 threads per CU streams rows of 128 bytes
// (8 lanes per row, as the refcos kernels do) from a window of `span` bytes, `depth` groups of 8 loads per thread in
// flight, for `iters` groups; prints bytes per clock and CU at the measured time and 2.4 GHz nominal.
// hipcc --offload-arch=gfx950 -O3 -o dma_rate tools/dma_rate.hip ; ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int wait_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
typedef int v4i __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256, 1) void dma_kernel(const unsigned char *src, size_t span, int iters, int *sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char sAll[4 * 32768];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t base = ((size_t)blockIdx.x * 977 * 32768) % span;
    unsigned off[8];
    for (int p = 0; p < 8; ++p)
        off[p] = (unsigned)((tid >> 3) + 32 * p) * 128u + (unsigned)(tid & 7) * 16u;
    auto fetch = [&](int it, unsigned char *dst) {
        const unsigned char *u = src + (base + (size_t)it * 32768) % span;
        asm volatile("" : "+s"(u));
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            asm volatile("" : "+v"(off[p]));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(u + off[p]),
                                             (__attribute__((address_space(3))) void *)&dst[(32 * p + 8 * wave) * 128], 16, 0, 0);
        }
    };
    for (int d = 0; d < DEPTH; ++d)
        fetch(d, &sAll[(d & 3) * 32768]);
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_waitcnt(wait_vmcnt(8 * (DEPTH - 1)));
        __syncthreads();                                  // (the LDS is not read inside the loop: the compiler would wait
        fetch(it + DEPTH, &sAll[((it + DEPTH) & 3) * 32768]);   //  for every DMA in front of a read it cannot tell apart)
    }
    __builtin_amdgcn_s_waitcnt(wait_vmcnt(0));
    __syncthreads();
    acc += *(const int *)&sAll[tid * 16];
    if (acc == 0x12345678)
        *sink = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256, 1) void reg_kernel(const unsigned char *src, size_t span, int iters, int *sink)
{
    const int tid = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 977 * 32768) % span;
    unsigned off[8];
    for (int p = 0; p < 8; ++p)
        off[p] = (unsigned)((tid >> 3) + 32 * p) * 128u + (unsigned)(tid & 7) * 16u;
    v4i r[DEPTH][8];
    auto fetch = [&](int it, v4i (&dst)[8]) {
        const unsigned char *u = src + (base + (size_t)it * 32768) % span;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            dst[p] = __builtin_nontemporal_load((const v4i *)(u + off[p]));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        fetch(d, r[d]);
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                acc ^= r[d][p];
            fetch(it + DEPTH + d, r[d]);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678)
        *sink = acc[0];
}

template <typename K>
static void run(const char *name, K kern, const unsigned char *d, size_t span, int iters, int *sink, int cus)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kern<<<cus, 256>>>(d, span, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<cus, 256>>>(d, span, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)iters * 32768.0;
    printf("%-34s span %6.1f MB: %.3f ms, %.1f GB/s per CU, %.1f bytes per clock and CU at 2.4 GHz (%.2f TB/s over %d CUs)\n", name,
           span / 1048576.0, ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9, bytes * cus / ms / 1e9, cus);
}

int main()
{
    const size_t maxSpan = 256u << 20;
    unsigned char *d;
    int *sink;
    hipMalloc(&d, maxSpan + (1 << 20));
    hipMalloc(&sink, 4);
    hipMemset(d, 1, maxSpan + (1 << 20));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int iters = 4096;
    for (size_t span : {(size_t)1 << 20, (size_t)16 << 20, (size_t)256 << 20}) {
        run("DMA to LDS, 3 groups in flight", dma_kernel<3>, d, span, iters, sink, cus);
        run("DMA to LDS, 1 group in flight", dma_kernel<1>, d, span, iters, sink, cus);
        run("loads to registers, 3 groups", reg_kernel<3>, d, span, iters, sink, cus);
        run("loads to registers, 6 groups", reg_kernel<6>, d, span, iters, sink, cus);
    }
    return 0;
}
