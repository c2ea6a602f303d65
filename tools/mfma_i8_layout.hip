// tools/mfma_i8_layout.hip -- which element of A and B does lane l hold for v_mfma_i32_32x32x32_i8 on gfx950?
// One wave multiplies A (32 x 32, A[r][k] = distinct small integers) by B (32 x 32, asymmetric) with the ASSUMED map
//   lane l: row (A) / column (B) = l & 31, k = 16 (l >> 5) + j for byte j = 0..15 of its four dwords,
//   C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
// and the host compares every element with the integer product.  hipcc --offload-arch=gfx950 -o mfma_i8_layout ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t *A, const int8_t *B, int *C)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) {
        ab[j] = A[r * 32 + 16 * h + j];          // A[row r][k]
        bb[j] = B[(16 * h + j) * 32 + r];        // B[k][col r]
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int g = 0; g < 16; ++g)
        C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
int main()
{
    int8_t hA[1024], hB[1024];
    int hC[1024], want[1024];
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            hA[i * 32 + j] = (int8_t)(((i * 7 + j * 13) % 251) - 125);
            hB[i * 32 + j] = (int8_t)(((i * 29 + j * 3 + (i > j ? 17 : 0)) % 241) - 120);
        }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int q = 0; q < 32; ++q)
                s += (int)hA[i * 32 + q] * (int)hB[q * 32 + j];
            want[i * 32 + j] = s;
        }
    int8_t *dA, *dB;
    int *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i)
        bad += hC[i] != want[i];
    printf("v_mfma_i32_32x32x32_i8 with the assumed operand and result maps: %d of 1024 elements wrong\n", bad);
    return bad != 0;
}
