// ds_read_b64_tr_b16 semantics check (gfx950): an LDS image [k][col] of 16-bit
// elements, pitch P columns; per 16-lane group, lane 4q+p supplies the address of
// row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 64;   // columns per k row
__global__ void k(short *out)
{
    __shared__ short lds[32 * PITCH];
    for (int i = threadIdx.x; i < 32 * PITCH; i += 64) lds[i] = (short)i;   // value = k * PITCH + col
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // group g: columns 16 * (g & 1) .., k rows 8 * (g >> 1) + q
    const int row = 8 * (g >> 1) + q, col = 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)(lds + row * PITCH + col));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = (8 * (g >> 1) + j) * PITCH + 16 * (g & 1) + i;   // k = 8*(g>>1)+j, col = 16*(g&1)+i
            if (h[lane * 4 + j] != want) ++bad;
        }
        if (lane < 4 || lane == 16 || lane == 32)
            printf("lane %2d: %d %d %d %d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
