// Does VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
// One wave per SIMD (256 threads/WG, 256 WGs): loop of MFMAs with V extra
// independent v_fma_f32 per MFMA, from the same wave.  Also 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NACC; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < V; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int m = 0; m < NACC; ++m) for (int i = 0; i < 16; ++i) s += acc[m][i];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, int NACC>
void run(const char *name, int blocks)
{
    float *out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, NACC><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    k<V, NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * 4 * iters * NACC;      // MFMAs
    double tf = mf * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %4d V=%2d acc=%d : %7.3f ms  %6.1f TF/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", name, blocks, V, NACC,
           ms, tf, ms * 1e-3 * 2.4e9 / (mf / (256.0 * 4) * (blocks > 256 ? 256.0 / blocks : 1.0)) );
    hipFree(out);
}

int main()
{
    run<0, 1>("mfma only 1acc", 256);
    run<0, 4>("mfma only 4acc", 256);
    run<4, 4>("mfma + 4 fma", 256);
    run<8, 4>("mfma + 8 fma", 256);
    run<16, 4>("mfma + 16 fma", 256);
    run<32, 4>("mfma + 32 fma", 256);
    run<0, 4>("2 waves/SIMD mfma only", 512);
    run<8, 4>("2 waves/SIMD mfma + 8 fma", 512);
    run<16, 4>("2 waves/SIMD mfma + 16 fma", 512);
    return 0;
}
