// Probe: v_mfma_f32_32x32x2_f32 issue rate for dependent / independent accumulator chains, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int g = 0; g < 16; ++g) s += acc[i][g];
    if (s == 123.456f) out[0] = s;
}
template <int NACC>
void run(int wg_per_cu, int cus, float *o) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<cus * wg_per_cu, 256>>>(o, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<cus * wg_per_cu, 256>>>(o, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)cus * wg_per_cu * 4 * iters * 16.0 * NACC;
    printf("NACC=%d, %d wave(s)/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", NACC, wg_per_cu, mf * 4096 / (ms * 1e-3) / 1e12,
           (ms * 1e-3) * 2.4e9 / (mf / (cus * 4)));
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    float *o; hipMalloc(&o, 4);
    for (int w : {1, 2}) { run<1>(w, pr.multiProcessorCount, o); run<2>(w, pr.multiProcessorCount, o); run<4>(w, pr.multiProcessorCount, o); }
    return 0;
}
