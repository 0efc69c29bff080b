// Probe 2: the same k-loop with the fragments read from LDS (ds_read_b128, lane-linear) and only NG global
// loads per k-group (cooperative fill not modelled: this measures what LDS-fed MFMAs can reach).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NLDS, int NG>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ w, int nfrag, int kgroups, float *out) {
    extern __shared__ float4 lds[];                 // 32 fragments of 64 float4
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32 * 64; i += 256) lds[i] = w[i];
    __syncthreads();
    unsigned pos = (wave * 2654435761u) % (unsigned)nfrag;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    float4 gl[2][NG > 0 ? NG : 1];
    auto ld = [&](unsigned p) { return (w + (size_t)(p % (unsigned)nfrag) * 64)[lane]; };
    for (int u = 0; u < 2; ++u) for (int l = 0; l < NG; ++l) gl[u][l] = ld(pos + u * NG + l);
    float4 lr[2][NLDS];
    for (int u = 0; u < 2; ++u) for (int l = 0; l < NLDS; ++l) lr[u][l] = lds[((u * NLDS + l) & 31) * 64 + lane];
    unsigned lp = 2 * NLDS;
    pos += 2 * NG;
#pragma unroll 1
    for (int g = 0; g < kgroups; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                float4 a = lr[u][oc % NLDS], b = lr[u][(oc + 1) % NLDS];
                if (NG > 0) { b.x += gl[u][0].x; }
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[oc], 0, 0, 0);
                if (oc == 3) {
#pragma unroll
                    for (int l = 0; l < NLDS; ++l) lr[u][l] = lds[((lp + l) & 31) * 64 + lane];
#pragma unroll
                    for (int l = 0; l < NG; ++l) gl[u][l] = ld(pos + l);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            lp += NLDS; pos += NG;
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) s += acc[i][g];
    if (s == 123.456f) out[0] = s;
}
template <int NLDS, int NG>
void run(int wg_per_cu, int cus, const float4 *w, int nfrag, float *o) {
    const int kg = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NLDS, NG><<<cus * wg_per_cu, 256, 32 * 1024>>>(w, nfrag, 64, o);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NLDS, NG><<<cus * wg_per_cu, 256, 32 * 1024>>>(w, nfrag, kg, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)cus * wg_per_cu * 4 * kg * 16.0;
    printf("per k-group (16 MFMAs): %d LDS fragment reads + %d global loads, %d wave(s)/SIMD: %.1f TFLOP/s\n", NLDS, NG, wg_per_cu, mf * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int nfrag = 2048;
    float4 *w; float *o;
    hipMalloc(&w, (size_t)nfrag * 1024); hipMalloc(&o, 4);
    float *h = (float *)malloc((size_t)nfrag * 1024);
    for (size_t i = 0; i < (size_t)nfrag * 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(w, h, (size_t)nfrag * 1024, hipMemcpyHostToDevice);
    const int cus = pr.multiProcessorCount;
    for (int wv : {1, 2, 3}) {
        run<4, 0>(wv, cus, w, nfrag, o);
        run<4, 1>(wv, cus, w, nfrag, o);
        run<5, 0>(wv, cus, w, nfrag, o);
    }
    return 0;
}
