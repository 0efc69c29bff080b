// Probe: the k-loop of the MLP kernels in isolation — per k-group NL coalesced 1 KB fragment loads from an
// L2-resident array (each wave its own phase) feeding 16 MFMAs (4 accumulators), loads PF k-groups ahead.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NL, int PF>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ w, int nfrag, int kgroups, float *out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned pos = (wave * 2654435761u) % (unsigned)nfrag;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    float4 ring[PF][NL];
    auto ld = [&](unsigned p) { return (w + (size_t)(p % (unsigned)nfrag) * 64)[lane]; };
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int l = 0; l < NL; ++l) ring[u][l] = ld(pos + u * NL + l);
    pos += PF * NL;
#pragma unroll 1
    for (int g = 0; g < kgroups; g += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                const float4 a = ring[u][oc % NL];
                const float4 b = ring[u][(oc + 1) % NL];
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[oc], 0, 0, 0);
                acc[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[oc], 0, 0, 0);
                if (oc == 3) {
#pragma unroll
                    for (int l = 0; l < NL; ++l) ring[u][l] = ld(pos + l);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            pos += NL;
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int g = 0; g < 16; ++g) s += acc[i][g];
    if (s == 123.456f) out[0] = s;
}
template <int NL, int PF>
void run(int wg_per_cu, int cus, const float4 *w, int nfrag, float *o) {
    const int kg = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NL, PF><<<cus * wg_per_cu, 256>>>(w, nfrag, 64, o);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NL, PF><<<cus * wg_per_cu, 256>>>(w, nfrag, kg, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)cus * wg_per_cu * 4 * kg * 16.0;
    printf("loads/kg %d, prefetch %d kg, %d wave(s)/SIMD: %.1f TFLOP/s  (%.1f B/clk/CU of fragment traffic at 2.4 GHz)\n", NL, PF, wg_per_cu,
           mf * 4096 / (ms * 1e-3) / 1e12, (double)cus * wg_per_cu * 4 * kg * NL * 1024.0 / (ms * 1e-3) / cus / 2.4e9);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int nfrag = 2048;   // 2 MB, L2-resident
    float4 *w; float *o;
    hipMalloc(&w, (size_t)nfrag * 1024); hipMalloc(&o, 4);
    float *h = (float *)malloc((size_t)nfrag * 1024);
    for (size_t i = 0; i < (size_t)nfrag * 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(w, h, (size_t)nfrag * 1024, hipMemcpyHostToDevice);
    const int cus = pr.multiProcessorCount;
    for (int wv : {1, 2, 3}) {
        run<1, 2>(wv, cus, w, nfrag, o);
        run<2, 2>(wv, cus, w, nfrag, o);
        run<4, 2>(wv, cus, w, nfrag, o);
        run<4, 4>(wv, cus, w, nfrag, o);
        run<5, 2>(wv, cus, w, nfrag, o);
    }
    return 0;
}
