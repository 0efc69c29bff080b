// Probe: how fast can every CU stream a small (L2-resident) array with wave-wide 16-byte loads?
// (the weight-fragment stream of csrc/mlp_reg.hip: 1 KB per wave-instruction, every wave reads the whole array)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void stream(const float4 *__restrict__ w, int nfrag, int reps, float *out) {
    const int lane = threadIdx.x & 63;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        const float4 *p = w + lane;
        // every wave starts somewhere else in the array (no two waves of a CU share a cache line in time)
        const int start = (int)(((blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u) % (unsigned)nfrag);
#pragma unroll 8
        for (int k0 = 0; k0 < nfrag; ++k0) {
            int k = start + k0; if (k >= nfrag) k -= nfrag;
            const float4 v = p[(size_t)k * 64];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.f) out[0] = acc.x;
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int kb : {64, 460, 2048, 16384}) {
        const int nfrag = kb;               // 1 KB fragments
        float4 *w; float *o;
        hipMalloc(&w, (size_t)kb * 1024); hipMalloc(&o, 4);
        hipMemset(w, 0, (size_t)kb * 1024);
        for (int wg_per_cu : {1, 2, 4}) {
            const int reps = kb >= 2048 ? 4 : 64;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            stream<<<cus * wg_per_cu, 256>>>(w, nfrag, 2, o);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            stream<<<cus * wg_per_cu, 256>>>(w, nfrag, reps, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)cus * wg_per_cu * 4 * reps * nfrag * 1024.0;
            printf("array %6d KB, %d WG/CU (%2d waves/CU): %.2f TB/s = %.1f B/clk/CU at 2.4 GHz\n", kb, wg_per_cu, wg_per_cu * 4,
                   bytes / ms / 1e9, bytes / (ms * 1e-3) / cus / 2.4e9);
        }
        hipFree(w); hipFree(o);
    }
    return 0;
}
