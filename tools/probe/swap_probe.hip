// Probe: semantics of __builtin_amdgcn_permlane32_swap and the DPP controls used by csrc/mlp_reg.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
    const int lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[lane] = r[0];
    o[64 + lane] = r[1];
    o[128 + lane] = __builtin_amdgcn_update_dpp(0, lane + 1, 0x111, 0xF, 0xF, true);   // row_shr:1
    o[192 + lane] = __builtin_amdgcn_update_dpp(0, lane + 1, 0x118, 0xF, 0xF, true);   // row_shr:8
    o[256 + lane] = __builtin_amdgcn_update_dpp(0, lane + 1, 0x142, 0xA, 0xF, true);   // row_bcast:15 rows 1,3
}
int main() {
    int *d, h[320];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[5] = {"swap r[0]", "swap r[1]", "row_shr:1", "row_shr:8", "row_bcast:15 (rows 1,3)"};
    for (int s = 0; s < 5; ++s) {
        printf("%s:\n", names[s]);
        for (int i = 0; i < 64; ++i) printf("%d%s", h[s * 64 + i], i % 16 == 15 ? "\n" : " ");
    }
    return 0;
}
