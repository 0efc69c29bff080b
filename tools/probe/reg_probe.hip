// Probe: the building blocks of csrc/mlp_reg.hip (ktile, to_operands) on one wave against the CPU.
#include "../../3dsad-main_amd/csrc/mlp_reg.hip"
#include <math.h>
#include <stdlib.h>
namespace sad { void set_error(const char *, ...) {} int get_option(int) { return 0; } }
// one layer: X[32 rows][8 ch] (row-major) -> Y[32][32] = W[32][8] x + b, raw accumulators dumped
__global__ void probe1(const float *X, const float *packed /*bias[32] + frags*/, float *Y, float *Y2, const float *packed2, float *DBG, int DIRECT) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    __shared__ float sb[64];
    if (lane < 32) { sb[lane] = packed[lane]; sb[32 + lane] = packed2[lane]; }
    __syncthreads();
    const float4 v = *reinterpret_cast<const float4 *>(X + j * 8 + 4 * h);
    float in0[4];
    to_operands(v.x, v.y, v.z, v.w, in0);
    for (int e = 0; e < 4; ++e) DBG[e * 64 + lane] = in0[e];
    if (DIRECT) for (int e = 0; e < 4; ++e) in0[e] = X[j * 8 + 2 * e + h];
    f32x16 t = bias_tile(sb, h);
    t = ktile<1>(t, reinterpret_cast<const float4 *>(packed + 32) + lane, in0);
    for (int a = 0; a < 4; ++a)
        for (int q = 0; q < 4; ++q) Y[j * 32 + 8 * a + 4 * h + q] = t[4 * a + q];
    // second layer 32 -> 32 from the accumulators
    t = relu16(t);
    float bt[16];
    for (int a = 0; a < 4; ++a) to_operands(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3], bt + 4 * a);
    f32x16 u = bias_tile(sb + 32, h);
    u = ktile<4>(u, reinterpret_cast<const float4 *>(packed2 + 32) + lane, bt);
    for (int a = 0; a < 4; ++a)
        for (int q = 0; q < 4; ++q) Y2[j * 32 + 8 * a + 4 * h + q] = u[4 * a + q];
}
static void pack(const float *W, const float *b, int Cin, int KP, float *dst) {   // one 32-channel tile
    for (int i = 0; i < 32; ++i) dst[i] = b[i];
    for (int t = 0; t < KP / 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
                const int oc = lane & 31, k = 8 * t + 2 * e + (lane >> 5);
                dst[32 + (t * 64 + lane) * 4 + e] = k < Cin ? W[oc * Cin + k] : 0.f;
            }
}
int main() {
    float X[32 * 8], W[32 * 8], b[32], W2[32 * 32], b2[32], P[32 + 64 * 4], P2[32 + 4 * 64 * 4], Y[1024], Y2[1024];
    srand(1);
    for (auto &x : X) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto &x : W) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto &x : b) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto &x : W2) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto &x : b2) x = rand() / (float)RAND_MAX - 0.5f;
    pack(W, b, 8, 8, P);
    pack(W2, b2, 32, 32, P2);
    float *dX, *dP, *dY, *dY2, *dP2;
    hipMalloc(&dX, sizeof X); hipMalloc(&dP, sizeof P); hipMalloc(&dY, sizeof Y); hipMalloc(&dY2, sizeof Y2); hipMalloc(&dP2, sizeof P2);
    hipMemcpy(dX, X, sizeof X, hipMemcpyHostToDevice); hipMemcpy(dP, P, sizeof P, hipMemcpyHostToDevice); hipMemcpy(dP2, P2, sizeof P2, hipMemcpyHostToDevice);
    float *dD; float D[256]; hipMalloc(&dD, sizeof D);
  for (int direct = 0; direct < 2; ++direct) {
    probe1<<<1, 64>>>(dX, dP, dY, dY2, dP2, dD, direct);
    hipMemcpy(Y, dY, sizeof Y, hipMemcpyDeviceToHost); hipMemcpy(Y2, dY2, sizeof Y2, hipMemcpyDeviceToHost);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int e = 0; e < 4; ++e) for (int l = 0; l < 64; ++l) bad += D[e * 64 + l] != X[(l & 31) * 8 + 2 * e + (l >> 5)];
    printf("direct=%d: operand registers wrong: %d of 256\n", direct, bad);
    for (int e = 0; e < 4; ++e) { int b0 = 0, b1 = 0; for (int l = 0; l < 32; ++l) { b0 += D[e*64+l] != X[l*8+2*e]; b1 += D[e*64+32+l] != X[l*8+2*e+1]; }
      printf("  e=%d wrong lo=%d hi=%d; lane0 got %.4f want %.4f; lane32 got %.4f want %.4f (row0: %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f)\n", e, b0, b1, D[e*64], X[2*e], D[e*64+32], X[2*e+1], X[0],X[1],X[2],X[3],X[4],X[5],X[6],X[7]); }
    double e1 = 0, e2 = 0;
    for (int r = 0; r < 32; ++r) {
        float y1[32];
        for (int o = 0; o < 32; ++o) {
            float acc = b[o];
            for (int k = 0; k < 8; ++k) acc = fmaf(W[o * 8 + k], X[r * 8 + k], acc);
            e1 = fmax(e1, fabs(acc - Y[r * 32 + o]));
            y1[o] = acc > 0 ? acc : 0;
        }
        for (int o = 0; o < 32; ++o) {
            float acc = b2[o];
            for (int k = 0; k < 32; ++k) acc = fmaf(W2[o * 32 + k], y1[k], acc);
            e2 = fmax(e2, fabs(acc - Y2[r * 32 + o]));
        }
    }
    printf("layer 1 max err %g, layer 2 max err %g\n", e1, e2);
  }
    printf("Y[0][0..7] gpu: "); for (int i = 0; i < 8; ++i) printf("%.4f ", Y[i]); printf("\n");
    return 0;
}
