// Feature-distance farthest point sampling (SPEC.md §15).  No reference
// source exists (/root/reference/README.md:1-2).
//
// The metric needs C+3 subtract/multiply/add triples per point pair, so evaluating it inside the
// serial sampling chain (one workgroup per scene, M dependent steps) would stream N*C floats per
// step through one CU.  Instead the work is split the way the hardware likes it:
//   1. pairdist_kernel — ALL N x N distances of a scene, embarrassingly parallel over the whole
//      chip: 64x64 output tiles, channel chunks staged in LDS, 4x4 pairs per thread, each pair
//      accumulated in exactly the SPEC §15 order (no FMA), coalesced float4 stores.  The matrix is
//      symmetric bit for bit ((a-b)^2 == (b-a)^2), it is computed in full to keep the stores simple.
//   2. fps_dmat_kernel — the serial chain: per step one coalesced row of the matrix (N floats, L2)
//      replaces the distance evaluation; min-distances live in registers; arg-max on 64-bit keys
//      (distance bits << 32 | ~index), ties -> lowest index.
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int PD_TILE = 64, PD_CK = 32;

__global__ __launch_bounds__(256) void pairdist_kernel(const float *__restrict__ xyz, const float *__restrict__ feat,
                                                       int ld_feat, int N, int C, float w_xyz,
                                                       float *__restrict__ dmat) {
    __shared__ float s_fi[PD_TILE][PD_CK + 1], s_fj[PD_TILE][PD_CK + 1];
    __shared__ float s_pi[PD_TILE][3], s_pj[PD_TILE][3];
    const int b = blockIdx.z, i0 = blockIdx.y * PD_TILE, j0 = blockIdx.x * PD_TILE;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float *p = xyz + (size_t)b * N * 3;
    const float *f = feat + (size_t)b * N * ld_feat;
    if (tid < PD_TILE * 3) {
        const int r = tid / 3, d = tid - r * 3;
        const int i = i0 + r < N ? i0 + r : N - 1, j = j0 + r < N ? j0 + r : N - 1;
        s_pi[r][d] = p[(size_t)i * 3 + d];
        s_pj[r][d] = p[(size_t)j * 3 + d];
    }
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *pi = s_pi[ty * 4 + a], *pj = s_pj[tx * 4 + q];
            acc[a][q] = sad::d2f(pi[0], pi[1], pi[2], pj[0], pj[1], pj[2]) * w_xyz;
        }
    for (int c0 = 0; c0 < C; c0 += PD_CK) {
        const int nc = C - c0 < PD_CK ? C - c0 : PD_CK;
        __syncthreads();
        for (int e = tid; e < PD_TILE * PD_CK; e += 256) {
            const int r = e / PD_CK, c = e - r * PD_CK;
            const int i = i0 + r < N ? i0 + r : N - 1, j = j0 + r < N ? j0 + r : N - 1;
            s_fi[r][c] = c < nc ? f[(size_t)i * ld_feat + c0 + c] : 0.f;
            s_fj[r][c] = c < nc ? f[(size_t)j * ld_feat + c0 + c] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            float fi[4], fj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { fi[a] = s_fi[ty * 4 + a][c]; fj[a] = s_fj[tx * 4 + a][c]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fi[a] - fj[q];
                    const float tt = t * t;
                    acc[a][q] = acc[a][q] + tt;
                }
        }
    }
    float *out = dmat + (size_t)b * N * N;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ty * 4 + a, j = j0 + tx * 4;
        if (i >= N) continue;
        if (j + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<float4 *>(out + (size_t)i * N + j) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j + q < N) out[(size_t)i * N + j + q] = acc[a][q];
        }
    }
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xF, 0xF, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {   // full-wave max, wave-uniform result
    v = dpp_max_u32<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_max_u32<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_max_u32<0x141>(v);   // row_half_mirror
    v = dpp_max_u32<0x140>(v);   // row_mirror
    unsigned o = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false);   // row_bcast:15
    v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false);            // row_bcast:31
    v = o > v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}

// The chain per step: one coalesced row of the matrix, min, per-thread best, a two-phase 32-bit DPP
// wave max (distance bits, then ~index among the ties), one LDS atomic max per wave into a rotating
// slot, one barrier, one broadcast read.
template <int PPT>
__global__ __launch_bounds__(1024) void fps_dmat_kernel(const float *__restrict__ dmat, int N, int M,
                                                        int *__restrict__ idx_out) {
    __shared__ u64 s_gkey[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const float *D = dmat + (size_t)blockIdx.x * N * N;
    int *out = idx_out + (size_t)blockIdx.x * M;
    float md[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) md[k] = tid + k * 1024 < N ? __builtin_inff() : -1.0f;   // padding never wins
    if (tid == 0) out[0] = 0;
    if (tid < 3) s_gkey[tid] = 0ull;
    __syncthreads();
    int last = 0, b3 = 1;
    for (int i = 1; i < M; ++i) {
        const float *row = D + (size_t)last * N;
        unsigned bhi = 0u, blo = 0u;        // key = (distance bits, ~index); real distances are >= +0
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int j = tid + k * 1024;
            if (j < N) {
                const float d = row[j];
                const float m = md[k] < d ? md[k] : d;
                md[k] = m;
                const unsigned hi = __builtin_bit_cast(unsigned, m), lo = ~(unsigned)j;
                const bool better = hi > bhi || (hi == bhi && lo > blo);
                bhi = better ? hi : bhi;
                blo = better ? lo : blo;
            }
        }
        const unsigned whi = wave_max_u32(bhi);
        const unsigned wlo = wave_max_u32(bhi == whi ? blo : 0u);
        const int b3n = b3 == 2 ? 0 : b3 + 1;
        if (lane == 0) {
            atomicMax(&s_gkey[b3], ((u64)whi << 32) | wlo);
            if (tid == 0) s_gkey[b3n] = 0ull;
        }
        __syncthreads();
        const u64 g = s_gkey[b3];
        last = __builtin_amdgcn_readfirstlane((int)(~(unsigned)g));
        b3 = b3n;
        if (tid == 0) out[i] = last;
    }
}

}  // namespace

SAD_API size_t sad_ffps_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * (size_t)N * (size_t)N * sizeof(float);
}

SAD_API int sad_pairdist_f32(const float *xyz, const float *feat, int ld_feat, int B, int N, int C, float w_xyz,
                             float *dmat, sad_stream_t stream) {
    SAD_REQUIRE(xyz && dmat && (feat || C == 0), "sad_pairdist_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && C >= 0 && (C == 0 || ld_feat >= C), "sad_pairdist_f32: bad sizes");
    const int T = (N + PD_TILE - 1) / PD_TILE;
    SAD_REQUIRE(T <= 65535, "sad_pairdist_f32: N too large");
    hipLaunchKernelGGL(pairdist_kernel, dim3(T, T, B), dim3(256), 0, (hipStream_t)stream, xyz, feat ? feat : xyz,
                       ld_feat, N, C, w_xyz, dmat);
    return sad::check_launch("sad_pairdist_f32");
}

SAD_API int sad_ffps_f32(const float *xyz, const float *feat, int ld_feat, int B, int N, int C, int M, float w_xyz,
                         int32_t *idx, void *workspace, sad_stream_t stream) {
    SAD_REQUIRE(idx && workspace, "sad_ffps_f32: NULL idx / workspace (sad_ffps_workspace_bytes)");
    SAD_REQUIRE(M >= 1 && M <= N, "sad_ffps_f32: need 1 <= M <= N (N=%d M=%d)", N, M);
    if (N > 16384) return sad::fail(SAD_EUNSUPPORTED, "sad_ffps_f32: N=%d > 16384", N);
    float *dmat = (float *)workspace;
    if (int e = sad_pairdist_f32(xyz, feat, ld_feat, B, N, C, w_xyz, dmat, stream)) return e;
    const int ppt = (N + 1023) / 1024;
    hipStream_t st = (hipStream_t)stream;
    if (ppt <= 1) hipLaunchKernelGGL((fps_dmat_kernel<1>), dim3(B), dim3(1024), 0, st, dmat, N, M, idx);
    else if (ppt <= 2) hipLaunchKernelGGL((fps_dmat_kernel<2>), dim3(B), dim3(1024), 0, st, dmat, N, M, idx);
    else if (ppt <= 4) hipLaunchKernelGGL((fps_dmat_kernel<4>), dim3(B), dim3(1024), 0, st, dmat, N, M, idx);
    else if (ppt <= 8) hipLaunchKernelGGL((fps_dmat_kernel<8>), dim3(B), dim3(1024), 0, st, dmat, N, M, idx);
    else hipLaunchKernelGGL((fps_dmat_kernel<16>), dim3(B), dim3(1024), 0, st, dmat, N, M, idx);
    return sad::check_launch("sad_ffps_f32");
}
