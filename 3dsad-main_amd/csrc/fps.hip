// Farthest point sampling for gfx950 (SPEC.md §2).  No reference source exists
// (/root/reference/README.md:1-2 is the whole upstream repository).
//
// One workgroup per scene.  The scene's points and their running min-distances live in VGPRs
// (PPT points per thread), so a sampling step touches no memory except the one broadcast read of
// the last pick's coordinates: the kernel is bound by the serial chain of M steps (per step: PPT
// distance updates per lane, a wave64 arg-max, one barrier, a 16-entry cross-wave arg-max), not by
// HBM — compulsory traffic is N*12 B in and M*4 B out per scene.
#include "common.h"

namespace {

// (d, i) is better than (bd, bi): larger distance, ties -> lower index (SPEC.md §2).
__device__ __forceinline__ bool better(float d, int i, float bd, int bi) {
    return d > bd || (d == bd && i < bi);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}

// All-lanes arg-max over `width` lanes (width = 64 or 16), every lane ends with the winner.
template <bool DPP, int WIDTH>
__device__ __forceinline__ void wave_argmax(float &bd, int &bi) {
    if constexpr (DPP) {
        // in-row butterfly on DPP (no LDS crossbar): quad xor 1, quad xor 2, half-row mirror,
        // row mirror; rows are then combined with two bpermute steps.
#define SAD_STEP(CTRL)                                             \
    {                                                              \
        float od = dpp_f<CTRL>(bd);                                \
        int oi = dpp_i<CTRL>(bi);                                  \
        bool t = better(od, oi, bd, bi);                           \
        bd = t ? od : bd;                                          \
        bi = t ? oi : bi;                                          \
    }
        SAD_STEP(0xB1)   // quad_perm [1,0,3,2]
        SAD_STEP(0x4E)   // quad_perm [2,3,0,1]
        SAD_STEP(0x141)  // row_half_mirror
        SAD_STEP(0x140)  // row_mirror
#undef SAD_STEP
        if constexpr (WIDTH == 64) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                float od = __shfl_xor(bd, off, 64);
                int oi = __shfl_xor(bi, off, 64);
                bool t = better(od, oi, bd, bi);
                bd = t ? od : bd;
                bi = t ? oi : bi;
            }
        }
    } else {
#pragma unroll
        for (int off = WIDTH / 2; off >= 1; off >>= 1) {
            float od = __shfl_xor(bd, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            bool t = better(od, oi, bd, bi);
            bd = t ? od : bd;
            bi = t ? oi : bi;
        }
    }
}

template <int THREADS, int PPT, bool DPP>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(const float *__restrict__ xyz, int N,
                                                          int M, int *__restrict__ idx_out) {
    constexpr int NW = THREADS / 64;
    static_assert(NW == 16 || NW == 4, "cross-wave reduce assumes 4 or 16 waves");
    __shared__ float s_d[2][16];
    __shared__ int s_i[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    int *out = idx_out + (size_t)blockIdx.x * M;

    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int j = k * THREADS + tid;
        if (j < N) {
            px[k] = p[j * 3 + 0];
            py[k] = p[j * 3 + 1];
            pz[k] = p[j * 3 + 2];
            md[k] = __builtin_inff();
        } else {  // padding never wins: real min-distances are >= 0
            px[k] = py[k] = pz[k] = 0.f;
            md[k] = -1.f;
        }
    }
    if (tid == 0) out[0] = 0;
    int last = 0;
    for (int i = 1; i < M; ++i) {
        const float cx = p[last * 3 + 0], cy = p[last * 3 + 1], cz = p[last * 3 + 2];
        float bd = -1.f;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float d = sad::d2f(px[k], py[k], pz[k], cx, cy, cz);
            const float m = md[k] < d ? md[k] : d;
            md[k] = m;
            if (m > bd) {  // strict: ascending j inside the thread keeps the lowest index
                bd = m;
                bi = k * THREADS + tid;
            }
        }
        wave_argmax<DPP, 64>(bd, bi);
        const int buf = i & 1;
        if (lane == 0) {
            s_d[buf][wave] = bd;
            s_i[buf][wave] = bi;
        }
        __syncthreads();
        bd = s_d[buf][lane & (NW - 1)];
        bi = s_i[buf][lane & (NW - 1)];
        if constexpr (NW == 16) {
            wave_argmax<DPP, 16>(bd, bi);
        } else {
#pragma unroll
            for (int off = 2; off >= 1; off >>= 1) {
                float od = __shfl_xor(bd, off, 64);
                int oi = __shfl_xor(bi, off, 64);
                bool t = better(od, oi, bd, bi);
                bd = t ? od : bd;
                bi = t ? oi : bi;
            }
        }
        last = __builtin_amdgcn_readfirstlane(bi);
        if (tid == 0) out[i] = last;
    }
}

// ---- v2: 64-bit (distance, ~index) keys, DPP/readlane reductions, pick coordinates via LDS -------
// key = float_bits(min_dist) << 32 | ~index : unsigned max = largest distance, ties -> lowest index
// (min-distances are >= 0, so their bit patterns order like the floats).  Per step: PPT distance
// updates per lane, a wave arg-max (4 DPP butterflies + 4 readlanes), the wave's winning lane puts
// its key and its point's coordinates into LDS, ONE barrier, every wave reduces the 16 keys and
// reads the winner's coordinates back from LDS — no global access on the serial chain.
typedef unsigned long long u64;

template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 row_max_u64(u64 k) {  // max over each row of 16 lanes
    k = umax64(k, dpp_u64<0xB1>(k));
    k = umax64(k, dpp_u64<0x4E>(k));
    k = umax64(k, dpp_u64<0x141>(k));
    k = umax64(k, dpp_u64<0x140>(k));
    return k;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_key_kernel(const float *__restrict__ xyz, int N,
                                                          int M, int *__restrict__ idx_out) {
    constexpr int NW = THREADS / 64;
    typedef float fvec __attribute__((ext_vector_type(PPT)));
    __shared__ u64 s_key[2][16];
    __shared__ float s_xyz[2][16][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    int *out = idx_out + (size_t)blockIdx.x * M;

    fvec px, py, pz, md;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int j = k * THREADS + tid;
        if (j < N) {
            px[k] = p[j * 3 + 0];
            py[k] = p[j * 3 + 1];
            pz[k] = p[j * 3 + 2];
            md[k] = __builtin_inff();
        } else {  // padding: distance 0 and an index above every real one never beats a real point
            px[k] = py[k] = pz[k] = 0.f;
            md[k] = 0.f;
        }
    }
    if (tid == 0) out[0] = 0;
    float cx = p[0], cy = p[1], cz = p[2];
    for (int i = 1; i < M; ++i) {
        // whole-vector form: independent per-point chains the scheduler can interleave (packed f32)
        const fvec dx = px - cx, dy = py - cy, dz = pz - cz;
        const fvec d = (dx * dx + dy * dy) + dz * dz;          // SPEC.md §1 order, no contraction
        md = __builtin_elementwise_min(md, d);                 // no NaNs by contract (SPEC.md)
        float bd = md[0];
        int bk = 0;
#pragma unroll
        for (int k = 1; k < PPT; ++k) {
            if (md[k] > bd) {  // strict: ascending k keeps the thread's lowest index
                bd = md[k];
                bk = k;
            }
        }
        const unsigned j = (unsigned)(bk * THREADS + tid);
        u64 key = ((u64)__builtin_bit_cast(unsigned, bd) << 32) | (unsigned)(~j);
        key = row_max_u64(key);
        const u64 k0 = readlane_u64(key, 0), k1 = readlane_u64(key, 16), k2 = readlane_u64(key, 32), k3 = readlane_u64(key, 48);
        const u64 wkey = umax64(umax64(k0, k1), umax64(k2, k3));   // wave-uniform
        const unsigned wj = ~(unsigned)wkey;                        // the wave's best point
        const int buf = i & 1;
        if ((unsigned)tid == (wj & (THREADS - 1))) {                // its owner lane (in this wave)
            const int kk = __builtin_amdgcn_readfirstlane((int)(wj / THREADS));
            s_key[buf][wave] = wkey;
            s_xyz[buf][wave][0] = px[kk];
            s_xyz[buf][wave][1] = py[kk];
            s_xyz[buf][wave][2] = pz[kk];
        }
        __syncthreads();
        u64 g = s_key[buf][lane & (NW - 1)];
        if constexpr (NW == 16) {
            g = row_max_u64(g);
        } else {
            g = umax64(g, dpp_u64<0xB1>(g));
            g = umax64(g, dpp_u64<0x4E>(g));
        }
        const unsigned gj = __builtin_amdgcn_readfirstlane(~(unsigned)g);
        const int gw = (gj & (THREADS - 1)) >> 6;                   // wave that owns the winner
        cx = s_xyz[buf][gw][0];
        cy = s_xyz[buf][gw][1];
        cz = s_xyz[buf][gw][2];
        if (tid == 0) out[i] = (int)gj;
    }
}

// Any N: min-distances in a global workspace (L2-resident), coordinates re-read every step.
__global__ __launch_bounds__(1024) void fps_big_kernel(const float *__restrict__ xyz, int N, int M,
                                                       float *__restrict__ mind_ws,
                                                       int *__restrict__ idx_out) {
    __shared__ float s_d[2][16];
    __shared__ int s_i[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    float *md = mind_ws + (size_t)blockIdx.x * N;
    int *out = idx_out + (size_t)blockIdx.x * M;
    for (int j = tid; j < N; j += 1024) md[j] = __builtin_inff();
    if (tid == 0) out[0] = 0;
    int last = 0;
    for (int i = 1; i < M; ++i) {
        const float cx = p[last * 3 + 0], cy = p[last * 3 + 1], cz = p[last * 3 + 2];
        float bd = -1.f;
        int bi = 0;
        for (int j = tid; j < N; j += 1024) {  // each thread owns the same j every step
            const float d = sad::d2f(p[j * 3 + 0], p[j * 3 + 1], p[j * 3 + 2], cx, cy, cz);
            const float o = md[j];
            const float m = o < d ? o : d;
            md[j] = m;
            if (m > bd) {
                bd = m;
                bi = j;
            }
        }
        wave_argmax<false, 64>(bd, bi);
        const int buf = i & 1;
        if (lane == 0) {
            s_d[buf][wave] = bd;
            s_i[buf][wave] = bi;
        }
        __syncthreads();
        bd = s_d[buf][lane & 15];
        bi = s_i[buf][lane & 15];
        wave_argmax<false, 16>(bd, bi);
        last = __builtin_amdgcn_readfirstlane(bi);
        if (tid == 0) out[i] = last;
    }
}

template <int THREADS, int PPT>
void launch_reg(const float *xyz, int B, int N, int M, int *idx, hipStream_t st, bool dpp) {
    if (sad::get_option(sad::OPT_FPS_VARIANT) != 1) {  // default: the key kernel; 1 = (d, idx) pair kernel
        hipLaunchKernelGGL((fps_key_kernel<THREADS, PPT>), dim3(B), dim3(THREADS), 0, st, xyz, N, M, idx);
        return;
    }
    if (dpp)
        hipLaunchKernelGGL((fps_reg_kernel<THREADS, PPT, true>), dim3(B), dim3(THREADS), 0, st, xyz, N, M, idx);
    else
        hipLaunchKernelGGL((fps_reg_kernel<THREADS, PPT, false>), dim3(B), dim3(THREADS), 0, st, xyz, N, M, idx);
}

}  // namespace

// N > 16384: min-distance array of the global-workspace kernel; 2048 <= N <= 16384: the Z-order
// permutation of the bucketed kernel (optional: without it the plain register kernel runs).
SAD_API size_t sad_fps_workspace_bytes(int B, int N) {
    if (B <= 0 || N < 2048) return 0;
    size_t n = (size_t)B * (size_t)N * sizeof(float);
    // + sorted float4 records (used above 16384 points, or for any N with fps_variant = 5)
    if (N <= 65536) n = ((n + 15) & ~(size_t)15) + (size_t)B * 65536 * 16;
    return n;
}

SAD_API int sad_fps_f32(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace,
                        sad_stream_t stream) {
    SAD_REQUIRE(xyz && idx, "sad_fps_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && N >= 1 && M >= 1 && M <= N, "sad_fps_f32: need B>=1, 1<=M<=N (B=%d N=%d M=%d)", B, N, M);
    SAD_REQUIRE((size_t)N * 3 < (1u << 31), "sad_fps_f32: N too large");
    hipStream_t st = (hipStream_t)stream;
    const bool dpp = sad::get_option(sad::OPT_FPS_DPP) != 0;
    const int variant0 = sad::get_option(sad::OPT_FPS_VARIANT);
    if (((N > 16384 && (variant0 == 0 || variant0 >= 3)) || (variant0 == 5 && N >= 2048)) && N <= 65536 && workspace) {
        SAD_REQUIRE((uintptr_t)workspace % 16 == 0, "sad_fps_f32: workspace must be 16-byte aligned");
        return sad::launch_fps_cellg(xyz, B, N, M, idx, workspace, st);
    }
    if (N > 16384) {
        SAD_REQUIRE(workspace, "sad_fps_f32: N=%d > 16384 needs sad_fps_workspace_bytes() of workspace", N);
        hipLaunchKernelGGL(fps_big_kernel, dim3(B), dim3(1024), 0, st, xyz, N, M, (float *)workspace, idx);
        return sad::check_launch("sad_fps_f32");
    }
    const int variant = sad::get_option(sad::OPT_FPS_VARIANT);   // 0 auto (cell buckets), 1 pair, 2 key, 3 wave buckets, 4 cell buckets, 5 cell buckets over sorted records for any N >= 2048, 6 cell buckets (first form of the kernel)
    if (workspace && N >= 2048 && (variant == 0 || variant >= 3))
        return sad::launch_fps_bucket(xyz, B, N, M, idx, workspace, st);
    if (N <= 2048) {
        const int ppt = (N + 255) / 256;
        if (ppt <= 1) launch_reg<256, 1>(xyz, B, N, M, idx, st, dpp);
        else if (ppt <= 2) launch_reg<256, 2>(xyz, B, N, M, idx, st, dpp);
        else if (ppt <= 4) launch_reg<256, 4>(xyz, B, N, M, idx, st, dpp);
        else launch_reg<256, 8>(xyz, B, N, M, idx, st, dpp);
    } else {
        const int ppt = (N + 1023) / 1024;
        if (ppt <= 4) launch_reg<1024, 4>(xyz, B, N, M, idx, st, dpp);
        else if (ppt <= 8) launch_reg<1024, 8>(xyz, B, N, M, idx, st, dpp);
        else launch_reg<1024, 16>(xyz, B, N, M, idx, st, dpp);
    }
    return sad::check_launch("sad_fps_f32");
}
