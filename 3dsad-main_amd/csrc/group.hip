// group_points / gather_points / gather_xyz for gfx950 (SPEC.md §5): index gathers.
// No reference source exists (/root/reference/README.md:1-2).
//
// HBM-bound copies: the (m,s) axis is the lane axis, so the index read and the grouped write are
// coalesced (16 B per lane when M*S is a multiple of 4); the gathered source row (N elements of one
// channel) is small enough to be served by L2.  Each thread reuses its indices for CH channels.
#include "common.h"

namespace {

constexpr int CH = 8;

template <typename T, int VEC>
__global__ __launch_bounds__(256) void group_kernel(const T *__restrict__ feat,
                                                    const int32_t *__restrict__ idx, int C, int N,
                                                    int MS, T *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CH;
    const int t = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (t >= MS) return;
    int ix[VEC];
    if constexpr (VEC == 4) {
        const int4 v = *reinterpret_cast<const int4 *>(idx + (size_t)b * MS + t);
        ix[0] = v.x; ix[1] = v.y; ix[2] = v.z; ix[3] = v.w;
    } else {
        ix[0] = idx[(size_t)b * MS + t];
    }
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const int c = c0 + cc;
        if (c >= C) break;
        const T *src = feat + ((size_t)b * C + c) * N;
        T *dst = out + ((size_t)b * C + c) * MS + t;
        if constexpr (VEC == 4 && sizeof(T) == 4) {
            uint4 v;
            v.x = src[ix[0]]; v.y = src[ix[1]]; v.z = src[ix[2]]; v.w = src[ix[3]];
            *reinterpret_cast<uint4 *>(dst) = v;
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) dst[k] = src[ix[k]];
        }
    }
}

// LDS-staged variant: a workgroup copies the source rows of CHL channels of one scene (CHL*N
// elements <= 128 KB) into LDS once — coalesced — and then serves the random per-neighbour reads
// from LDS instead of L1/L2; the (m,s) axis stays the lane axis, 16-byte index loads and 16-byte
// grouped stores.  Several workgroups split the (m,s) range of one (scene, channel block).
template <typename T>
__global__ __launch_bounds__(1024) void group_lds_kernel(const T *__restrict__ feat, const int32_t *__restrict__ idx,
                                                         int C, int N, int MS, int CHL, int per_split,
                                                         T *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s = reinterpret_cast<T *>(smem_raw);
    const int b = blockIdx.z, c0 = blockIdx.y * CHL;
    const int nch = C - c0 < CHL ? C - c0 : CHL;
    const T *src = feat + ((size_t)b * C + c0) * N;
    const int total = nch * N;
    constexpr int EV = 16 / sizeof(T);
    if ((total % EV) == 0 && ((uintptr_t)src % 16) == 0) {
        for (int e = threadIdx.x; e < total / EV; e += 1024)
            reinterpret_cast<uint4 *>(s)[e] = reinterpret_cast<const uint4 *>(src)[e];
    } else {
        for (int e = threadIdx.x; e < total; e += 1024) s[e] = src[e];
    }
    __syncthreads();
    const int t0 = blockIdx.x * per_split;
    const int t1 = t0 + per_split < MS ? t0 + per_split : MS;
    for (int t = t0 + threadIdx.x * 4; t < t1; t += 1024 * 4) {
        const int4 v = *reinterpret_cast<const int4 *>(idx + (size_t)b * MS + t);
        for (int cc = 0; cc < nch; ++cc) {
            const T *row = s + cc * N;
            T *dst = out + ((size_t)b * C + c0 + cc) * MS + t;
            if constexpr (sizeof(T) == 4) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 o;
                o.x = row[v.x]; o.y = row[v.y]; o.z = row[v.z]; o.w = row[v.w];
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst));   // streamed once, never re-read here
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                u32x2 o;
                o.x = (unsigned)row[v.x] | ((unsigned)row[v.y] << 16);
                o.y = (unsigned)row[v.z] | ((unsigned)row[v.w] << 16);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x2 *>(dst));
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_xyz_kernel(const float *__restrict__ xyz,
                                                         const int32_t *__restrict__ idx, int N,
                                                         int M, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float *s = xyz + ((size_t)b * N + idx[(size_t)b * M + m]) * 3;
    float *o = out + ((size_t)b * M + m) * 3;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
}

template <typename T>
int launch_group(const void *feat, const int32_t *idx, int B, int C, int N, long long MS, void *out,
                 hipStream_t st) {
    const bool vec = (MS % 4 == 0) && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0);
    // LDS-staged path: worth it when every source element is gathered several times
    const size_t slab_budget = 32 * 1024;   // two workgroups per CU: one stages while the other gathers
    int chl = (int)(slab_budget / ((size_t)N * sizeof(T)));
    chl = chl > 8 ? 8 : chl;
    if (vec && chl >= 1 && MS >= 4 * (long long)N && MS >= 8192 && sad::get_option(sad::OPT_GROUP_VARIANT) != 1) {
        static std::atomic<uint64_t> attr_done32{0}, attr_done16{0};
        sad::lds_attr_once(attr_done32, reinterpret_cast<const void *>(&group_lds_kernel<uint32_t>), 160 * 1024);
        sad::lds_attr_once(attr_done16, reinterpret_cast<const void *>(&group_lds_kernel<uint16_t>), 160 * 1024);
        if (chl > C) chl = C;
        const int cblocks = (C + chl - 1) / chl;
        // enough workgroups for ~4 per CU, each with at least 4096 outputs per channel
        long long splits = (1024LL + (long long)B * cblocks - 1) / ((long long)B * cblocks);
        const long long max_splits = (MS + 4095) / 4096;
        splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
        long long per_split = ((MS + splits - 1) / splits + 4095) / 4096 * 4096;
        splits = (MS + per_split - 1) / per_split;
        if (cblocks <= 65535 && splits <= 65535) {
            hipLaunchKernelGGL((group_lds_kernel<T>), dim3((unsigned)splits, cblocks, B), dim3(1024),
                               (size_t)chl * N * sizeof(T), st, (const T *)feat, idx, C, N, (int)MS, chl, (int)per_split,
                               (T *)out);
            return sad::check_launch("sad_group_points (lds)");
        }
    }
    const int per = vec ? 4 : 1;
    dim3 grid((unsigned)((MS + 256LL * per - 1) / (256LL * per)), (C + CH - 1) / CH, B);
    if (vec)
        hipLaunchKernelGGL((group_kernel<T, 4>), grid, dim3(256), 0, st, (const T *)feat, idx, C, N, (int)MS, (T *)out);
    else
        hipLaunchKernelGGL((group_kernel<T, 1>), grid, dim3(256), 0, st, (const T *)feat, idx, C, N, (int)MS, (T *)out);
    return sad::check_launch("sad_group_points");
}

}  // namespace

SAD_API int sad_group_points(const void *feat, const int32_t *idx, int B, int C, int N, int M, int S,
                             int elem_size, void *out, sad_stream_t stream) {
    SAD_REQUIRE(feat && idx && out, "sad_group_points: NULL pointer");
    SAD_REQUIRE(B >= 1 && C >= 1 && N >= 1 && M >= 1 && S >= 1, "sad_group_points: sizes must be >= 1");
    SAD_REQUIRE(elem_size == 2 || elem_size == 4, "sad_group_points: elem_size=%d (2 or 4)", elem_size);
    SAD_REQUIRE(B <= 65535 && (C + CH - 1) / CH <= 65535, "sad_group_points: B or C too large");
    const long long MS = (long long)M * S;
    SAD_REQUIRE(MS < (1LL << 31), "sad_group_points: M*S too large");
    if (elem_size == 4) return launch_group<uint32_t>(feat, idx, B, C, N, MS, out, (hipStream_t)stream);
    return launch_group<uint16_t>(feat, idx, B, C, N, MS, out, (hipStream_t)stream);
}

SAD_API int sad_gather_points(const void *src, const int32_t *idx, int B, int C, int N, int M,
                              int elem_size, void *out, sad_stream_t stream) {
    return sad_group_points(src, idx, B, C, N, M, 1, elem_size, out, stream);
}

SAD_API int sad_gather_xyz_f32(const float *xyz, const int32_t *idx, int B, int N, int M, float *out,
                               sad_stream_t stream) {
    SAD_REQUIRE(xyz && idx && out, "sad_gather_xyz_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && N >= 1 && M >= 1 && B <= 65535, "sad_gather_xyz_f32: bad sizes");
    dim3 grid((M + 255) / 256, B);
    hipLaunchKernelGGL(gather_xyz_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, idx, N, M, out);
    return sad::check_launch("sad_gather_xyz_f32");
}
