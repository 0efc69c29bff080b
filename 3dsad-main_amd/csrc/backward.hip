// Backward kernels of the unfused operators (SPEC.md §16), so that
// group_points / gather_points / max-over-nsample are trainable.  No reference source exists
// (/root/reference/README.md:1-2).
//
// group_points_grad / gather_points_grad: scatter-add of the grouped gradient back to the source
// points — the (m,s) axis is the lane axis (coalesced gradient reads, indices reused for CH
// channels), the adds are hardware float atomics in L2 (global_atomic_add_f32).  The order of the
// additions is therefore not fixed: parity is a stated tolerance against a binary64 sum.
// max_pool_s: max over the nsample axis with the arg-max (ties -> lowest s), and its backward
// (route the gradient to the arg-max slot) — both exact.
#include "common.h"

namespace {

constexpr int CH = 8;

__global__ __launch_bounds__(256) void group_grad_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx,
                                                         int C, int N, int MS, float *__restrict__ gfeat) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CH;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= MS) return;
    const int j = idx[(size_t)b * MS + t];
    if ((unsigned)j >= (unsigned)N) return;   // out-of-range / negative index: contributes nothing (both variants agree)
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const int c = c0 + cc;
        if (c >= C) break;
        const float g = gout[((size_t)b * C + c) * MS + t];
        if (g != 0.f) atomicAdd(gfeat + ((size_t)b * C + c) * N + j, g);
    }
}

// Point-major variant: gradient rows are added into grad_feat_pm[B,N,C] with the CHANNEL axis on the
// lanes, so one wave-instruction adds 256 contiguous bytes — the shape the memory-side atomic units
// run at full rate (≈1.3 TB/s), where the channel-major scatter above has 64 lanes in 64 rows (≈17x
// slower, MI355X_MICROARCH.md "Global float atomics").  A 64-channel x 64-(m,s) tile of grad_out is
// read coalesced along (m,s), transposed through LDS, then each wave walks (m,s) rows.
__global__ __launch_bounds__(256) void group_grad_pm_kernel(const float *__restrict__ gout, const int32_t *__restrict__ idx,
                                                            int C, int N, int MS, float *__restrict__ gfeat_pm) {
    __shared__ float tile[64][65];
    __shared__ int sidx[64];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) sidx[threadIdx.x] = t0 + threadIdx.x < MS ? idx[(size_t)b * MS + t0 + threadIdx.x] : -1;
    for (int cc = wave; cc < 64; cc += 4) {            // lanes along (m,s): coalesced
        const int c = c0 + cc, t = t0 + lane;
        tile[cc][lane] = (c < C && t < MS) ? gout[((size_t)b * C + c) * MS + t] : 0.f;
    }
    __syncthreads();
    const int c = c0 + lane;
    for (int tt = wave; tt < 64; tt += 4) {            // lanes along channels: contiguous atomics
        const int j = sidx[tt];
        if ((unsigned)j >= (unsigned)N || c >= C) continue;   // padding slot (-1) or out-of-range index
        const float g = tile[lane][tt];
        if (g != 0.f) atomicAdd(gfeat_pm + ((size_t)b * N + j) * C + c, g);
    }
}

__global__ __launch_bounds__(256) void maxpool_s_kernel(const float *__restrict__ x, long long rows, int S,
                                                        float *__restrict__ out, int32_t *__restrict__ arg) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *p = x + (size_t)r * S;
    float best = p[0];
    int bi = 0;
    for (int s = 1; s < S; ++s) {
        const float v = p[s];
        if (v > best) { best = v; bi = s; }      // strict: ties keep the lowest s
    }
    out[r] = best;
    arg[r] = bi;
}

__global__ __launch_bounds__(256) void maxpool_s_grad_kernel(const float *__restrict__ gout, const int32_t *__restrict__ arg,
                                                             long long total, int S, float *__restrict__ gx) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long r = e / S;
    const int s = (int)(e - r * S);
    gx[e] = arg[r] == s ? gout[r] : 0.f;
}

}  // namespace

SAD_API int sad_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, int S,
                                      float *grad_feat, sad_stream_t stream) {
    SAD_REQUIRE(grad_out && idx && grad_feat, "sad_group_points_grad_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && C >= 1 && N >= 1 && M >= 1 && S >= 1, "sad_group_points_grad_f32: sizes must be >= 1");
    SAD_REQUIRE(B <= 65535 && (C + CH - 1) / CH <= 65535, "sad_group_points_grad_f32: B or C too large");
    const long long MS = (long long)M * S;
    SAD_REQUIRE(MS < (1LL << 31), "sad_group_points_grad_f32: M*S too large");
    dim3 grid((unsigned)((MS + 255) / 256), (C + CH - 1) / CH, B);
    hipLaunchKernelGGL(group_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, idx, C, N, (int)MS, grad_feat);
    return sad::check_launch("sad_group_points_grad_f32");
}

SAD_API int sad_group_points_grad_pm_f32(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, int S,
                                         float *grad_feat_pm, sad_stream_t stream) {
    SAD_REQUIRE(grad_out && idx && grad_feat_pm, "sad_group_points_grad_pm_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && C >= 1 && N >= 1 && M >= 1 && S >= 1, "sad_group_points_grad_pm_f32: sizes must be >= 1");
    const long long MS = (long long)M * S;
    SAD_REQUIRE(MS < (1LL << 31) && B <= 65535 && (C + 63) / 64 <= 65535, "sad_group_points_grad_pm_f32: sizes too large");
    dim3 grid((unsigned)((MS + 63) / 64), (C + 63) / 64, B);
    hipLaunchKernelGGL(group_grad_pm_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, idx, C, N, (int)MS, grad_feat_pm);
    return sad::check_launch("sad_group_points_grad_pm_f32");
}

SAD_API int sad_max_pool_s_f32(const float *x, int B, int C, int M, int S, float *out, int32_t *arg,
                               sad_stream_t stream) {
    SAD_REQUIRE(x && out && arg, "sad_max_pool_s_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && C >= 1 && M >= 1 && S >= 1, "sad_max_pool_s_f32: sizes must be >= 1");
    const long long rows = (long long)B * C * M;
    SAD_REQUIRE((rows + 255) / 256 < (1LL << 31), "sad_max_pool_s_f32: too many rows");
    hipLaunchKernelGGL(maxpool_s_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       rows, S, out, arg);
    return sad::check_launch("sad_max_pool_s_f32");
}

SAD_API int sad_max_pool_s_grad_f32(const float *grad_out, const int32_t *arg, int B, int C, int M, int S,
                                    float *grad_x, sad_stream_t stream) {
    SAD_REQUIRE(grad_out && arg && grad_x, "sad_max_pool_s_grad_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && C >= 1 && M >= 1 && S >= 1, "sad_max_pool_s_grad_f32: sizes must be >= 1");
    const long long total = (long long)B * C * M * S;
    SAD_REQUIRE((total + 255) / 256 < (1LL << 31), "sad_max_pool_s_grad_f32: too many elements");
    hipLaunchKernelGGL(maxpool_s_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, arg, total, S, grad_x);
    return sad::check_launch("sad_max_pool_s_grad_f32");
}
