// Candidate centres + adaptive radius (SPEC.md §8 steps 2-4) and box decode (SPEC.md §9).
// No reference source exists (/root/reference/README.md:1-2).  Tiny element-wise kernels.
#include "common.h"

namespace {

struct F3 { float v[3]; };
struct F9 { float v[9]; };

__global__ __launch_bounds__(256) void candidates_kernel(const float *__restrict__ xyz3,
                                                         const float *__restrict__ c, int M3, int K,
                                                         float shift_max, float r_min, float r_max,
                                                         F3 anchor, float *__restrict__ cand,
                                                         float *__restrict__ radius) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const float *ci = c + ((size_t)b * K + i) * 6;
    const float *p = xyz3 + ((size_t)b * M3 + i) * 3;
    float *o = cand + ((size_t)b * K + i) * 3;
    float sz[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float sh = ci[d];
        sh = sh < -shift_max ? -shift_max : sh;
        sh = sh > shift_max ? shift_max : sh;
        o[d] = p[d] + sh;
        float s = ci[3 + d];
        s = s < -1.0f ? -1.0f : s;
        s = s > 1.0f ? 1.0f : s;
        const float t1 = s * s;
        const float t2 = 0.5f * t1;
        const float t3 = 1.0f + s;
        const float q = t3 + t2;
        sz[d] = anchor.v[d] * q;
    }
    const float ll = sz[0] * sz[0], ww = sz[1] * sz[1], hh = sz[2] * sz[2];
    float sum = ll + ww;
    sum = sum + hh;
    float r = 0.5f * sqrtf(sum);  // correctly rounded (IEEE) under hipcc defaults, as on the host
    r = r < r_min ? r_min : r;
    r = r > r_max ? r_max : r;
    radius[(size_t)b * K + i] = r;
}

__global__ __launch_bounds__(256) void decode_kernel(const float *__restrict__ cand,
                                                     const float *__restrict__ o, int total, F9 anchors,
                                                     float *__restrict__ boxes) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const float *oi = o + (size_t)t * 10;
    const float *p = cand + (size_t)t * 3;
    float *bx = boxes + (size_t)t * 9;
    int label = 0;
    float best = oi[0];
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (oi[k] > best) { best = oi[k]; label = k; }
    const float score = 1.0f / (1.0f + expf(-best));
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        bx[d] = p[d] + oi[3 + d];
        float e = oi[6 + d];
        e = e < -2.0f ? -2.0f : e;
        e = e > 2.0f ? 2.0f : e;
        const float a = label == 0 ? anchors.v[d] : (label == 1 ? anchors.v[3 + d] : anchors.v[6 + d]);
        bx[3 + d] = a * expf(e);
    }
    bx[6] = oi[9];
    bx[7] = score;
    bx[8] = (float)label;
}

}  // namespace

SAD_API int sad_candidates_f32(const float *xyz3, const float *c, int B, int M3, int K,
                               float shift_max, float r_min, float r_max, const float *anchor,
                               float *cand, float *radius, sad_stream_t stream) {
    SAD_REQUIRE(xyz3 && c && anchor && cand && radius, "sad_candidates_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && B <= 65535 && K >= 1 && K <= M3, "sad_candidates_f32: need 1 <= K <= M3 (K=%d M3=%d)", K, M3);
    F3 a{{anchor[0], anchor[1], anchor[2]}};
    dim3 grid((K + 255) / 256, B);
    hipLaunchKernelGGL(candidates_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz3, c, M3, K,
                       shift_max, r_min, r_max, a, cand, radius);
    return sad::check_launch("sad_candidates_f32");
}

SAD_API int sad_decode_boxes_f32(const float *cand, const float *o, int B, int K, const float *anchors,
                                 float *boxes, sad_stream_t stream) {
    SAD_REQUIRE(cand && o && anchors && boxes, "sad_decode_boxes_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && K >= 1 && (long long)B * K < (1LL << 31), "sad_decode_boxes_f32: bad sizes");
    F9 a;
    for (int i = 0; i < 9; ++i) a.v[i] = anchors[i];
    const int total = B * K;
    hipLaunchKernelGGL(decode_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       cand, o, total, a, boxes);
    return sad::check_launch("sad_decode_boxes_f32");
}

// ---- SPEC.md §17: ragged scenes -> fixed point count ------------------------------------------------------
// HBM-bound row copy (algorithmic bytes: n_points * C * 4 read + written per scene); one thread per output row,
// the source row is a pure integer function of (seed, scene, row), identical to io.fix_size and the oracle.
namespace {
__device__ __forceinline__ unsigned mix32(unsigned a) {
    a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
    return a;
}
__device__ __forceinline__ unsigned h17(unsigned seed, unsigned b, unsigned i) {
    return mix32(seed * 0x9E3779B1u + b * 0x85EBCA77u + i * 0xC2B2AE3Du + 0x27D4EB2Fu);
}
__global__ __launch_bounds__(256) void subsample_pad_kernel(const float *__restrict__ points, const int32_t *__restrict__ offsets,
                                                            int C, int n_points, unsigned seed, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_points) return;
    const long long o = offsets[b], n = (long long)offsets[b + 1] - o;
    float *dst = out + ((size_t)b * n_points + i) * C;
    if (n <= 0) {
        for (int c = 0; c < C; ++c) dst[c] = 0.f;
        return;
    }
    long long j;
    if (n == n_points) j = i;
    else if (n > n_points) j = ((long long)i * n + (long long)(h17(seed, (unsigned)b, 0xFFFFFFFFu) % (unsigned)n)) / n_points;
    else j = i < n ? i : (long long)(h17(seed, (unsigned)b, (unsigned)i) % (unsigned)n);
    const float *src = points + (size_t)(o + j) * C;
    if (C == 4 && (((uintptr_t)points | (uintptr_t)out) & 15) == 0) {
        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
    } else {
        for (int c = 0; c < C; ++c) dst[c] = src[c];
    }
}
}  // namespace

SAD_API int sad_subsample_pad_f32(const float *points, const int32_t *offsets, int B, int C, int n_points,
                                  unsigned seed, float *out, sad_stream_t stream) {
    SAD_REQUIRE(offsets && out, "sad_subsample_pad_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && B <= 65535 && C >= 1 && C <= 64 && n_points >= 1, "sad_subsample_pad_f32: need 1 <= B <= 65535, 1 <= C <= 64, n_points >= 1");
    hipLaunchKernelGGL(subsample_pad_kernel, dim3((unsigned)((n_points + 255) / 256), B), dim3(256), 0, (hipStream_t)stream,
                       points, offsets, C, n_points, seed, out);
    return sad::check_launch("sad_subsample_pad_f32");
}

// ---- strided row copy (the host side's replacement for framework copies on the step) -------------------------
namespace {
__global__ __launch_bounds__(256) void copy_rows_kernel(const unsigned *__restrict__ src, long long ss, unsigned *__restrict__ dst, long long ds,
                                                        long long total, unsigned row_words) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / row_words;
        const long long w = i - r * row_words;
        dst[r * ds + w] = src[r * ss + w];
    }
}
// rows of whole 16-byte chunks, both sides 16-byte aligned (the large copies: centroid prefixes, candidate rows)
__global__ __launch_bounds__(256) void copy_rows4_kernel(const uint4 *__restrict__ src, long long ss4, uint4 *__restrict__ dst, long long ds4,
                                                         long long total4, unsigned row4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long r = i / row4;
        const long long w = i - r * row4;
        dst[r * ds4 + w] = src[r * ss4 + w];
    }
}
}  // namespace

SAD_API int sad_copy_rows_u32(const void *src, long long src_stride_words, void *dst, long long dst_stride_words,
                              long long n_rows, long long row_words, sad_stream_t stream) {
    SAD_REQUIRE(src && dst, "sad_copy_rows_u32: NULL pointer");
    SAD_REQUIRE(n_rows >= 1 && row_words >= 1 && row_words <= 0x7FFFFFFFll && src_stride_words >= 0 && dst_stride_words >= row_words,
                "sad_copy_rows_u32: need n_rows >= 1, 1 <= row_words < 2^31, dst_stride_words >= row_words");
    SAD_REQUIRE(((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0, "sad_copy_rows_u32: pointers must be 4-byte aligned");
    const bool v4 = (row_words % 4 == 0) && (src_stride_words % 4 == 0) && (dst_stride_words % 4 == 0) &&
                    (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const long long total = v4 ? n_rows * (row_words / 4) : n_rows * row_words;
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)sad::device_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (v4)
        hipLaunchKernelGGL(copy_rows4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4 *>(src),
                           src_stride_words / 4, reinterpret_cast<uint4 *>(dst), dst_stride_words / 4, total, (unsigned)(row_words / 4));
    else
        hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned *>(src),
                           src_stride_words, reinterpret_cast<unsigned *>(dst), dst_stride_words, total, (unsigned)row_words);
    return sad::check_launch("sad_copy_rows_u32");
}
