// Ball query (fixed, multi-radius and per-centroid adaptive radius) and kNN for gfx950
// (SPEC.md §3, §4).  No reference source exists (/root/reference/README.md:1-2).
//
// ball_query: one wave64 owns CW centroids.  Lanes stride over the scene's points in index order
// (64 points per step, one per lane); for every (centroid, radius) pair the in-radius predicate is
// turned into a 64-bit ballot, each accepted lane's output slot is count_so_far + popcount of the
// lower accepted lanes, and the running count is a wave-uniform scalar — so the first `nsample`
// accepted indices land in ascending index order without sorting, exactly as the sequential scan
// of SPEC.md §3 produces them.  d2 is evaluated once per pair and shared by all radii.
#include "common.h"

namespace {

struct BQParams {
    float radii[SAD_MAX_RADII];
    int nsample[SAD_MAX_RADII];
    int32_t *idx[SAD_MAX_RADII];
    int32_t *cnt[SAD_MAX_RADII];   // optional: number of accepted points per centroid, capped at nsample
};

constexpr int BQ_WAVES = 4;

template <int NR, int CW>
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ radius_pc, BQParams prm, int N, int M) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int m0 = (blockIdx.x * BQ_WAVES + wave) * CW;
    const float *p = xyz + (size_t)b * N * 3;

    float cx[CW], cy[CW], cz[CW], r2[CW][NR];
    int cnt[CW][NR], first[CW][NR];
    bool active[CW];
    int ndone = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        active[c] = (m0 + c) < M;
        const int m = active[c] ? m0 + c : M - 1;
        const float *q = new_xyz + ((size_t)b * M + m) * 3;
        cx[c] = q[0];
        cy[c] = q[1];
        cz[c] = q[2];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float rad = prm.radii[r];
            if (radius_pc) rad = rad * radius_pc[(size_t)b * M + m];
            r2[c][r] = rad * rad;
            first[c][r] = 0;
            cnt[c][r] = active[c] ? 0 : prm.nsample[r];
            if (!active[c]) ++ndone;
        }
    }

    for (int base = 0; base < N && ndone < CW * NR; base += 64) {
        const int j = base + lane;
        const bool valid = j < N;
        const int jj = valid ? j : N - 1;
        const float px = p[jj * 3 + 0], py = p[jj * 3 + 1], pz = p[jj * 3 + 2];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float d = sad::d2f(px, py, pz, cx[c], cy[c], cz[c]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool in = valid && (d < r2[c][r]);
                const unsigned long long mask = __ballot(in);
                if (mask != 0ull && cnt[c][r] < prm.nsample[r]) {  // wave-uniform
                    const int slot = cnt[c][r] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (in && slot < prm.nsample[r])
                        prm.idx[r][((size_t)b * M + (m0 + c)) * prm.nsample[r] + slot] = j;
                    if (cnt[c][r] == 0) first[c][r] = base + __builtin_ctzll(mask);
                    cnt[c][r] += __builtin_popcountll(mask);
                    if (cnt[c][r] >= prm.nsample[r]) ++ndone;
                }
            }
        }
    }
    // SPEC.md §3 padding: remaining slots repeat the first accepted index (0 if none).
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        if (!active[c]) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int S = prm.nsample[r];
            const int n = cnt[c][r] < S ? cnt[c][r] : S;
            for (int s = n + lane; s < S; s += 64)
                prm.idx[r][((size_t)b * M + (m0 + c)) * S + s] = first[c][r];
            if (prm.cnt[r] && lane == 0) prm.cnt[r][(size_t)b * M + (m0 + c)] = n;
        }
    }
}

template <int NR>
void launch_bq(const float *xyz, const float *new_xyz, const float *radius_pc, const BQParams &prm,
               int B, int N, int M, hipStream_t st) {
    constexpr int CW = 4;
    dim3 grid((M + BQ_WAVES * CW - 1) / (BQ_WAVES * CW), B);
    hipLaunchKernelGGL((ball_query_kernel<NR, CW>), grid, dim3(BQ_WAVES * 64), 0, st, xyz, new_xyz,
                       radius_pc, prm, N, M);
}

// ---- kNN: one wave per centroid; lane i holds the i-th best (d2, j) so far -------------------
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ xyz,
                                                  const float *__restrict__ new_xyz, int N, int M,
                                                  int K, int32_t *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    const float *p = xyz + (size_t)b * N * 3;
    const float *q = new_xyz + ((size_t)b * M + m) * 3;
    const float cx = q[0], cy = q[1], cz = q[2];
    float ld = __builtin_inff();  // sorted list, ascending (d2, j); lanes >= K stay +inf
    int lj = 0;
    float worst = __builtin_inff();
    for (int base = 0; base < N; base += 64) {
        const int j = base + lane;
        const bool valid = j < N;
        const int jj = valid ? j : N - 1;
        const float d = sad::d2f(p[jj * 3 + 0], p[jj * 3 + 1], p[jj * 3 + 2], cx, cy, cz);
        unsigned long long mask = __ballot(valid && d < worst);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float xd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), l));
            if (!(xd < worst)) continue;  // the list tightened since the ballot
            // equal d2 with a lower index is already in the list -> the new entry goes after it
            const int pos = __builtin_popcountll(__ballot(ld <= xd));
            const float ud = __shfl_up(ld, 1, 64);
            const int uj = __shfl_up(lj, 1, 64);
            if (lane > pos) {
                ld = ud;
                lj = uj;
            } else if (lane == pos) {
                ld = xd;
                lj = base + l;
            }
            if (lane >= K) ld = __builtin_inff();
            worst = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), K - 1));
        }
    }
    if (lane < K) idx[((size_t)b * M + m) * K + lane] = lj;
}

int check_common(const char *fn, const void *xyz, const void *new_xyz, int B, int N, int M) {
    if (!xyz || !new_xyz) return sad::fail(SAD_EINVAL, "%s: NULL pointer", fn);
    if (B < 1 || N < 1 || M < 1) return sad::fail(SAD_EINVAL, "%s: need B,N,M >= 1 (B=%d N=%d M=%d)", fn, B, N, M);
    if (B > 65535) return sad::fail(SAD_EUNSUPPORTED, "%s: B=%d > 65535", fn, B);
    if ((size_t)N * 3 >= (1u << 31)) return sad::fail(SAD_EUNSUPPORTED, "%s: N too large", fn);
    return SAD_OK;
}

}  // namespace

SAD_API int sad_ball_query_multi_f32(const float *xyz, const float *new_xyz, int n_radii,
                                     const float *radii, const float *radius_pc,
                                     const int *nsamples, int32_t *const *idx, int32_t *const *cnt,
                                     int B, int N, int M, sad_stream_t stream) {
    if (int e = check_common("sad_ball_query_multi_f32", xyz, new_xyz, B, N, M)) return e;
    SAD_REQUIRE(n_radii >= 1 && n_radii <= SAD_MAX_RADII, "sad_ball_query_multi_f32: n_radii=%d not in 1..%d", n_radii, SAD_MAX_RADII);
    SAD_REQUIRE(radii && nsamples && idx, "sad_ball_query_multi_f32: NULL parameter array");
    BQParams prm{};
    for (int r = 0; r < n_radii; ++r) {
        SAD_REQUIRE(nsamples[r] >= 1 && nsamples[r] <= 64, "sad_ball_query: nsample=%d not in 1..64", nsamples[r]);
        SAD_REQUIRE(idx[r], "sad_ball_query: NULL idx output");
        prm.radii[r] = radii[r];
        prm.nsample[r] = nsamples[r];
        prm.idx[r] = idx[r];
        prm.cnt[r] = cnt ? cnt[r] : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (n_radii) {
        case 1: launch_bq<1>(xyz, new_xyz, radius_pc, prm, B, N, M, st); break;
        case 2: launch_bq<2>(xyz, new_xyz, radius_pc, prm, B, N, M, st); break;
        case 3: launch_bq<3>(xyz, new_xyz, radius_pc, prm, B, N, M, st); break;
        default: launch_bq<4>(xyz, new_xyz, radius_pc, prm, B, N, M, st); break;
    }
    return sad::check_launch("sad_ball_query");
}

SAD_API int sad_ball_query_f32(const float *xyz, const float *new_xyz, float radius,
                               const float *radius_pc, int B, int N, int M, int S, int32_t *idx,
                               sad_stream_t stream) {
    const float radii[1] = {radius_pc ? 1.0f : radius};  // 1.0f * r == r exactly
    const int ns[1] = {S};
    int32_t *const outs[1] = {idx};
    return sad_ball_query_multi_f32(xyz, new_xyz, 1, radii, radius_pc, ns, outs, nullptr, B, N, M, stream);
}

SAD_API int sad_knn_f32(const float *xyz, const float *new_xyz, int B, int N, int M, int K,
                        int32_t *idx, sad_stream_t stream) {
    if (int e = check_common("sad_knn_f32", xyz, new_xyz, B, N, M)) return e;
    SAD_REQUIRE(idx, "sad_knn_f32: NULL idx");
    SAD_REQUIRE(K >= 1 && K <= 64 && K <= N, "sad_knn_f32: need 1 <= K <= min(64, N) (K=%d N=%d)", K, N);
    dim3 grid((M + 3) / 4, B);
    hipLaunchKernelGGL(knn_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, M, K, idx);
    return sad::check_launch("sad_knn_f32");
}
