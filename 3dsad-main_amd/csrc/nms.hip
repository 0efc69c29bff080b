// Rotated-box NMS in bird's-eye view for gfx950 (SPEC.md §13; SURVEY.md §8(f) row 1: the step right
// after the measured path).  No reference source exists (/root/reference/README.md:1-2).
//
// One workgroup per scene.  (1) rank the boxes by (score desc, index asc) with an O(K^2) counting
// rank (K <= 512, one box per thread); (2) thread p computes row p of the suppression matrix —
// bit q set iff IoU(rank p, rank q) > thr for q > p — with exact Sutherland-Hodgman clipping in
// binary32 (no contraction; sin/cos by the reproducible routine of SPEC §13, so the CPU oracle and
// this kernel make identical keep decisions); (3) one wave walks the ranking with 64-bit mask words.
#include "common.h"

namespace {

constexpr int NMS_MAXK = 512;
constexpr int NMS_WORDS = NMS_MAXK / 64;

__device__ __forceinline__ void sincos_r(float th, float &s_out, float &c_out) {
    const float n = rintf(th * 0.63661975f);
    float r = th - n * 1.5703125f;
    r = r - n * 4.8375129699707031e-4f;
    r = r - n * 7.5497899548918861e-8f;
    const int q = ((int)n) & 3;
    const float r2 = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * r2; ps = ps + 8.3321608736e-3f;
    ps = ps * r2; ps = ps + -1.6666654611e-1f;
    float S = r * r2; S = S * ps; S = r + S;
    float pc = 2.443315711809948e-5f;
    pc = pc * r2; pc = pc + -1.388731625493765e-3f;
    pc = pc * r2; pc = pc + 4.166664568298827e-2f;
    float C = r2 * r2; C = C * pc;
    const float h = 0.5f * r2;
    const float one = 1.0f - h;
    C = one + C;
    s_out = q == 0 ? S : (q == 1 ? C : (q == 2 ? -S : -C));
    c_out = q == 0 ? C : (q == 1 ? -S : (q == 2 ? -C : S));
}

__device__ __forceinline__ void box_corners(const float *bx, float *cx, float *cy) {
    float s, c;
    sincos_r(bx[6], s, c);
    const float hl = 0.5f * bx[3], hw = 0.5f * bx[4];
    const float dx[4] = {hl, -hl, -hl, hl}, dy[4] = {hw, hw, -hw, -hw};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = c * dx[k], b = s * dy[k];
        float t = bx[0] + a;
        cx[k] = t - b;
        a = s * dx[k]; b = c * dy[k];
        t = bx[1] + a;
        cy[k] = t + b;
    }
}

// area of (polygon a) ∩ (convex quad b); corners counter-clockwise; vertex lists live in LDS
// scratch vertex lists: element v of list L lives at sc[(L*10 + v) * STRIDE] (thread-interleaved LDS)
template <int STRIDE>
__device__ float poly_clip_area(const float *ax, const float *ay, const float *bxs, const float *bys,
                                float *sc) {
    // two vertex lists, A = lists 0/1 (x/y), B = lists 2/3; a pass reads one and writes the other (round 5: the copy back
    // and the integer modulo of the neighbour index are gone — same floating-point operations, same order, same results)
#define vx(L, i) sc[((L) * 20 + (i)) * STRIDE]
#define vy(L, i) sc[((L) * 20 + 10 + (i)) * STRIDE]
    int n = 4;
    int cur = 0;
    for (int i = 0; i < 4; ++i) { vx(0, i) = ax[i]; vy(0, i) = ay[i]; }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const float q0x = bxs[e], q0y = bys[e], q1x = bxs[(e + 1) & 3], q1y = bys[(e + 1) & 3];
        const float ex = q1x - q0x, ey = q1y - q0y;
        const int nxt = cur ^ 1;
        int m = 0;
        // the previous vertex of vertex 0 is vertex n - 1; afterwards it is the vertex just visited (kept in registers)
        float ppx = vx(cur, n - 1), ppy = vy(cur, n - 1);
        float cp;
        {
            const float a = ppy - q0y, b = ppx - q0x;
            const float t1 = ex * a, t2 = ey * b;
            cp = t1 - t2;
        }
        for (int i = 0; i < n; ++i) {
            const float cxi = vx(cur, i), cyi = vy(cur, i);
            const float a = cyi - q0y, b = cxi - q0x;
            const float t1 = ex * a, t2 = ey * b;
            const float cc = t1 - t2;
            const bool in_c = cc >= 0.0f, in_p = cp >= 0.0f;
            if (in_c != in_p) {
                const float den = cp - cc;
                const float t = cp / den;
                float d = cxi - ppx;
                d = t * d;
                vx(nxt, m) = ppx + d;
                d = cyi - ppy;
                d = t * d;
                vy(nxt, m) = ppy + d;
                ++m;
            }
            if (in_c) { vx(nxt, m) = cxi; vy(nxt, m) = cyi; ++m; }
            ppx = cxi; ppy = cyi; cp = cc;
        }
        n = m;
        cur = nxt;
    }
    if (n < 3) return 0.0f;
    float sum = 0.0f;
    const float x0 = vx(cur, 0), y0 = vy(cur, 0);
    float xi = x0, yi = y0;
    for (int i = 0; i < n; ++i) {
        const bool last = i + 1 == n;
        const float xj = last ? x0 : vx(cur, last ? 0 : i + 1), yj = last ? y0 : vy(cur, last ? 0 : i + 1);
        const float t1 = xi * yj, t2 = xj * yi;
        const float d = t1 - t2;
        sum = sum + d;
        xi = xj; yi = yj;
    }
    sum = sum < 0.0f ? -sum : sum;
    return 0.5f * sum;
#undef vx
#undef vy
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void nms_bev_kernel(const float *__restrict__ boxes, int K,
                                                          float iou_thr, float score_thr,
                                                          int32_t *__restrict__ keep,
                                                          int32_t *__restrict__ order,
                                                          int32_t *__restrict__ count) {
    __shared__ float s_cx[NMS_MAXK][4], s_cy[NMS_MAXK][4], s_area[NMS_MAXK], s_score[NMS_MAXK];
    __shared__ int s_rank2idx[NMS_MAXK];
    __shared__ unsigned long long s_mask[NMS_MAXK][NMS_WORDS];
    __shared__ float s_poly[4 * 10 * THREADS];   // per-thread clipping scratch, thread-interleaved
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const float *bx = boxes + (size_t)blockIdx.x * K * 9;
    int32_t *kp = keep + (size_t)blockIdx.x * K, *od = order + (size_t)blockIdx.x * K;

    for (int i = tid; i < K; i += THREADS) {
        s_score[i] = bx[i * 9 + 7];
        kp[i] = 0;
        od[i] = -1;
    }
    __syncthreads();
    // 1. rank among the candidates (score >= threshold): score descending, index ascending
    int nloc = 0;
    for (int i = tid; i < K; i += THREADS) {
        const float si = s_score[i];
        if (si >= score_thr) {
            int r = 0;
            for (int j = 0; j < K; ++j) {
                const float sj = s_score[j];
                r += (sj >= score_thr) && (sj > si || (sj == si && j < i));
            }
            s_rank2idx[r] = i;
            ++nloc;
        }
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    atomicAdd(&s_n, nloc);
    __syncthreads();
    const int n = s_n;
    // corners and areas in rank order
    for (int p = tid; p < n; p += THREADS) {
        const float *b = bx + (size_t)s_rank2idx[p] * 9;
        float cx[4], cy[4];
        box_corners(b, cx, cy);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_cx[p][k] = cx[k]; s_cy[p][k] = cy[k]; }
        s_area[p] = b[3] * b[4];
    }
    __syncthreads();
    // 2. suppression matrix
    for (int p = tid; p < n; p += THREADS) {
        unsigned long long w[NMS_WORDS];
#pragma unroll
        for (int k = 0; k < NMS_WORDS; ++k) w[k] = 0ull;
        for (int q = p + 1; q < n; ++q) {
            const float inter = poly_clip_area<THREADS>(s_cx[p], s_cy[p], s_cx[q], s_cy[q], s_poly + tid);
            float den = s_area[p] + s_area[q];
            den = den - inter;
            const float iou = den > 0.0f ? inter / den : 0.0f;
            if (iou > iou_thr) {
#pragma unroll
                for (int k = 0; k < NMS_WORDS; ++k)
                    if ((q >> 6) == k) w[k] |= 1ull << (q & 63);
            }
        }
#pragma unroll
        for (int k = 0; k < NMS_WORDS; ++k) s_mask[p][k] = w[k];
    }
    __syncthreads();
    // 3. walk the ranking (lanes 0..NMS_WORDS-1 of wave 0 hold one removed-word each)
    if (tid < 64) {
        unsigned long long removed = 0ull;
        int nk = 0;
        for (int p = 0; p < n; ++p) {
            const unsigned long long word = __shfl(removed, p >> 6, 64);
            if (!((word >> (p & 63)) & 1ull)) {            // wave-uniform
                if (tid == 0) {
                    const int i = s_rank2idx[p];
                    od[nk] = i;
                    kp[i] = 1;
                }
                ++nk;
                if (tid < NMS_WORDS) removed |= s_mask[p][tid];
            }
        }
        if (tid == 0) count[blockIdx.x] = nk;
    }
}

// ---- three-kernel variant (caller workspace): the K x K suppression matrix is spread over the chip --
// prep (one workgroup per scene): ranking, corners, areas -> workspace; mask (one WAVE per ranked
// box, 16 boxes per workgroup): lane l tests box p against boxes 64k + l, the ballot is the mask
// word; walk (one wave per scene).  Same arithmetic as nms_bev_kernel, same keep decisions.
struct NmsWs {            // per-scene slices of the workspace
    int *n;               // [4] (n, pad)
    int *rank2idx;        // [K]
    float *area;          // [K]
    float *cx, *cy;       // [K][4]
    unsigned long long *mask;   // [K][NMS_WORDS]
};
__host__ __device__ inline size_t nms_ws_scene_bytes(int K) {
    return 16 + (size_t)K * 4 * 2 + (size_t)K * 16 * 2 + (size_t)K * NMS_WORDS * 8;
}
__device__ __forceinline__ NmsWs nms_ws(void *base, int scene, int K) {
    unsigned char *q = (unsigned char *)base + (size_t)scene * nms_ws_scene_bytes(K);
    NmsWs w;
    w.n = (int *)q; q += 16;
    w.rank2idx = (int *)q; q += (size_t)K * 4;
    w.area = (float *)q; q += (size_t)K * 4;
    w.cx = (float *)q; q += (size_t)K * 16;
    w.cy = (float *)q; q += (size_t)K * 16;
    w.mask = (unsigned long long *)q;
    return w;
}

__global__ __launch_bounds__(256) void nms_prep_kernel(const float *__restrict__ boxes, int K, float score_thr,
                                                       int32_t *__restrict__ keep, int32_t *__restrict__ order,
                                                       void *__restrict__ wsbase) {
    __shared__ float s_score[NMS_MAXK];
    __shared__ int s_rank2idx[NMS_MAXK];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const float *bx = boxes + (size_t)blockIdx.x * K * 9;
    int32_t *kp = keep + (size_t)blockIdx.x * K, *od = order + (size_t)blockIdx.x * K;
    const NmsWs w = nms_ws(wsbase, blockIdx.x, K);
    for (int i = tid; i < K; i += 256) {
        s_score[i] = bx[i * 9 + 7];
        kp[i] = 0;
        od[i] = -1;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    int nloc = 0;
    for (int i = tid; i < K; i += 256) {
        const float si = s_score[i];
        if (si >= score_thr) {
            int r = 0;
            for (int j = 0; j < K; ++j) {
                const float sj = s_score[j];
                r += (sj >= score_thr) && (sj > si || (sj == si && j < i));
            }
            s_rank2idx[r] = i;
            ++nloc;
        }
    }
    atomicAdd(&s_n, nloc);
    __syncthreads();
    const int n = s_n;
    if (tid == 0) w.n[0] = n;
    for (int p = tid; p < n; p += 256) {
        const int i = s_rank2idx[p];
        const float *b = bx + (size_t)i * 9;
        float cx[4], cy[4];
        box_corners(b, cx, cy);
#pragma unroll
        for (int k = 0; k < 4; ++k) { w.cx[p * 4 + k] = cx[k]; w.cy[p * 4 + k] = cy[k]; }
        w.area[p] = b[3] * b[4];
        w.rank2idx[p] = i;
    }
}

constexpr int NMS_ROWS_PER_WG = 16;
__global__ __launch_bounds__(256) void nms_mask_kernel(int K, float iou_thr, void *__restrict__ wsbase) {
    __shared__ float s_poly[4 * 10 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const NmsWs w = nms_ws(wsbase, blockIdx.y, K);
    const int n = w.n[0];
    for (int rr = wave; rr < NMS_ROWS_PER_WG; rr += 4) {
        const int p = blockIdx.x * NMS_ROWS_PER_WG + rr;
        if (p >= n) continue;                                   // wave-uniform
        float pcx[4], pcy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pcx[k] = w.cx[p * 4 + k]; pcy[k] = w.cy[p * 4 + k]; }
        const float pa = w.area[p];
        for (int k = 0; k < NMS_WORDS; ++k) {
            unsigned long long word = 0ull;
            if (64 * k + 63 > p && 64 * k < n) {                // some q > p in this chunk
                const int q = 64 * k + lane;
                bool sup = false;
                if (q > p && q < n) {
                    float qcx[4], qcy[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { qcx[c] = w.cx[q * 4 + c]; qcy[c] = w.cy[q * 4 + c]; }
                    const float inter = poly_clip_area<256>(pcx, pcy, qcx, qcy, s_poly + tid);
                    float den = pa + w.area[q];
                    den = den - inter;
                    const float iou = den > 0.0f ? inter / den : 0.0f;
                    sup = iou > iou_thr;
                }
                word = __ballot(sup);
            }
            if (lane == 0) w.mask[(size_t)p * NMS_WORDS + k] = word;
        }
    }
}

// Greedy walk of the ranking, 64 ranks at a time (round 5; the first version took one dependent global load and one
// cross-lane read per kept box: 79 us for 32 scenes of 256 boxes).  Lane i of chunk c holds row p = 64c + i of the matrix.
// Inside a chunk the decision chain runs on a 64-bit scalar mask (who is still alive) and lane reads of the chunk's own
// word; what the chunk's kept boxes suppress in LATER chunks is or-ed into per-chunk words in LDS.  Same greedy rule:
// box p is kept iff no kept box of higher rank suppresses it.
__global__ __launch_bounds__(64) void nms_walk_kernel(int K, int32_t *__restrict__ keep, int32_t *__restrict__ order,
                                                      int32_t *__restrict__ count, void *__restrict__ wsbase) {
    __shared__ unsigned long long s_rem[NMS_WORDS];
    const int lane = threadIdx.x;
    const NmsWs w = nms_ws(wsbase, blockIdx.x, K);
    int32_t *kp = keep + (size_t)blockIdx.x * K, *od = order + (size_t)blockIdx.x * K;
    const int n = __builtin_amdgcn_readfirstlane(w.n[0]);          // (wave-uniform by construction: tell the compiler)
    const int nch = (n + 63) >> 6;
    if (lane < NMS_WORDS) s_rem[lane] = 0ull;
    __syncthreads();
    int nk = 0;
    for (int c = 0; c < nch; ++c) {                                  // (wave-uniform)
        const int p = 64 * c + lane;
        const bool have = p < n;
        const unsigned long long own = have ? w.mask[(size_t)p * NMS_WORDS + c] : 0ull;     // suppressed by p inside its chunk
        const int idx = have ? w.rank2idx[p] : 0;
        const int left = n - 64 * c;
        const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        const unsigned long long rem_lo = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(s_rem[c] & 0xFFFFFFFFull));
        const unsigned long long rem_hi = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(s_rem[c] >> 32));
        unsigned long long alive = valid & ~(rem_lo | (rem_hi << 32));
        unsigned long long kept = 0ull;
        // only boxes whose own word is non-zero change anything for their chunk mates: jump from one such box to the next,
        // everything alive in between is kept as it stands (a scalar chain of a few steps per chunk: ctz, two lane reads, masks)
        const unsigned long long nz = __ballot(own != 0ull);
        const unsigned own_lo = (unsigned)own, own_hi = (unsigned)(own >> 32);
        while (true) {
            const unsigned long long cand = alive & nz;
            if (cand == 0ull) {
                kept |= alive;
                break;
            }
            const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cand));
            const unsigned long long upto = (2ull << i) - 1ull;                  // ranks 0 .. i of the chunk
            kept |= alive & upto;                                              // i itself (alive), and the inert ones below it
            const unsigned long long r = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)own_lo, i) |
                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)own_hi, i) << 32);
            alive &= ~(r | upto);
        }
        const bool mine = (kept >> lane) & 1ull;
        if (mine) {
            od[nk + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = idx;
            kp[idx] = 1;
            for (int k = c + 1; k < nch; ++k) {                      // later chunks: usually nothing
                const unsigned long long wk = w.mask[(size_t)p * NMS_WORDS + k];
                if (wk) atomicOr(&s_rem[k], wk);
            }
        }
        nk += __builtin_popcountll(kept);
        __syncthreads();                                             // (one wave: orders the LDS atomics before the next chunk's read)
    }
    if (lane == 0) count[blockIdx.x] = nk;
}

}  // namespace

SAD_API int sad_nms_bev_f32(const float *boxes, int B, int K, float iou_thr, float score_thr,
                            int32_t *keep, int32_t *order, int32_t *count, sad_stream_t stream) {
    SAD_REQUIRE(boxes && keep && order && count, "sad_nms_bev_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && K >= 1, "sad_nms_bev_f32: need B,K >= 1");
    if (K > NMS_MAXK) return sad::fail(SAD_EUNSUPPORTED, "sad_nms_bev_f32: K=%d > %d", K, NMS_MAXK);
    hipLaunchKernelGGL((nms_bev_kernel<256>), dim3(B), dim3(256), 0, (hipStream_t)stream, boxes, K, iou_thr,
                       score_thr, keep, order, count);
    return sad::check_launch("sad_nms_bev_f32");
}

SAD_API size_t sad_nms_bev_workspace_bytes(int B, int K) {
    if (B < 1 || K < 1 || K > NMS_MAXK) return 0;
    return (size_t)B * nms_ws_scene_bytes(K);
}

SAD_API int sad_nms_bev_ws_f32(const float *boxes, int B, int K, float iou_thr, float score_thr,
                               int32_t *keep, int32_t *order, int32_t *count, void *workspace, sad_stream_t stream) {
    SAD_REQUIRE(boxes && keep && order && count && workspace, "sad_nms_bev_ws_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && B <= 65535 && K >= 1, "sad_nms_bev_ws_f32: need 1 <= B <= 65535, K >= 1");
    SAD_REQUIRE((uintptr_t)workspace % 16 == 0, "sad_nms_bev_ws_f32: workspace must be 16-byte aligned");
    if (K > NMS_MAXK) return sad::fail(SAD_EUNSUPPORTED, "sad_nms_bev_ws_f32: K=%d > %d", K, NMS_MAXK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_prep_kernel, dim3(B), dim3(256), 0, st, boxes, K, score_thr, keep, order, workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3((K + NMS_ROWS_PER_WG - 1) / NMS_ROWS_PER_WG, B), dim3(256), 0, st, K, iou_thr, workspace);
    hipLaunchKernelGGL(nms_walk_kernel, dim3(B), dim3(64), 0, st, K, keep, order, count, workspace);
    return sad::check_launch("sad_nms_bev_ws_f32");
}
