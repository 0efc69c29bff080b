// Bucketed farthest point sampling for gfx950 (SPEC.md §2): the same indices as the plain scan,
// with most of the per-step distance updates skipped.  No reference source exists
// (/root/reference/README.md:1-2).
//
// Idea (exact, not approximate).  Points are first binned into Z-order grid cells so that each
// thread owns PPT spatially coherent points and each wave a compact region.  A sampling step with
// new centre c changes min_dist[p] only if d2(p,c) < min_dist[p].  For a box B containing the
// points of a thread (or wave), q = clamp(c, B) is the closest point of the box, and in binary32 the
// SPEC §1 expression is monotone in each |coordinate difference|, so d2f(q,c) <= d2f(p,c) for every
// p in B.  Hence if d2f(q,c) >= max_{p in B} min_dist[p], nothing in B changes and its cached
// (max distance, index) key is still valid.  After a few hundred samples almost every wave is
// skipped on every step, and what remains per step is the serial chain: publish key -> barrier ->
// 16-key arg-max -> read the winner's coordinates from LDS.
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int SORT_T = 1024;
constexpr int SORT_CELLS = 16384;   // 32 x 32 (Z-order) x 16

__device__ __forceinline__ int morton_cell(int ix, int iy, int iz) {
    int m = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) m |= (((ix >> i) & 1) << (2 * i)) | (((iy >> i) & 1) << (2 * i + 1));
    return (m << 4) | iz;
}

// One workgroup per scene: perm[pos] = original index of the pos-th point in Z-order cell order.
__global__ __launch_bounds__(SORT_T) void fps_sort_kernel(const float *__restrict__ xyz, int N,
                                                          int *__restrict__ perm_out) {
    extern __shared__ int hist[];               // SORT_CELLS ints
    __shared__ float red[6][16];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    int *perm = perm_out + (size_t)blockIdx.x * N;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int j = tid; j < N; j += SORT_T)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = p[j * 3 + d];
            lo[d] = v < lo[d] ? v : lo[d];
            hi[d] = v > hi[d] ? v : hi[d];
        }
#pragma unroll
    for (int d = 0; d < 3; ++d)
        for (int off = 32; off >= 1; off >>= 1) {
            const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
            lo[d] = a < lo[d] ? a : lo[d];
            hi[d] = b > hi[d] ? b : hi[d];
        }
    if (lane == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[d][wave] = lo[d]; red[3 + d][wave] = hi[d]; }
    for (int c = tid; c < SORT_CELLS; c += SORT_T) hist[c] = 0;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d)
        for (int w = 0; w < 16; ++w) {
            lo[d] = red[d][w] < lo[d] ? red[d][w] : lo[d];
            hi[d] = red[3 + d][w] > hi[d] ? red[3 + d][w] : hi[d];
        }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const float sx = ex > 0.f ? 32.0f / ex : 0.f, sy = ey > 0.f ? 32.0f / ey : 0.f, sz = ez > 0.f ? 16.0f / ez : 0.f;
    auto cell_of = [&](int j) {
        int ix = (int)((p[j * 3 + 0] - lo[0]) * sx), iy = (int)((p[j * 3 + 1] - lo[1]) * sy), iz = (int)((p[j * 3 + 2] - lo[2]) * sz);
        ix = ix < 0 ? 0 : (ix > 31 ? 31 : ix);
        iy = iy < 0 ? 0 : (iy > 31 ? 31 : iy);
        iz = iz < 0 ? 0 : (iz > 15 ? 15 : iz);
        return morton_cell(ix, iy, iz);
    };
    for (int j = tid; j < N; j += SORT_T) atomicAdd(&hist[cell_of(j)], 1);
    __syncthreads();
    constexpr int CPT = SORT_CELLS / SORT_T;    // 16 cells per thread
    int loc[CPT];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        loc[k] = sum;
        sum += hist[tid * CPT + k];
    }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const int tbase = wbase + incl - sum;
#pragma unroll
    for (int k = 0; k < CPT; ++k) hist[tid * CPT + k] = tbase + loc[k];
    __syncthreads();
    for (int j = tid; j < N; j += SORT_T) perm[atomicAdd(&hist[cell_of(j)], 1)] = j;
}

template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 umax64(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 row_max_u64(u64 k) {
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
__device__ __forceinline__ u64 wave_max_u64(u64 k) {
    k = row_max_u64(k);
    return umax64(umax64(readlane_u64(k, 0), readlane_u64(k, 16)), umax64(readlane_u64(k, 32), readlane_u64(k, 48)));
}
// 32-bit building blocks: a 64-bit key max is done as max(high words), then max of the low words
// among the lanes that hold that high word — two cheap v_max_u32 butterflies instead of 64-bit
// compare/select chains.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xF, 0xF, false);
    return o > v ? o : v;
}
template <int STEPS>
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {   // max over aligned groups of 2^STEPS lanes (<= 16)
    if constexpr (STEPS >= 1) v = dpp_max_u32<0xB1>(v);
    if constexpr (STEPS >= 2) v = dpp_max_u32<0x4E>(v);
    if constexpr (STEPS >= 3) v = dpp_max_u32<0x141>(v);
    if constexpr (STEPS >= 4) v = dpp_max_u32<0x140>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {  // wave-uniform result
    v = row_max_u32<4>(v);
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned wave_max_u32_bcast(unsigned v) {   // same, cross-row part by row_bcast DPP
    v = row_max_u32<4>(v);
    unsigned o = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false);            // row_bcast:31 -> rows 2,3
    v = o > v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}
__device__ __forceinline__ float rdl_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_bucket_kernel(const float *__restrict__ xyz,
                                                             const int *__restrict__ perm_in, int N,
                                                             int M, int *__restrict__ idx_out) {
    constexpr int NW = THREADS / 64;
    typedef float fvec __attribute__((ext_vector_type(PPT)));
    typedef int ivec __attribute__((ext_vector_type(PPT)));
    __shared__ __attribute__((aligned(16))) unsigned s_rec[2][16][8];   // per wave: key lo, key hi, x, y, z
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    const int *perm = perm_in + (size_t)blockIdx.x * N;
    int *out = idx_out + (size_t)blockIdx.x * M;

    fvec px, py, pz, md;
    ivec oi;
    float tlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, thi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pos = tid * PPT + k;
        if (pos < N) {
            const int j = perm[pos];
            px[k] = p[j * 3 + 0];
            py[k] = p[j * 3 + 1];
            pz[k] = p[j * 3 + 2];
            md[k] = __builtin_inff();
            oi[k] = j;
            tlo[0] = px[k] < tlo[0] ? px[k] : tlo[0]; thi[0] = px[k] > thi[0] ? px[k] : thi[0];
            tlo[1] = py[k] < tlo[1] ? py[k] : tlo[1]; thi[1] = py[k] > thi[1] ? py[k] : thi[1];
            tlo[2] = pz[k] < tlo[2] ? pz[k] : tlo[2]; thi[2] = pz[k] > thi[2] ? pz[k] : thi[2];
        } else {  // padding: distance 0 and the largest index never beat a real point
            px[k] = py[k] = pz[k] = 0.f;
            md[k] = 0.f;
            oi[k] = 0x7fffffff;
        }
    }
    // per-thread cache: largest min-distance, its slot and key (ties -> lowest original index)
    float tmax;
    int tbk;
    u64 tkey;
    auto thread_best = [&]() {
        float bd = md[0];
        int bk = 0, bo = oi[0];
#pragma unroll
        for (int k = 1; k < PPT; ++k) {
            const bool t = md[k] > bd || (md[k] == bd && oi[k] < bo);
            bd = t ? md[k] : bd;
            bo = t ? oi[k] : bo;
            bk = t ? k : bk;
        }
        tmax = bd;
        tbk = bk;
        tkey = ((u64)__builtin_bit_cast(unsigned, bd) << 32) | (unsigned)(~(unsigned)bo);
    };
    thread_best();
    // wave bounding box (fixed) and wave cache: key + coordinates of the wave's best point
    float wlo[3], whi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float a = tlo[d], b = thi[d];
        for (int off = 32; off >= 1; off >>= 1) {
            const float a2 = __shfl_xor(a, off, 64), b2 = __shfl_xor(b, off, 64);
            a = a2 < a ? a2 : a;
            b = b2 > b ? b2 : b;
        }
        wlo[d] = rdl_f(a, 0);
        whi[d] = rdl_f(b, 0);
    }
    u64 wkey;
    float wx, wy, wz, wmaxf;
    auto wave_best = [&]() {
        const unsigned hi = wave_max_u32((unsigned)(tkey >> 32));
        const unsigned lo = wave_max_u32((unsigned)(tkey >> 32) == hi ? (unsigned)tkey : 0u);
        wkey = ((u64)hi << 32) | lo;
        const unsigned long long own = __ballot(tkey == wkey);     // exactly one lane (indices are unique)
        const int ol = __builtin_ctzll(own);
        const int kk = __builtin_amdgcn_readlane(tbk, ol);
        wx = rdl_f(px[kk], ol);
        wy = rdl_f(py[kk], ol);
        wz = rdl_f(pz[kk], ol);
        wmaxf = __builtin_bit_cast(float, hi);
    };
    wave_best();

    if (tid == 0) out[0] = 0;
    float cx = p[0], cy = p[1], cz = p[2];
    for (int i = 1; i < M; ++i) {
        // wave-level skip test (wave-uniform values)
        const float dqw = sad::d2f(clampf(cx, wlo[0], whi[0]), clampf(cy, wlo[1], whi[1]), clampf(cz, wlo[2], whi[2]), cx, cy, cz);
        if (dqw < wmaxf) {
            const float dqt = sad::d2f(clampf(cx, tlo[0], thi[0]), clampf(cy, tlo[1], thi[1]), clampf(cz, tlo[2], thi[2]), cx, cy, cz);
            if (__any(dqt < tmax)) {
                const fvec dx = px - cx, dy = py - cy, dz = pz - cz;
                const fvec d = (dx * dx + dy * dy) + dz * dz;   // SPEC.md §1 order, no contraction
                md = __builtin_elementwise_min(md, d);
                thread_best();
                wave_best();
            }
        }
        const int buf = i & 1;
        if (lane == 0) {
            uint4 r0;
            r0.x = (unsigned)wkey;
            r0.y = (unsigned)(wkey >> 32);
            r0.z = __builtin_bit_cast(unsigned, wx);
            r0.w = __builtin_bit_cast(unsigned, wy);
            *reinterpret_cast<uint4 *>(&s_rec[buf][wave][0]) = r0;
            s_rec[buf][wave][4] = __builtin_bit_cast(unsigned, wz);
        }
        __syncthreads();
        // every lane fetches one wave's record (lane & (NW-1)); the winner's coordinates then come
        // out of the winning lane's registers — no second LDS round trip
        const uint4 rec = *reinterpret_cast<const uint4 *>(&s_rec[buf][lane & (NW - 1)][0]);
        const unsigned recz = s_rec[buf][lane & (NW - 1)][4];
        constexpr int RSTEPS = NW == 16 ? 4 : (NW == 8 ? 3 : 2);
        const unsigned ghi = __builtin_amdgcn_readlane(row_max_u32<RSTEPS>(rec.y), 0);
        const unsigned glo = __builtin_amdgcn_readlane(row_max_u32<RSTEPS>(rec.y == ghi ? rec.x : 0u), 0);
        const int slot = __builtin_ctzll(__ballot(rec.y == ghi && rec.x == glo));
        cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rec.z, slot));
        cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rec.w, slot));
        cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(recz, slot));
        const u64 gk = ((u64)ghi << 32) | glo;
        if (tid == 0) out[i] = (int)(~(unsigned)gk);
    }
}


// ---- cell-bucket kernel ------------------------------------------------------------------------
// Finer buckets, balanced over the waves.  A bucket is 64 consecutive points of the Z-order
// permutation — slot k of wave w holds bucket k*NW + w, one point per lane — so spatially adjacent
// buckets sit in DIFFERENT waves (and SIMDs), and a sampling step that disturbs a handful of
// buckets costs each wave one or two 64-point updates instead of a whole-wave update.  Lane k
// (< PPT <= 32) of a wave keeps bucket k's state: bounding box, exact max min-distance, and the key
// + coordinates of the point that attains it.  The skip test of all PPT buckets is one
// lane-parallel evaluation; its ballot is walked with scalar bit scans; slot k's registers are
// reached by uniform register indexing (s_set_gpr_idx), so there is ONE update body and no
// per-slot code; the wave's best is a row reduction of the bucket keys.  Few, fat waves (8 x 32
// slots for 16 384 points) keep the part of the step every wave must execute — test, publish,
// barrier, NW-key arg-max — short.  Same proof of exactness as above, per bucket.
// v_writelane_b32 through the LLVM intrinsic (clang has no builtin for it): the compiler then owns m0 — it puts the lane select there
// (gfx9: one SGPR on the constant bus), merges the m0 writes of consecutive writelanes with the same lane, and keeps them apart from the
// s_set_gpr_idx / M0 traffic of the register-indexed vectors.  (An asm statement with "s_mov_b32 m0" cannot declare the clobber: m0 is a
// reserved register and the clobber is ignored with -Winline-asm.)
extern "C" __device__ unsigned sad_writelane(unsigned val_uniform, unsigned lane_uniform, unsigned old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ unsigned wrl_dyn(unsigned old, unsigned val_uniform, int lane_uniform) {
    // v_writelane_b32 with data and lane select both scalar: the lane select goes through M0
    // (gfx9 allows one SGPR on the constant bus)
    val_uniform = __builtin_amdgcn_readfirstlane(val_uniform);
    lane_uniform = __builtin_amdgcn_readfirstlane(lane_uniform);
    // (the builtin, not an asm statement with "s_mov_b32 m0": m0 is a reserved register, a clobber of it is ignored
    // (-Winline-asm), and hipcc lowers register-indexed vectors through s_set_gpr_idx / M0 in these kernels — the compiler
    // has to see every write of m0)
    return sad_writelane(val_uniform, lane_uniform, old);
}
__device__ __forceinline__ float wrl_dyn_f(float old, float val_uniform, int lane_uniform) {
    return __builtin_bit_cast(float, wrl_dyn(__builtin_bit_cast(unsigned, old), __builtin_bit_cast(unsigned, val_uniform), lane_uniform));
}
template <int STEPS>
__device__ __forceinline__ unsigned half_max_u32(unsigned v) {   // max over lanes 0..2^STEPS-1 (<= 32), valid in lane 0
    if constexpr (STEPS <= 4) return row_max_u32<STEPS>(v);
    v = row_max_u32<4>(v);
    const unsigned o = __builtin_amdgcn_readlane(v, 16);
    return o > v ? o : v;
}

#ifdef SAD_FPS_STAMPS
// measurement build: per wave of workgroup 0, cycles (s_memtime) summed over the second half of the steps:
// [0] skip test, [1] bucket updates + wave best, [2] publish up to the barrier, [3] barrier wait, [4] read + broadcast,
// [5] steps in which the wave was active, [6] buckets updated, [7] steps
__device__ unsigned long long g_fpst[16 * 8];
extern "C" __attribute__((visibility("default"))) int sad_debug_read_fps_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fpst), sizeof(unsigned long long) * 16 * 8);
}
#define FPS_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FPS_ACC(slot, expr) do { if (sample) acc_##slot += (expr); } while (0)
#else
#define FPS_T(v)
#define FPS_ACC(slot, expr)
#endif

template <int NW, int PPT>
__device__ __forceinline__ void fps_cell_body(const float *__restrict__ xyz, const int *__restrict__ perm_in, int N,
                                              int M, int *__restrict__ idx_out) {
    static_assert(PPT <= 32 && NW <= 16, "bucket state lives in lanes 0..31; one record per wave");
    typedef float fvec __attribute__((ext_vector_type(PPT)));
    typedef unsigned uvec __attribute__((ext_vector_type(PPT)));
    constexpr int PSTEPS = PPT > 16 ? 5 : 4;
    // cross-wave arg-max: every wave folds its 64-bit key (wave id in the 4 low bits) into s_gkey
    // with one LDS atomic max; after the barrier one broadcast read gives the winner and its wave,
    // whose coordinates are taken from that wave's record.  Three key slots rotate (the slot of step
    // i+1 is cleared during step i), records are double-buffered.
    __shared__ u64 s_gkey[3];
    __shared__ __attribute__((aligned(16))) float s_wxyz[2][16][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    const int *perm = perm_in + (size_t)blockIdx.x * N;
    int *out = idx_out + (size_t)blockIdx.x * M;

    fvec px, py, pz, md;
    uvec ni;                                  // ~original index (larger = lower index, for the tie rule)
    // bucket state, valid in lane k for bucket k; other lanes: never active, never winning
    float blo0 = 0.f, blo1 = 0.f, blo2 = 0.f, bhi0 = 0.f, bhi1 = 0.f, bhi2 = 0.f;
    float bx = 0.f, by = 0.f, bz = 0.f;
    unsigned bmax = 0u, bidx = 0u;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pos = (k * NW + wave) * 64 + lane;
        const bool real = pos < N;
        float x = 0.f, y = 0.f, z = 0.f;
        unsigned n = 0x80000000u;             // padding: distance 0 and the largest index never beat a real point
        if (real) {
            const int j = perm[pos];
            x = p[j * 3 + 0];
            y = p[j * 3 + 1];
            z = p[j * 3 + 2];
            n = ~(unsigned)j;
        }
        px[k] = x; py[k] = y; pz[k] = z; ni[k] = n;
        md[k] = real ? __builtin_inff() : 0.f;
        float lo[3] = {real ? x : 3.0e38f, real ? y : 3.0e38f, real ? z : 3.0e38f};
        float hi[3] = {real ? x : -3.0e38f, real ? y : -3.0e38f, real ? z : -3.0e38f};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off >= 1; off >>= 1) {
                const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = b > hi[d] ? b : hi[d];
            }
        const bool any_real = __ballot(real) != 0ull;
        if (lane == k) {
            blo0 = lo[0]; blo1 = lo[1]; blo2 = lo[2];
            bhi0 = hi[0]; bhi1 = hi[1]; bhi2 = hi[2];
            bmax = any_real ? 0x7f800000u : 0u;   // +inf: the first step updates every real bucket
            bidx = 0x80000000u;
        }
    }

    float cx = p[0], cy = p[1], cz = p[2];
    unsigned wk_hi = 0u, wk_lo = 0u;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    if (tid == 0) out[0] = 0;
    if constexpr (NW > 1) {
        if (tid < 3) s_gkey[tid] = 0ull;
        __syncthreads();
    }
    int b3 = 1;
#ifdef SAD_FPS_STAMPS
    unsigned long long acc_0 = 0, acc_1 = 0, acc_2 = 0, acc_3 = 0, acc_4 = 0, acc_5 = 0, acc_6 = 0, acc_7 = 0;
#endif
    for (int i = 1; i < M; ++i) {
#ifdef SAD_FPS_STAMPS
        const bool sample = blockIdx.x == 0 && i >= M / 2;
#endif
        FPS_T(ta);
        // all PPT skip tests at once: lane k tests bucket k (a never-active lane has bmax = 0)
        const float dq = sad::d2f(__builtin_amdgcn_fmed3f(cx, blo0, bhi0), __builtin_amdgcn_fmed3f(cy, blo1, bhi1),
                                  __builtin_amdgcn_fmed3f(cz, blo2, bhi2), cx, cy, cz);
        unsigned act = __builtin_amdgcn_readfirstlane((unsigned)__ballot(dq < __builtin_bit_cast(float, bmax)));
        FPS_T(tb);
        FPS_ACC(0, tb - ta);
        FPS_ACC(5, act ? 1 : 0);
        FPS_ACC(6, __builtin_popcount(act));
        FPS_ACC(7, 1);
        if (act) {
            // a disturbed wave's chain is the step's critical path: it issues ahead of the three idle siblings on its SIMD
            // (2.87 -> 2.81 ms per 4 096 samples; keeping the priority through the publish: 2.83)
            __builtin_amdgcn_s_setprio(3);
            do {
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctz(act));
                act &= act - 1;
                const float x = px[k], y = py[k], z = pz[k];
                const unsigned n = ni[k];
                const float d = sad::d2f(x, y, z, cx, cy, cz);
                const float m = __builtin_fminf(md[k], d);
                md[k] = m;
                const unsigned mb = __builtin_bit_cast(unsigned, m);     // m >= +0: integer order = float order
                const unsigned hi = wave_max_u32_bcast(mb);
                const unsigned long long tie = __ballot(mb == hi);
                int l = __builtin_ctzll(tie);
                if (tie & (tie - 1)) {                                    // several lanes attain the max: lowest index wins
                    const unsigned lo = wave_max_u32_bcast(mb == hi ? n : 0u);
                    l = __builtin_ctzll(__ballot(mb == hi && n == lo));
                }
                bmax = wrl_dyn(bmax, hi, k);
                bidx = wrl_dyn(bidx, __builtin_amdgcn_readlane(n, l), k);
                bx = wrl_dyn_f(bx, rdl_f(x, l), k);
                by = wrl_dyn_f(by, rdl_f(y, l), k);
                bz = wrl_dyn_f(bz, rdl_f(z, l), k);
            } while (act);
            // wave best = arg-max over the bucket keys in lanes 0..PPT-1
            wk_hi = __builtin_amdgcn_readlane(half_max_u32<PSTEPS>(bmax), 0);
            const unsigned long long wt = __ballot(bmax == wk_hi);
            int kb = __builtin_ctzll(wt);
            if (wt & (wt - 1)) {                                          // equal maxima in several buckets
                wk_lo = __builtin_amdgcn_readlane(half_max_u32<PSTEPS>(bmax == wk_hi ? bidx : 0u), 0);
                kb = __builtin_ctzll(__ballot(bmax == wk_hi && bidx == wk_lo));
            } else {
                wk_lo = __builtin_amdgcn_readlane(bidx, kb);
            }
            wx = rdl_f(bx, kb);
            wy = rdl_f(by, kb);
            wz = rdl_f(bz, kb);
        }
        __builtin_amdgcn_s_setprio(0);
        FPS_T(tc);
        FPS_ACC(1, tc - tb);
        unsigned glo;
        if constexpr (NW == 1) {
            glo = wk_lo;
            cx = wx; cy = wy; cz = wz;
        } else {
            const int buf = i & 1;
            const int b3n = b3 == 2 ? 0 : b3 + 1;
            if (lane == 0) {
                const unsigned lo = ((wk_lo & 0x0fffffffu) << 4) | (unsigned)wave;   // index order kept, N < 2^28
                atomicMax(&s_gkey[b3], ((u64)wk_hi << 32) | lo);
                float4 r;
                r.x = wx; r.y = wy; r.z = wz; r.w = 0.f;
                *reinterpret_cast<float4 *>(&s_wxyz[buf][wave][0]) = r;
                if (wave == 0) s_gkey[b3n] = 0ull;
            }
#ifdef SAD_FPS_STAMPS
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the publish has left
#endif
            FPS_T(td);
            FPS_ACC(2, td - tc);
            __syncthreads();
            FPS_T(te);
            FPS_ACC(3, te - td);
            const u64 gk = s_gkey[b3];
            const float4 rec = *reinterpret_cast<const float4 *>(&s_wxyz[buf][lane & (NW - 1)][0]);
            const unsigned g = __builtin_amdgcn_readfirstlane((unsigned)gk);
            const int slot = g & 15;
            glo = ~(0x0fffffffu - (g >> 4));
            cx = rdl_f(rec.x, slot);
            cy = rdl_f(rec.y, slot);
            cz = rdl_f(rec.z, slot);
            b3 = b3n;
#ifdef SAD_FPS_STAMPS
            asm volatile("" :: "v"(cx), "v"(cy), "v"(cz));
#endif
            FPS_T(tf);
            FPS_ACC(4, tf - te);
        }
        if (tid == 0) out[i] = (int)(~glo);
    }
#ifdef SAD_FPS_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        unsigned long long *d = g_fpst + wave * 8;
        d[0] = acc_0; d[1] = acc_1; d[2] = acc_2; d[3] = acc_3; d[4] = acc_4; d[5] = acc_5; d[6] = acc_6; d[7] = acc_7;
    }
#endif
}

template <int NW, int PPT>
__global__ __launch_bounds__(NW * 64) void fps_cell_kernel(const float *__restrict__ xyz, const int *__restrict__ perm_in,
                                                           int N, int M, int *__restrict__ idx_out) {
    fps_cell_body<NW, PPT>(xyz, perm_in, N, M, idx_out);
}

// ---- cell-bucket kernel, second form ("lane-direct publish", round 4) ----------------------------------------------
// Same buckets, same skip test, same proof — what changed is what a serial step makes every wave ISSUE.  Measured on
// the first form: a step is not only the disturbed wave's dependent chain, the four waves of a SIMD also share its
// vector issue port, and every one of them ran ~31 vector instructions per step just to publish its (unchanged) best
// and read the winner (mbcnt lane election of the LDS atomic, 8 v_mov of wave-uniform values into LDS data registers,
// 4 v_readlane of the wave's best point), the disturbed wave ~85 more (5 v_readlane + 5 v_writelane per bucket update to
// cache the bucket's best point in the bucket's state lane, 4 more per wave best).  Here
//  * a bucket keeps only (box, max min-distance, LANE of the point that attains it): two v_writelane per update;
//  * the wave's best point is never moved: the lane that owns it (lane `wl`, register slot `kb`) writes the wave's
//    record and key itself — ds_write_b128 / ds_max_u64 under a one-lane exec mask, no readlane, no v_mov;
//  * a wave republishes its record only when (kb, wl, key) changed (and once more for the other record buffer);
//    its key lives on in lane wl's registers: an undisturbed wave's step is the skip test, one LDS atomic, the
//    barrier and the read of the winner;
//  * the wave best is recomputed only when the best bucket itself was updated or an updated bucket reaches its key;
//  * the 64-lane max uses single-instruction row_bcast DPP steps (hipcc expands the builtin into mov + mov_dpp + max).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)p; }   // low half of a flat LDS address
__device__ __forceinline__ unsigned wave_max_u32_b(unsigned v) {   // wave-uniform result (SGPR)
    v = row_max_u32<4>(v);
    // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3; dst == src1, so the rows left out keep their value
    // (no wait states are inserted inside asm: two after the VALU write of v, two between the DPP steps)
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

#ifdef SAD_FPS_STAMPS2
// measurement build of the second form: s_memtime ticks per phase, summed by every wave of workgroup 0 over the steps in
// which IT held the sampled point (its chain is the step's critical path), second half of the run:
// [0] barrier -> centre known, [1] skip test, [2] bucket updates, [3] wave best, [4] record, [5] key atomic issued -> barrier passed,
// [6] such steps, [7] buckets updated in them
__device__ unsigned long long g_fpst2[16 * 8];
extern "C" __attribute__((visibility("default"))) int sad_debug_read_fps_stamps2(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fpst2), sizeof(unsigned long long) * 16 * 8);
}
#define FPS2_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FPS2_ACC(slot, expr) do { if (s.crit) s.acc[slot] += (expr); } while (0)
#else
#define FPS2_T(v)
#define FPS2_ACC(slot, expr)
#endif

// State of one wave (everything is a register; the struct exists so that the six unrolled step bodies share one text).
template <int NW, int PPT>
struct Cell2 {
    typedef float fvec __attribute__((ext_vector_type(PPT)));
    typedef unsigned uvec __attribute__((ext_vector_type(PPT)));
    fvec px, py, pz, md;
    uvec ni;                                  // ~original index (larger = lower index, for the tie rule)
    float blo0, blo1, blo2, bhi0, bhi1, bhi2; // lane k: bounding box of bucket k
    unsigned bmax, blane;                     // lane k: bucket k's max min-distance (bits) and the lane that attains it
    float cx, cy, cz;                         // the centre sampled last (wave-uniform)
    int kb, wl;                               // the wave's best point: bucket (register slot) kb, lane wl
    unsigned wk_hi;                           // its min-distance bits
    unsigned long long wlmask;                // 1 << wl: the exec mask of the publishing lane
    unsigned klo, khi;                        // the wave's key, valid in lane wl
    unsigned pending;                         // the other record buffer still holds the previous best
    unsigned a_key, a_rec_mine, a_rec0;       // LDS byte addresses: key slots, own record, record of wave 0
    int wave;
#ifdef SAD_FPS_STAMPS2
    unsigned long long acc[8], t_bar;
    bool crit, sample;
#endif
};

// The rare part of a step: bucket updates, the wave's new best, its record.  BUF = record buffer of this step.
// Everything here is on the step's critical path when this wave holds the sampled point (its bucket is always disturbed),
// so the text is kept short: no divergent control flow (lane-masked work runs under an exec mask set inside asm, which
// keeps every flag in a scalar register), one register-indexed region per group of reads, popcounts for the tie checks.
template <int NW, int PPT, int BUF>
__device__ __forceinline__ void cell2_slow(Cell2<NW, PPT> &s, unsigned act, int lane) {
    constexpr int PSTEPS = PPT > 16 ? 5 : 4;
    constexpr unsigned long long PMASK = PPT >= 32 ? 0xffffffffull : ((1ull << PPT) - 1ull);
    unsigned publish = s.pending;
    s.pending = 0u;
    FPS2_T(ts0);
    FPS2_ACC(7, __builtin_popcount(act));
    if (act) {
        __builtin_amdgcn_s_setprio(3);        // the disturbed wave's chain ahead of its idle siblings on the SIMD
        // the wave best must be recomputed when its bucket is updated, or when an updated bucket ends up with the same
        // key high word (min-distances only fall, so none can end up above it)
        const unsigned kb_hit = (act >> s.kb) & 1u;
        unsigned top = 0u;
        do {
            const int k = __builtin_amdgcn_readfirstlane(__builtin_ctz(act));
            act &= act - 1;
            float x = s.px[k], y = s.py[k], z = s.pz[k], om = s.md[k];
            // (pinned: without it hipcc folds the register-indexed read of md[k] into v_min_u32's operand AFTER
            // s_set_gpr_idx_off, i.e. reads md[0] — seen in the disassembly of this kernel, ROCm 7.2)
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(om));
            const float d = sad::d2f(x, y, z, s.cx, s.cy, s.cz);
            // min on the bit patterns: both are >= +0 (no NaN: SPEC §2), integer order = float order, and
            // v_min_u32 needs no canonicalising v_max first
            const unsigned db = __builtin_bit_cast(unsigned, d), ob = __builtin_bit_cast(unsigned, om);
            const unsigned mb = db < ob ? db : ob;
            s.md[k] = __builtin_bit_cast(float, mb);
#if defined(SAD_FPS_DUP) && SAD_FPS_DUP == 2      // measurement: the 64-lane reduction a second time
            { unsigned t = mb ^ 1u; asm volatile("" : "+v"(t)); unsigned r = wave_max_u32_b(t); asm volatile("" :: "s"(r)); }
#endif
#ifdef SAD_FPS_KEEP      // (measurement build; NOT the default: see the comment)
            {   // Round 5: two thirds of the bucket updates leave the bucket's maximum where it was (the sample lowers a few
                // points near it, not the farthest one).  Min-distances only fall, so if the recorded lane still holds the
                // recorded value, maximum and lane are unchanged (a lane that ties now tied before, and the recorded lane was
                // the lowest index among those): three lane reads instead of the 64-lane reduction, ballot and writelanes.
                // Exact (all FPS tests) and SLOWER here: 2.195 against 2.094 ms per 32 scenes — the wave that holds the sampled
                // point never takes the shortcut for that point's bucket and pays the three dependent lane reads on the step's
                // critical chain.  The record-streaming kernel below keeps its version (11.96 against 12.08 ms: its updates
                // carry four more lane reads and five writelanes, and wait for L2 anyway).
                const unsigned obm = __builtin_amdgcn_readlane(s.bmax, k);
                const int obl = (int)__builtin_amdgcn_readlane(s.blane, k);
                if (__builtin_amdgcn_readlane(mb, obl) == obm) {
                    top = obm > top ? obm : top;
                    continue;
                }
            }
#endif
            const unsigned hi = wave_max_u32_b(mb);
            const unsigned long long tie = __ballot(mb == hi);
            int l = __builtin_ctzll(tie);
            if (__builtin_popcountll(tie) > 1) {                      // several lanes attain the max: lowest index wins
                const unsigned n = s.ni[k];
                const unsigned lo = wave_max_u32_b(mb == hi ? n : 0u);
                l = __builtin_ctzll(__ballot(mb == hi && n == lo));
            }
            // bucket k's state lane: its new maximum and the lane that holds it (both writes through one m0)
            s.bmax = sad_writelane(hi, k, s.bmax);
            s.blane = sad_writelane((unsigned)l, k, s.blane);
            top = hi > top ? hi : top;
        } while (act);
        FPS2_T(ts1);
        FPS2_ACC(2, ts1 - ts0);
        if (kb_hit | (top >= s.wk_hi ? 1u : 0u)) {
            const unsigned nh = __builtin_amdgcn_readlane(half_max_u32<PSTEPS>(s.bmax), 0);
            const unsigned long long wt = __ballot(s.bmax == nh) & PMASK;
            int nk = __builtin_ctzll(wt);
            int nl = (int)__builtin_amdgcn_readlane(s.blane, nk);
            if (__builtin_popcountll(wt) > 1) {                       // equal maxima in several buckets: lowest index wins
                unsigned best_n = 0u;
                unsigned long long rest = wt;
                do {
                    const int k2 = __builtin_amdgcn_readfirstlane(__builtin_ctzll(rest));
                    rest &= rest - 1;
                    const int l2 = (int)__builtin_amdgcn_readlane(s.blane, k2);
                    const unsigned n2 = __builtin_amdgcn_readlane(s.ni[k2], l2);
                    if (n2 >= best_n) { best_n = n2; nk = k2; nl = l2; }   // (indices are unique: >= only matters for the first)
                } while (rest);
            }
            s.kb = nk; s.wl = nl; s.wk_hi = nh;
            s.wlmask = 1ull << nl;
            publish = 3u;                     // new key + record (even if it came out the same: rare, and cheaper than comparing)
        }
        __builtin_amdgcn_s_setprio(0);
        FPS2_T(ts2);
        FPS2_ACC(3, ts2 - ts1);
    }
    FPS2_T(ts3);
    if (publish) {
        // the owner of the wave's best point (lane wl, register slot kb) publishes it from its own registers: every lane reads
        // slot kb, lane wl alone writes (s_and_saveexec: the statement saves the exec mask it finds and restores it)
        f4v r;
        unsigned long long sv;
        r.x = s.px[s.kb]; r.y = s.py[s.kb]; r.z = s.pz[s.kb]; r.w = 0.f;
        if (publish & 2u) {
            const unsigned n = s.ni[s.kb];
            asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                         "v_lshl_or_b32 %[klo], %[n], 4, %[w]\n\t"    // index order kept (the top four bits fall off: N < 2^28)
                         "v_mov_b32 %[khi], %[hi]\n\t"
                         "ds_write_b128 %[a], %[r] offset:%[o]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [klo] "+v"(s.klo), [khi] "+v"(s.khi), [sv] "=&s"(sv)
                         : [n] "v"(n), [w] "s"(s.wave), [m] "s"(s.wlmask), [hi] "s"(s.wk_hi),
                           [a] "v"(s.a_rec_mine), [r] "v"(r), [o] "n"(BUF * 256) : "memory", "scc");
            s.pending = 1u;                   // the other buffer gets the record in the next step
        } else {
            asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b128 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(s.wlmask), "v"(s.a_rec_mine), "v"(r), "n"(BUF * 256) : "memory", "scc");
        }
    }
    FPS2_T(ts4);
    FPS2_ACC(4, ts4 - ts3);
}

// One sampling step.  BUF / B3: record buffer and key slot of this step (compile-time: the loop is unrolled six times, so
// every LDS address is a base register plus an immediate); W0: this is wave 0, which also clears the next key slot and
// stores the sampled index.  What an undisturbed wave issues: the skip test (10 vector instructions), one LDS atomic
// under a one-lane exec mask, the barrier, two LDS reads, four lane reads — and about as many scalar instructions.
template <int NW, int PPT, int BUF, int B3, bool W0>
__device__ __forceinline__ void cell2_step(Cell2<NW, PPT> &s, int lane, int *__restrict__ out_i) {
    constexpr int B3N = B3 == 2 ? 0 : B3 + 1;
    FPS2_T(ta);
    FPS2_ACC(0, ta - s.t_bar);
    FPS2_ACC(6, 1);
    const float dq = sad::d2f(__builtin_amdgcn_fmed3f(s.cx, s.blo0, s.bhi0), __builtin_amdgcn_fmed3f(s.cy, s.blo1, s.bhi1),
                              __builtin_amdgcn_fmed3f(s.cz, s.blo2, s.bhi2), s.cx, s.cy, s.cz);
    unsigned act = __builtin_amdgcn_readfirstlane((unsigned)__ballot(dq < __builtin_bit_cast(float, s.bmax)));
#if defined(SAD_FPS_ABL) && SAD_FPS_ABL == 1
    act = 0;                                  // ablation: no bucket updates (timing only, wrong results)
#endif
    FPS2_T(tb);
    FPS2_ACC(1, tb - ta);
    if (act | s.pending) cell2_slow<NW, PPT, BUF>(s, act, lane);
    FPS2_T(tc);
    {
        const u64 key = ((u64)s.khi << 32) | s.klo;
        // (the exec mask found on entry is saved and restored inside the statement; scc is declared clobbered)
        unsigned long long sv;
        asm volatile("s_and_saveexec_b64 %0, %1\n\tds_max_u64 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv) : "s"(s.wlmask), "v"(s.a_key), "v"(key), "n"(B3 * 8) : "memory", "scc");
        if (W0) {
            const u64 zero = 0ull;
            asm volatile("s_and_saveexec_b64 %0, 1\n\tds_write_b64 %1, %2 offset:%3\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(s.a_key), "v"(zero), "n"(B3N * 8) : "memory", "scc");
        }
        // After the barrier: the winning key (one broadcast read), then the winner's record alone (a second broadcast
        // read at an address computed from the key in vector registers: no scalar round trip).  Sixteen waves reading all
        // sixteen records at once (64 lanes x 16 bytes each) queued in the LDS pipeline for ~200 cycles; two dependent
        // single-address reads take less, and the centre arrives lane-uniform in vector registers: no v_readlane.
        unsigned gk;                          // low word of the winning key: (index order << 4) | wave
        f4v rec;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SAD_FPS_STAMPS2
        { FPS2_T(td); FPS2_ACC(5, td - tc); s.t_bar = td; }
#endif
        {
            unsigned ra;
            asm volatile("ds_read_b32 %0, %3 offset:%5\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_and_b32 %2, 15, %0\n\t"                  // the winner's wave (low four bits of the key's low word)
                         "v_lshl_add_u32 %2, %2, 4, %4\n\t"          // its record: 16 bytes per wave
                         "ds_read_b128 %1, %2 offset:%6\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(gk), "=&v"(rec), "=&v"(ra) : "v"(s.a_key), "v"(s.a_rec0), "n"(B3 * 8), "n"(BUF * 256) : "memory");
        }
        s.cx = rec.x; s.cy = rec.y; s.cz = rec.z;
#ifdef SAD_FPS_STAMPS2
        asm volatile("" :: "v"(s.cx), "v"(s.cy), "v"(s.cz));
        s.crit = ((__builtin_amdgcn_readfirstlane(gk) & 15) == (unsigned)s.wave) && s.sample;
#endif
        if (W0) {
            if (lane == 0) *out_i = (int)(0x0fffffffu - (gk >> 4));
        }
    }
}

template <int NW, int PPT, bool W0>
__device__ __forceinline__ void cell2_loop(Cell2<NW, PPT> &s, int lane, int M, int *__restrict__ out) {
    int i = 1;                                // step i: record buffer i & 1, key slot i % 3  (i = 1: 1, 1)
    for (; i + 6 <= M; i += 6) {
#ifdef SAD_FPS_STAMPS2
        s.sample = blockIdx.x == 0 && i >= M / 2;
#endif
        cell2_step<NW, PPT, 1, 1, W0>(s, lane, out + i);
        cell2_step<NW, PPT, 0, 2, W0>(s, lane, out + i + 1);
        cell2_step<NW, PPT, 1, 0, W0>(s, lane, out + i + 2);
        cell2_step<NW, PPT, 0, 1, W0>(s, lane, out + i + 3);
        cell2_step<NW, PPT, 1, 2, W0>(s, lane, out + i + 4);
        cell2_step<NW, PPT, 0, 0, W0>(s, lane, out + i + 5);
    }
    // the last M - i < 6 steps, same order
    if (i < M) cell2_step<NW, PPT, 1, 1, W0>(s, lane, out + i);
    if (i + 1 < M) cell2_step<NW, PPT, 0, 2, W0>(s, lane, out + i + 1);
    if (i + 2 < M) cell2_step<NW, PPT, 1, 0, W0>(s, lane, out + i + 2);
    if (i + 3 < M) cell2_step<NW, PPT, 0, 1, W0>(s, lane, out + i + 3);
    if (i + 4 < M) cell2_step<NW, PPT, 1, 2, W0>(s, lane, out + i + 4);
}

template <int NW, int PPT>
__global__ __launch_bounds__(NW * 64) void fps_cell2_kernel(const float *__restrict__ xyz, const int *__restrict__ perm_in,
                                                            int N, int M, int *__restrict__ idx_out) {
    static_assert(PPT <= 32 && NW <= 16 && NW >= 2, "bucket state lives in lanes 0..31; one record per wave");
    __shared__ u64 s_gkey[3];
    __shared__ __attribute__((aligned(16))) float s_wxyz[2][16][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    const int *perm = perm_in + (size_t)blockIdx.x * N;
    int *out = idx_out + (size_t)blockIdx.x * M;

    Cell2<NW, PPT> s;
    s.blo0 = s.blo1 = s.blo2 = s.bhi0 = s.bhi1 = s.bhi2 = 0.f;
    s.bmax = 0u; s.blane = 0u;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pos = (k * NW + wave) * 64 + lane;
        const bool real = pos < N;
        float x = 0.f, y = 0.f, z = 0.f;
        unsigned n = 0x80000000u;             // padding: distance 0 and the largest index never beat a real point
        if (real) {
            const int j = perm[pos];
            x = p[j * 3 + 0];
            y = p[j * 3 + 1];
            z = p[j * 3 + 2];
            n = ~(unsigned)j;
        }
        s.px[k] = x; s.py[k] = y; s.pz[k] = z; s.ni[k] = n;
        s.md[k] = real ? __builtin_inff() : 0.f;
        float lo[3] = {real ? x : 3.0e38f, real ? y : 3.0e38f, real ? z : 3.0e38f};
        float hi[3] = {real ? x : -3.0e38f, real ? y : -3.0e38f, real ? z : -3.0e38f};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off >= 1; off >>= 1) {
                const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = b > hi[d] ? b : hi[d];
            }
        const bool any_real = __ballot(real) != 0ull;
        if (lane == k) {
            s.blo0 = lo[0]; s.blo1 = lo[1]; s.blo2 = lo[2];
            s.bhi0 = hi[0]; s.bhi1 = hi[1]; s.bhi2 = hi[2];
            s.bmax = any_real ? 0x7f800000u : 0u;   // +inf: the first step updates every real bucket
        }
    }
    s.cx = p[0]; s.cy = p[1]; s.cz = p[2];
    if (tid == 0) out[0] = 0;
    if (tid < 3) s_gkey[tid] = 0ull;
    if (tid < 2 * 16 * 4) (&s_wxyz[0][0][0])[tid] = 0.f;
    __syncthreads();
    s.a_key = lds_off(&s_gkey[0]);
    s.a_rec_mine = lds_off(&s_wxyz[0][wave][0]);           // + 256 for the second buffer
    s.a_rec0 = lds_off(&s_wxyz[0][0][0]);
    s.wave = wave;
    s.kb = 0; s.wl = 0; s.wk_hi = 0u; s.wlmask = 1ull;
    s.klo = (unsigned)wave; s.khi = 0u;       // (a wave without a real point keeps key 0 | wave and never wins: wave 0 has one)
    s.pending = 1u;
#ifdef SAD_FPS_STAMPS2
    for (int q = 0; q < 8; ++q) s.acc[q] = 0;
    s.t_bar = 0; s.crit = false; s.sample = false;
#endif
    if (wave == 0) cell2_loop<NW, PPT, true>(s, lane, M, out);
    else cell2_loop<NW, PPT, false>(s, lane, M, out);
#ifdef SAD_FPS_STAMPS2
    if (blockIdx.x == 0 && lane == 0)
        for (int q = 0; q < 8; ++q) g_fpst2[wave * 8 + q] = s.acc[q];
#endif
}

// ---- cell-bucket kernel, third form ("several samples per round", round 5) ------------------------------------------
// The first two forms take ONE sample per barrier round: 4 096 dependent rounds of ~1 000 cycles for a 16 384-point scene,
// whatever the chain inside a round is trimmed to.  This form takes SEVERAL samples per round, exactly:
//   Let c1 > c2 > ... be the points in descending key order (key = (min-distance, index order): a strict total order).  Plain
//   FPS samples c1, lowers min-distances within reach of c1, and takes the new maximum.  Min-distances only fall, so if c2's own
//   min-distance is NOT lowered by c1 (d2(c2, c1) >= mind[c2]) and is positive, c2 is that new maximum: every other point was
//   below c2 before and cannot have risen, and c1 itself dropped to 0.  By induction c(j+1) is the sample after c1..cj whenever
//   none of c1..cj lowers it.  So a round may take the longest prefix c1..cP of the global order in which no member is within
//   reach of an earlier one — provided the prefix is KNOWN to be the global order.
//   Every wave publishes the two largest keys among its points (k1, k2: exact, see wave top-2 below).  In the merged list of
//   the 2 NW published keys, an unpublished point of wave w lies below k2(w); so the list's prefix is the global order up to
//   and including the first k2 it contains (beyond that, the wave that supplied it may hide a larger third point).
// A round: [apply the round's samples: lane-parallel box tests, bucket updates, the wave's new top-2, published by the lanes
// that own the two points] barrier [every wave ranks ITS two candidates among the 32 table entries and tests them against
// the higher-ranked ones: two ballots + one distance evaluation per candidate] barrier [every wave derives the same prefix
// length P from the 32 results and reads the P sample coordinates out of its table registers].  On KITTI-shaped scenes a
// round takes 5 samples on average (tools/probe/fps_rounds.py; 16 384 -> 4 096 in ~810 rounds instead of 4 095).
// Same buckets, same box test, same tie rules as the other forms; indices are bit-exact (tests/test_gpu_ops.py).
constexpr int MP_KMAX = 8;                    // samples per round (cap)
#ifdef SAD_FPS_STAMPS3
// measurement build: s_memtime ticks per phase, per wave of workgroup 0, summed over the rounds of the second half of the run:
// [0] box tests, [1] bucket updates, [2] top-2 + publish, [3] wait at the first barrier, [4] ranking, [5] wait at the second
// barrier, [6] prefix + samples, [7] rounds, [8] buckets updated, [9] rounds with a top-2 recompute, [10] samples
__device__ unsigned long long g_fpst3[16 * 12];
extern "C" __attribute__((visibility("default"))) int sad_debug_read_fps_stamps3(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fpst3), sizeof(unsigned long long) * 16 * 12);
}
#define FPS3_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define FPS3_ACC(slot, expr) do { if (sample3) acc3[slot] += (expr); } while (0)
#else
#define FPS3_T(v)
#define FPS3_ACC(slot, expr)
#endif

template <int NW, int PPT>
__global__ __launch_bounds__(NW * 64) void fps_cell3_kernel(const float *__restrict__ xyz, const int *__restrict__ perm_in,
                                                            int N, int M, int *__restrict__ idx_out) {
    static_assert(PPT <= 32 && NW <= 16 && NW >= 1, "bucket state lives in lanes 0..31; two table entries per wave");
    typedef float fvec __attribute__((ext_vector_type(PPT)));
    typedef unsigned uvec __attribute__((ext_vector_type(PPT)));
    constexpr int PSTEPS = PPT > 16 ? 5 : 4;
    constexpr unsigned long long PMASK = PPT >= 32 ? 0xffffffffull : ((1ull << PPT) - 1ull);
    // table: entry 2 w + s = wave w's best (s = 0) / second best (s = 1) point: key, coordinates; its rank | bad << 8 (written by
    // its wave); the round's samples in rank order (x, y, z, ~index), written by the waves whose candidates rank below MP_KMAX
    __shared__ __attribute__((aligned(16))) u64 s_key[32];
    __shared__ __attribute__((aligned(16))) float s_rec[32][4];
    __shared__ __attribute__((aligned(16))) unsigned s_res[32];
    __shared__ __attribute__((aligned(16))) float s_cen[MP_KMAX][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    const int *perm = perm_in + (size_t)blockIdx.x * N;
    int *out = idx_out + (size_t)blockIdx.x * M;

    fvec px, py, pz, md;
    uvec ni;                                  // ~original index (larger = lower index, for the tie rule)
    float blo0 = 0.f, blo1 = 0.f, blo2 = 0.f, bhi0 = 0.f, bhi1 = 0.f, bhi2 = 0.f;
    unsigned bmax = 0u, blane = 0u;           // lane k: bucket k's max min-distance (bits) and the lane that attains it
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pos = (k * NW + wave) * 64 + lane;
        const bool real = pos < N;
        float x = 0.f, y = 0.f, z = 0.f;
        unsigned n = 0x80000000u;             // padding: distance 0 and the largest index never beat a real point
        if (real) {
            const int j = perm[pos];
            x = p[j * 3 + 0];
            y = p[j * 3 + 1];
            z = p[j * 3 + 2];
            n = ~(unsigned)j;
        }
        px[k] = x; py[k] = y; pz[k] = z; ni[k] = n;
        md[k] = real ? __builtin_inff() : 0.f;
        float lo[3] = {real ? x : 3.0e38f, real ? y : 3.0e38f, real ? z : 3.0e38f};
        float hi[3] = {real ? x : -3.0e38f, real ? y : -3.0e38f, real ? z : -3.0e38f};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off >= 1; off >>= 1) {
                const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = b > hi[d] ? b : hi[d];
            }
        const bool any_real = __ballot(real) != 0ull;
        if (lane == k) {
            blo0 = lo[0]; blo1 = lo[1]; blo2 = lo[2];
            bhi0 = hi[0]; bhi1 = hi[1]; bhi2 = hi[2];
            bmax = any_real ? 0x7f800000u : 0u;   // +inf: the first round updates every real bucket
        }
    }
    if (tid < 32) {
        s_key[tid] = 0ull;
        s_res[tid] = 63u | (1u << 8);
        s_rec[tid][0] = s_rec[tid][1] = s_rec[tid][2] = s_rec[tid][3] = 0.f;
    }
    if (tid < MP_KMAX) { s_cen[tid][0] = p[0]; s_cen[tid][1] = p[1]; s_cen[tid][2] = p[2]; s_cen[tid][3] = __builtin_bit_cast(float, ~0u); }
    if (tid == 0) out[0] = 0;
    __syncthreads();

    // the wave's two best points (all wave-uniform): bucket (register slot), min-distance bits, ~index, coordinates
    int kb = 0, kb2 = 0;
    unsigned v1 = 0u, n1 = 0u, v2 = 0u, n2 = 0u;
    float q1x = 0.f, q1y = 0.f, q1z = 0.f, q2x = 0.f, q2y = 0.f, q2z = 0.f;
    bool fresh = true;                        // nothing published yet
    // the samples of the current round (wave-uniform: scalar registers, one per vector instruction is free on gfx9); slots >= P
    // repeat sample 0 (applying a sample twice changes nothing), so nothing below depends on P except the second group of four
    float cx[MP_KMAX], cy[MP_KMAX], cz[MP_KMAX];
    {
        const float p0x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p[0])));
        const float p0y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p[1])));
        const float p0z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p[2])));
#pragma unroll
        for (int i = 0; i < MP_KMAX; ++i) { cx[i] = p0x; cy[i] = p0y; cz[i] = p0z; }
    }
    int P = 1, t = 1;
#ifdef SAD_FPS_STAMPS3
    unsigned long long acc3[12];
    for (int q = 0; q < 12; ++q) acc3[q] = 0;
#endif
    while (t < M) {
#ifdef SAD_FPS_STAMPS3
        const bool sample3 = blockIdx.x == 0 && t >= M / 2;
#endif
        FPS3_T(t0);
        FPS3_ACC(7, 1);
        FPS3_ACC(10, P);
        // ---- apply the round's samples: box tests (lane k tests bucket k), then the disturbed buckets ---------------
        const float bmf = __builtin_bit_cast(float, bmax);
        bool any = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dq = sad::d2f(__builtin_amdgcn_fmed3f(cx[i], blo0, bhi0), __builtin_amdgcn_fmed3f(cy[i], blo1, bhi1),
                                      __builtin_amdgcn_fmed3f(cz[i], blo2, bhi2), cx[i], cy[i], cz[i]);
            any = any || (dq < bmf);
        }
        const bool more = P > 4;              // (wave-uniform)
        if (more) {
#pragma unroll
            for (int i = 4; i < MP_KMAX; ++i) {
                const float dq = sad::d2f(__builtin_amdgcn_fmed3f(cx[i], blo0, bhi0), __builtin_amdgcn_fmed3f(cy[i], blo1, bhi1),
                                          __builtin_amdgcn_fmed3f(cz[i], blo2, bhi2), cx[i], cy[i], cz[i]);
                any = any || (dq < bmf);
            }
        }
        unsigned rest = __builtin_amdgcn_readfirstlane((unsigned)__ballot(any));
        FPS3_T(t1);
        FPS3_ACC(0, t1 - t0);
        if (rest) {
            const bool hit = fresh || ((((rest >> kb) | (rest >> kb2)) & 1u) != 0u);
            FPS3_ACC(8, __builtin_popcount(rest));
            do {
                // two disturbed buckets per pass (the second repeats the first when only one is left: the update is idempotent),
                // so that the two dependent chains — distances, minimum, 64-lane maximum, lane of the maximum — interleave
                const int k0 = __builtin_amdgcn_readfirstlane(__builtin_ctz(rest));
                rest &= rest - 1;
                const int k1 = rest ? __builtin_amdgcn_readfirstlane(__builtin_ctz(rest)) : k0;
                rest &= rest - 1;             // (0 & anything = 0)
                float x0 = px[k0], y0 = py[k0], z0 = pz[k0], o0 = md[k0];
                asm volatile("" : "+v"(x0), "+v"(y0), "+v"(z0), "+v"(o0));      // (register-indexed reads kept apart: see fps_cell2)
                float x1 = px[k1], y1 = py[k1], z1 = pz[k1], o1 = md[k1];
                asm volatile("" : "+v"(x1), "+v"(y1), "+v"(z1), "+v"(o1));
                unsigned m0 = __builtin_bit_cast(unsigned, o0), m1 = __builtin_bit_cast(unsigned, o1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned d0 = __builtin_bit_cast(unsigned, sad::d2f(x0, y0, z0, cx[i], cy[i], cz[i]));
                    const unsigned d1 = __builtin_bit_cast(unsigned, sad::d2f(x1, y1, z1, cx[i], cy[i], cz[i]));
                    m0 = d0 < m0 ? d0 : m0;       // (bit patterns of values >= +0: integer order = float order)
                    m1 = d1 < m1 ? d1 : m1;
                }
                if (more) {
#pragma unroll
                    for (int i = 4; i < MP_KMAX; ++i) {
                        const unsigned d0 = __builtin_bit_cast(unsigned, sad::d2f(x0, y0, z0, cx[i], cy[i], cz[i]));
                        const unsigned d1 = __builtin_bit_cast(unsigned, sad::d2f(x1, y1, z1, cx[i], cy[i], cz[i]));
                        m0 = d0 < m0 ? d0 : m0;
                        m1 = d1 < m1 ? d1 : m1;
                    }
                }
                md[k0] = __builtin_bit_cast(float, m0);
                md[k1] = __builtin_bit_cast(float, m1);       // (k1 == k0: the same value again)
                const unsigned h0 = wave_max_u32_b(m0), h1 = wave_max_u32_b(m1);
                const unsigned long long tie0 = __ballot(m0 == h0), tie1 = __ballot(m1 == h1);
                int l0 = __builtin_ctzll(tie0), l1 = __builtin_ctzll(tie1);
                if (__builtin_popcountll(tie0) > 1 || __builtin_popcountll(tie1) > 1) {     // equal maxima: lowest index wins
                    unsigned nk0 = ni[k0], nk1 = ni[k1];
                    asm volatile("" : "+v"(nk0), "+v"(nk1));
                    const unsigned lo0 = wave_max_u32_b(m0 == h0 ? nk0 : 0u), lo1 = wave_max_u32_b(m1 == h1 ? nk1 : 0u);
                    l0 = __builtin_ctzll(__ballot(m0 == h0 && nk0 == lo0));
                    l1 = __builtin_ctzll(__ballot(m1 == h1 && nk1 == lo1));
                }
                bmax = sad_writelane(h0, k0, bmax);
                blane = sad_writelane((unsigned)l0, k0, blane);
                bmax = sad_writelane(h1, k1, bmax);
                blane = sad_writelane((unsigned)l1, k1, blane);
            } while (rest);
            FPS3_T(t2);
            FPS3_ACC(1, t2 - t1);
            if (hit) {
                FPS3_ACC(9, 1);
                // the wave's best point: best bucket (ties: lowest index), the lane recorded for it
                auto bucket_of = [&](unsigned top, int excl, int &bk, int &bl, unsigned &bn) {
                    unsigned long long wt = __ballot(bmax == top && lane != excl) & PMASK;
                    bk = __builtin_ctzll(wt);
                    bl = (int)__builtin_amdgcn_readlane(blane, bk);
                    unsigned nv = ni[bk];
                    asm volatile("" : "+v"(nv));
                    bn = __builtin_amdgcn_readlane(nv, bl);
                    wt &= wt - 1;
                    while (wt) {                             // equal maxima in several buckets: lowest index wins
                        const int k2 = __builtin_amdgcn_readfirstlane(__builtin_ctzll(wt));
                        wt &= wt - 1;
                        const int l2 = (int)__builtin_amdgcn_readlane(blane, k2);
                        unsigned nv2 = ni[k2];
                        asm volatile("" : "+v"(nv2));
                        const unsigned c2 = __builtin_amdgcn_readlane(nv2, l2);
                        if (c2 > bn) { bn = c2; bk = k2; bl = l2; }
                    }
                };
                const unsigned nh = __builtin_amdgcn_readlane(half_max_u32<PSTEPS>(bmax), 0);
                int wl = 0;
                bucket_of(nh, -1, kb, wl, n1);
                v1 = nh;
                // second best: the runner-up inside bucket kb, or the best of the other buckets (both reductions first, ties after)
                float omk = md[kb], xk = px[kb], yk = py[kb], zk = pz[kb];
                unsigned nik = ni[kb];
                asm volatile("" : "+v"(omk), "+v"(nik), "+v"(xk), "+v"(yk), "+v"(zk));
                const unsigned mbk = lane != wl ? __builtin_bit_cast(unsigned, omk) : 0u;
                const unsigned bm2 = lane != kb ? bmax : 0u;
                const unsigned sA = wave_max_u32_b(mbk);
                const unsigned vB = __builtin_amdgcn_readlane(half_max_u32<PSTEPS>(bm2), 0);
                q1x = rdl_f(xk, wl); q1y = rdl_f(yk, wl); q1z = rdl_f(zk, wl);
                const unsigned long long tieA = __ballot(lane != wl && mbk == sA);
                int lA = __builtin_ctzll(tieA);
                if (__builtin_popcountll(tieA) > 1) {
                    const unsigned loA = wave_max_u32_b((lane != wl && mbk == sA) ? nik : 0u);
                    lA = __builtin_ctzll(__ballot(lane != wl && mbk == sA && nik == loA));
                }
                const unsigned nA = __builtin_amdgcn_readlane(nik, lA);
                int kB = kb, lB = lA;
                unsigned nB = 0u;
                if (PPT > 1) bucket_of(vB, kb, kB, lB, nB);
                if (PPT > 1 && (vB > sA || (vB == sA && nB > nA))) {
                    kb2 = kB; v2 = vB; n2 = nB;
                    float xb = px[kB], yb = py[kB], zb = pz[kB];
                    asm volatile("" : "+v"(xb), "+v"(yb), "+v"(zb));
                    q2x = rdl_f(xb, lB); q2y = rdl_f(yb, lB); q2z = rdl_f(zb, lB);
                } else {
                    kb2 = kb; v2 = sA; n2 = nA;
                    q2x = rdl_f(xk, lA); q2y = rdl_f(yk, lA); q2z = rdl_f(zk, lA);
                }
                fresh = false;
                // publish both entries (wave-uniform values: one lane writes them)
                if (lane == 0) {
                    typedef unsigned u4v __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<u4v *>(&s_key[2 * wave]) = u4v{n1, v1, n2, v2};
                    *reinterpret_cast<f4v *>(&s_rec[2 * wave][0]) = f4v{q1x, q1y, q1z, 0.f};
                    *reinterpret_cast<f4v *>(&s_rec[2 * wave + 1][0]) = f4v{q2x, q2y, q2z, 0.f};
                }
                FPS3_T(t3);
                FPS3_ACC(2, t3 - t2);
            }
        }
        FPS3_T(t4);
        __syncthreads();
        FPS3_T(t5);
        FPS3_ACC(3, t5 - t4);
        // ---- every wave ranks its two candidates in the table and tests them against the higher-ranked entries ------
        unsigned tkhi = 0u, tklo = 0u;
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (lane < 2 * NW) {
            const u64 key = s_key[lane];
            const f4v r = *reinterpret_cast<const f4v *>(&s_rec[lane][0]);
            tkhi = (unsigned)(key >> 32); tklo = (unsigned)key;
            tx = r.x; ty = r.y; tz = r.z;
        }
        {
            // (no branches: the two candidates' chains interleave; everything but the two distance evaluations is scalar)
            const unsigned long long g1 = __ballot(tkhi > v1 || (tkhi == v1 && tklo > n1));
            const unsigned long long g2 = __ballot(tkhi > v2 || (tkhi == v2 && tklo > n2));
            const float d1 = sad::d2f(q1x, q1y, q1z, tx, ty, tz);      // the candidate as the point, entry j as the sample
            const float d2 = sad::d2f(q2x, q2y, q2z, tx, ty, tz);
            const unsigned long long w1 = __ballot(d1 < __builtin_bit_cast(float, v1)) & g1;
            const unsigned long long w2 = __ballot(d2 < __builtin_bit_cast(float, v2)) & g2;
            unsigned r1 = (unsigned)__builtin_popcountll(g1), r2 = (unsigned)__builtin_popcountll(g2);
            // bad: lowered by a higher-ranked entry; or at distance 0 (such a point is only ever sampled first); or nothing published
            unsigned b1 = (w1 != 0ull || (v1 == 0u && r1 > 0u)) ? 1u : 0u;
            unsigned b2 = (w2 != 0ull || (v2 == 0u && r2 > 0u)) ? 1u : 0u;
            if (fresh || (v1 == 0u && n1 == 0u)) { r1 = 63u; b1 = 1u; }
            if (fresh || (v2 == 0u && n2 == 0u)) { r2 = 63u; b2 = 1u; }
            if (lane == 0) {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u2v *>(&s_res[2 * wave]) = u2v{r1 | (b1 << 8), r2 | (b2 << 8)};
                if (r1 < (unsigned)MP_KMAX) *reinterpret_cast<f4v *>(&s_cen[r1][0]) = f4v{q1x, q1y, q1z, __builtin_bit_cast(float, n1)};
                if (r2 < (unsigned)MP_KMAX) *reinterpret_cast<f4v *>(&s_cen[r2][0]) = f4v{q2x, q2y, q2z, __builtin_bit_cast(float, n2)};
            }
        }
        FPS3_T(t6);
        FPS3_ACC(4, t6 - t5);
        __syncthreads();
        FPS3_T(t7);
        FPS3_ACC(5, t7 - t6);
        // ---- the prefix: every wave derives the same P; the samples come sorted from the table ---------------------------
        const unsigned res = lane < 2 * NW ? s_res[lane] : (63u | (1u << 8));
        const f4v cen = *reinterpret_cast<const f4v *>(&s_cen[lane & (MP_KMAX - 1)][0]);     // lane i (< MP_KMAX): sample i of the round
        const unsigned rank = res & 0xFFu;
        const unsigned limit = (res >> 8) ? rank : ((lane & 1) ? rank + 1u : 255u);   // bad: stop before it; a second-best: stop behind it
        int np = 255 - (int)wave_max_u32_b(255u - limit);
        np = np < MP_KMAX ? np : MP_KMAX;
        np = np < M - t ? np : M - t;
        P = np;                                   // >= 1: the entry of rank 0 is never bad
#pragma unroll
        for (int i = 0; i < MP_KMAX; ++i) {
            const int j = i < P ? i : 0;          // (wave-uniform; slots beyond P repeat sample 0)
            cx[i] = rdl_f(cen.x, j);
            cy[i] = rdl_f(cen.y, j);
            cz[i] = rdl_f(cen.z, j);
        }
        const float cenw = cen.w;                 // (hipcc: a bit_cast applied directly to a vector element reads element 0)
        if (wave == 0 && lane < P) out[t + lane] = (int)(~__builtin_bit_cast(unsigned, cenw));
        t += P;
        FPS3_T(t8);
        FPS3_ACC(6, t8 - t7);
    }
#ifdef SAD_FPS_STAMPS3
    if (blockIdx.x == 0 && lane == 0)
        for (int q = 0; q < 12; ++q) g_fpst3[wave * 12 + q] = acc3[q];
#endif
}

template <int NW, int PPT>
void launch_cell3(const float *xyz, const int *perm, int B, int N, int M, int *idx, hipStream_t st) {
    hipLaunchKernelGGL((fps_cell3_kernel<NW, PPT>), dim3(B), dim3(NW * 64), 0, st, xyz, perm, N, M, idx);
}

template <int NW, int PPT>
void launch_cell2(const float *xyz, const int *perm, int B, int N, int M, int *idx, hipStream_t st) {
    hipLaunchKernelGGL((fps_cell2_kernel<NW, PPT>), dim3(B), dim3(NW * 64), 0, st, xyz, perm, N, M, idx);
}

template <int NW, int PPT>
void launch_cell(const float *xyz, const int *perm, int B, int N, int M, int *idx, hipStream_t st) {
    hipLaunchKernelGGL((fps_cell_kernel<NW, PPT>), dim3(B), dim3(NW * 64), 0, st, xyz, perm, N, M, idx);
}

// ---- cell-bucket kernel for 16 384 < N <= 65 536 -------------------------------------------------
// Same algorithm, but a scene no longer fits the register file of one CU: only the running
// min-distances (64 per lane, two 32-wide register-indexed vectors) and the 64 bucket states per wave
// stay in registers; a bucket's coordinates and (~index) come from a Z-order-sorted float4 record
// array in the workspace (1 MB per scene, L2 / Infinity-Cache resident) — ONE coalesced 16-byte load
// per lane, issued only for the handful of buckets a step disturbs.
__global__ __launch_bounds__(256) void fps_records_kernel(const float *__restrict__ xyz, const int *__restrict__ perm_in,
                                                          int N, int NP, float4 *__restrict__ rec_out) {
    const float *p = xyz + (size_t)blockIdx.y * N * 3;
    const int *perm = perm_in + (size_t)blockIdx.y * N;
    float4 *rec = rec_out + (size_t)blockIdx.y * NP;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < NP; pos += gridDim.x * blockDim.x) {
        float4 r = {0.f, 0.f, 0.f, __builtin_bit_cast(float, 0x80000000u)};   // padding record
        if (pos < N) {
            const int j = perm[pos];
            r.x = p[j * 3 + 0];
            r.y = p[j * 3 + 1];
            r.z = p[j * 3 + 2];
            r.w = __builtin_bit_cast(float, ~(unsigned)j);
        }
        rec[pos] = r;
    }
}

template <int PPT>      // slots per wave: 64 (up to 65 536 points) or 16 (up to 16 384 points, ~60 VGPRs)
__global__ __launch_bounds__(1024) void fps_cellg_kernel(const float4 *__restrict__ rec_in, int N, int NP, int M,
                                                         int *__restrict__ idx_out) {
    constexpr int NW = 16;
    constexpr int VW = PPT < 32 ? PPT : 32;
    typedef float fvec32 __attribute__((ext_vector_type(VW)));
    __shared__ u64 s_gkey[3];
    __shared__ __attribute__((aligned(16))) float s_wxyz[2][16][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float4 *rec = rec_in + (size_t)blockIdx.x * NP;
    int *out = idx_out + (size_t)blockIdx.x * M;

    fvec32 md0, md1;       // min-distance of point `lane` of slot k: md0[k] (k < 32) / md1[k - 32]
    float blo0 = 0.f, blo1 = 0.f, blo2 = 0.f, bhi0 = 0.f, bhi1 = 0.f, bhi2 = 0.f;
    float bx = 0.f, by = 0.f, bz = 0.f;
    unsigned bmax = 0u, bidx = 0x80000000u;
#pragma unroll 1
    for (int k = 0; k < PPT; ++k) {
        const int pos = (k * NW + wave) * 64 + lane;       // < NP by construction
        const bool real = pos < N;
        const float4 r = rec[pos];
        if (k < 32) md0[k & (VW - 1)] = real ? __builtin_inff() : 0.f; else md1[k & (VW - 1)] = real ? __builtin_inff() : 0.f;
        float lo[3] = {real ? r.x : 3.0e38f, real ? r.y : 3.0e38f, real ? r.z : 3.0e38f};
        float hi[3] = {real ? r.x : -3.0e38f, real ? r.y : -3.0e38f, real ? r.z : -3.0e38f};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off >= 1; off >>= 1) {
                const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = b > hi[d] ? b : hi[d];
            }
        const bool any_real = __ballot(real) != 0ull;
        if (lane == k) {
            blo0 = lo[0]; blo1 = lo[1]; blo2 = lo[2];
            bhi0 = hi[0]; bhi1 = hi[1]; bhi2 = hi[2];
            bmax = any_real ? 0x7f800000u : 0u;
        }
    }
    float cx, cy, cz;
    {   // the first sample is original index 0: find its record (one lane of one wave owns it)
        if (tid < 3) s_gkey[tid] = 0ull;
        __syncthreads();
        for (int k = 0; k < PPT; ++k) {
            const float4 r = rec[(k * NW + wave) * 64 + lane];
            if (__builtin_bit_cast(unsigned, r.w) == 0xffffffffu) {
                s_wxyz[0][0][0] = r.x; s_wxyz[0][0][1] = r.y; s_wxyz[0][0][2] = r.z;
            }
        }
        __syncthreads();
        cx = s_wxyz[0][0][0]; cy = s_wxyz[0][0][1]; cz = s_wxyz[0][0][2];
        __syncthreads();
    }
    unsigned wk_hi = 0u, wk_lo = 0u;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    if (tid == 0) out[0] = 0;
    int b3 = 1;
    for (int i = 1; i < M; ++i) {
        const float dq = sad::d2f(__builtin_amdgcn_fmed3f(cx, blo0, bhi0), __builtin_amdgcn_fmed3f(cy, blo1, bhi1),
                                  __builtin_amdgcn_fmed3f(cz, blo2, bhi2), cx, cy, cz);
        unsigned long long act = __ballot(dq < __builtin_bit_cast(float, bmax));
        if (act) {
            __builtin_amdgcn_s_setprio(3);       // (as in fps_cell_body: the disturbed wave's chain ahead of its idle siblings)
            do {
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(act));
                act &= act - 1;
                const float4 r = rec[(k * NW + wave) * 64 + lane];
                const unsigned n = __builtin_bit_cast(unsigned, r.w);
                const float d = sad::d2f(r.x, r.y, r.z, cx, cy, cz);
                float m;
                if (PPT <= 32 || k < 32) { m = __builtin_fminf(md0[k & (VW - 1)], d); md0[k & (VW - 1)] = m; }
                else { m = __builtin_fminf(md1[k & (VW - 1)], d); md1[k & (VW - 1)] = m; }
                const unsigned mb = __builtin_bit_cast(unsigned, m);
                const unsigned hi = wave_max_u32_bcast(mb);
                const unsigned long long tie = __ballot(mb == hi);
                int l = __builtin_ctzll(tie);
                if (tie & (tie - 1)) {
                    const unsigned lo = wave_max_u32_bcast(mb == hi ? n : 0u);
                    l = __builtin_ctzll(__ballot(mb == hi && n == lo));
                }
                bmax = wrl_dyn(bmax, hi, k);
                bidx = wrl_dyn(bidx, __builtin_amdgcn_readlane(n, l), k);
                bx = wrl_dyn_f(bx, rdl_f(r.x, l), k);
                by = wrl_dyn_f(by, rdl_f(r.y, l), k);
                bz = wrl_dyn_f(bz, rdl_f(r.z, l), k);
            } while (act);
            wk_hi = wave_max_u32_bcast(bmax);
            const unsigned long long wt = __ballot(bmax == wk_hi);
            int kb = __builtin_ctzll(wt);
            if (wt & (wt - 1)) {
                wk_lo = wave_max_u32_bcast(bmax == wk_hi ? bidx : 0u);
                kb = __builtin_ctzll(__ballot(bmax == wk_hi && bidx == wk_lo));
            } else {
                wk_lo = __builtin_amdgcn_readlane(bidx, kb);
            }
            wx = rdl_f(bx, kb);
            wy = rdl_f(by, kb);
            wz = rdl_f(bz, kb);
            __builtin_amdgcn_s_setprio(0);
        }
        const int buf = i & 1;
        const int b3n = b3 == 2 ? 0 : b3 + 1;
        if (lane == 0) {
            const unsigned lo = ((wk_lo & 0x0fffffffu) << 4) | (unsigned)wave;
            atomicMax(&s_gkey[b3], ((u64)wk_hi << 32) | lo);
            float4 r;
            r.x = wx; r.y = wy; r.z = wz; r.w = 0.f;
            *reinterpret_cast<float4 *>(&s_wxyz[buf][wave][0]) = r;
            if (wave == 0) s_gkey[b3n] = 0ull;
        }
        __syncthreads();
        const u64 gk = s_gkey[b3];
        const float4 wr = *reinterpret_cast<const float4 *>(&s_wxyz[buf][lane & (NW - 1)][0]);
        const unsigned g = __builtin_amdgcn_readfirstlane((unsigned)gk);
        const int slot = g & 15;
        cx = rdl_f(wr.x, slot);
        cy = rdl_f(wr.y, slot);
        cz = rdl_f(wr.z, slot);
        b3 = b3n;
        if (tid == 0) out[i] = (int)(0x0fffffffu - (g >> 4));
    }
}

// ---- record-streaming cell kernel, second form (round 4): the step skeleton of fps_cell2_kernel ------------------------
// (unrolled six times so that every LDS address is a base plus an immediate; the key atomic and the record are written by
// the bucket's state lane under a one-lane exec mask — that lane holds the bucket's best point already, so nothing is
// moved; the winning key is read first and then the winner's record alone; an undisturbed wave does not touch its record).
// The state is plain locals and the step a macro: with the state in a struct handed to inlined functions (as in
// fps_cell2) the 64-slot instantiation stayed in scratch — SROA gave up on the struct with two 32-wide register-indexed
// vectors — and every flag loaded from there counts as divergent.
// one bucket update: slot KS of the wave (register KI of the vector MD).  The two halves of the 64 slots run in SEPARATE
// loops, each touching one 32-wide vector only: with "if (k < 32) md0[k] ... else md1[k - 32] ..." around the read and
// again around the write, hipcc copied a whole 32-register vector into a temporary and back on every update
// (32 v_mov_b64: the first form of this kernel paid that too)
#ifdef SAD_FPS_NOKEEP
#define SAD_FPS_KEEP_ON false
#else
#define SAD_FPS_KEEP_ON true
#endif
#define CELLG2_UPD(MD, KI, KS)                                                                                        \
    {                                                                                                                 \
        const float4 r = rec[(KS) * 1024];                                                                            \
        const unsigned n = __builtin_bit_cast(unsigned, r.w);                                                         \
        const float d = sad::d2f(r.x, r.y, r.z, cx, cy, cz);                                                          \
        float om = MD[KI];                                                                                            \
        asm volatile("" : "+v"(om)); /* keeps the register-indexed read a separate v_mov (see fps_cell2) */           \
        const unsigned db = __builtin_bit_cast(unsigned, d), ob = __builtin_bit_cast(unsigned, om);                   \
        const unsigned mb = db < ob ? db : ob;                                                                        \
        MD[KI] = __builtin_bit_cast(float, mb);                                                                       \
        /* round 5: the bucket's best point (recorded by its ~index) still holds the recorded value: nothing to redo */ \
        const unsigned obm = __builtin_amdgcn_readlane(bmax, KS);                                                     \
        const unsigned obn = __builtin_amdgcn_readlane(bidx, KS);                                                     \
        const unsigned long long own = __ballot(n == obn);                                                            \
        const bool keep = SAD_FPS_KEEP_ON && own != 0ull &&                                                           \
                          __builtin_amdgcn_readlane(mb, (int)__builtin_ctzll(own | (1ull << 63))) == obm;             \
        if (keep) {                                                                                                   \
            top = obm > top ? obm : top;                                                                              \
        } else {                                                                                                      \
        const unsigned hi = wave_max_u32_b(mb);                                                                       \
        const unsigned long long tie = __ballot(mb == hi);                                                            \
        int l = __builtin_ctzll(tie);                                                                                 \
        if (__builtin_popcountll(tie) > 1) {                                                                          \
            const unsigned lo = wave_max_u32_b(mb == hi ? n : 0u);                                                    \
            l = __builtin_ctzll(__ballot(mb == hi && n == lo));                                                       \
        }                                                                                                             \
        const unsigned bn = __builtin_amdgcn_readlane(n, l);                                                          \
        const unsigned ux = __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, r.x), l);                          \
        const unsigned uy = __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, r.y), l);                          \
        const unsigned uz = __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, r.z), l);                          \
        bmax = sad_writelane(hi, KS, bmax);                                                              \
        bidx = sad_writelane(bn, KS, bidx);                                                              \
        bx = __builtin_bit_cast(float, sad_writelane(ux, KS, __builtin_bit_cast(unsigned, bx)));         \
        by = __builtin_bit_cast(float, sad_writelane(uy, KS, __builtin_bit_cast(unsigned, by)));         \
        bz = __builtin_bit_cast(float, sad_writelane(uz, KS, __builtin_bit_cast(unsigned, bz)));         \
        top = hi > top ? hi : top;                                                                                    \
        }                                                                                                             \
    }

#define CELLG2_SLOW(BUF)                                                                                              \
    {                                                                                                                 \
        unsigned publish = pending;                                                                                   \
        pending = 0u;                                                                                                 \
        if (act) {                                                                                                    \
            __builtin_amdgcn_s_setprio(3);                                                                            \
            const unsigned kb_hit = (unsigned)(act >> kb) & 1u;                                                       \
            unsigned top = 0u;                                                                                        \
            unsigned alo = (unsigned)act, ahi = (unsigned)(act >> 32);                                                \
            while (alo) {                                                                                             \
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctz(alo));                                     \
                alo &= alo - 1;                                                                                       \
                CELLG2_UPD(md0, k, k)                                                                                 \
            }                                                                                                         \
            if (PPT > 32) {                                                                                           \
                while (ahi) {                                                                                         \
                    const int k = __builtin_amdgcn_readfirstlane(__builtin_ctz(ahi));                                 \
                    ahi &= ahi - 1;                                                                                   \
                    CELLG2_UPD(md1, k, k + 32)                                                                        \
                }                                                                                                     \
            }                                                                                                         \
            if (kb_hit | (top >= wk_hi ? 1u : 0u)) {                                                                  \
                const unsigned nh = wave_max_u32_b(bmax);                                                             \
                const unsigned long long wt = __ballot(bmax == nh);                                                   \
                int nk = __builtin_ctzll(wt);                                                                         \
                if (__builtin_popcountll(wt) > 1) { /* equal maxima in several buckets: lowest index wins */         \
                    const unsigned lo = wave_max_u32_b(bmax == nh ? bidx : 0u);                                       \
                    nk = __builtin_ctzll(__ballot(bmax == nh && bidx == lo));                                         \
                }                                                                                                     \
                kb = __builtin_amdgcn_readfirstlane(nk);                                                              \
                wk_hi = nh;                                                                                           \
                publish = 3u;                                                                                         \
            }                                                                                                         \
            __builtin_amdgcn_s_setprio(0);                                                                            \
        }                                                                                                             \
        if (publish) {                                                                                                \
            f4v pr;                                                                                                   \
            unsigned long long sv;      /* (the exec mask the statement finds: saved and restored inside it) */      \
            pr.x = bx; pr.y = by; pr.z = bz; pr.w = 0.f;                                                              \
            if (publish & 2u) {                                                                                       \
                /* lane kb alone: it holds the bucket's best point; index order kept in the key (N < 2^28) */         \
                asm volatile("s_lshl_b64 %[sv], 1, %[kb]\n\ts_and_saveexec_b64 %[sv], %[sv]\n\t"                       \
                             "v_lshl_or_b32 %[klo], %[n], 4, %[w]\n\tv_mov_b32 %[khi], %[hi]\n\t"                     \
                             "ds_write_b128 %[a], %[r] offset:%[o]\n\ts_mov_b64 exec, %[sv]"                          \
                             : [klo] "+v"(klo), [khi] "+v"(khi), [sv] "=&s"(sv)                                        \
                             : [n] "v"(bidx), [w] "s"(wave), [kb] "s"(kb), [hi] "v"(bmax),                            \
                               [a] "v"(a_rec_mine), [r] "v"(pr), [o] "n"((BUF) * 256) : "memory", "scc");             \
                pending = 1u;                                                                                         \
            } else {                                                                                                  \
                asm volatile("s_lshl_b64 %0, 1, %1\n\ts_and_saveexec_b64 %0, %0\n\tds_write_b128 %2, %3 offset:%4\n\t" \
                             "s_mov_b64 exec, %0"                                                                     \
                             : "=&s"(sv) : "s"(kb), "v"(a_rec_mine), "v"(pr), "n"((BUF) * 256) : "memory", "scc");    \
            }                                                                                                         \
        }                                                                                                             \
    }

#define CELLG2_STEP(BUF, B3, OUT_I)                                                                                   \
    {                                                                                                                 \
        const float dq = sad::d2f(__builtin_amdgcn_fmed3f(cx, blo0, bhi0), __builtin_amdgcn_fmed3f(cy, blo1, bhi1),   \
                                  __builtin_amdgcn_fmed3f(cz, blo2, bhi2), cx, cy, cz);                               \
        const unsigned long long act0 = __ballot(dq < __builtin_bit_cast(float, bmax));                               \
        unsigned long long act = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(act0 >> 32)) << 32) | \
                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)act0);        \
        if ((act != 0ull) | (pending != 0u)) CELLG2_SLOW(BUF)                                                         \
        const u64 key = ((u64)khi << 32) | klo;                                                                       \
        unsigned long long sv;                                                                                        \
        asm volatile("s_lshl_b64 %0, 1, %1\n\ts_and_saveexec_b64 %0, %0\n\tds_max_u64 %2, %3 offset:%4\n\t"            \
                     "s_mov_b64 exec, %0"                                                                             \
                     : "=&s"(sv) : "s"(kb), "v"(a_key), "v"(key), "n"((B3) * 8) : "memory", "scc");                   \
        if (W0) {                                                                                                     \
            const u64 zero = 0ull;                                                                                    \
            asm volatile("s_and_saveexec_b64 %0, 1\n\tds_write_b64 %1, %2 offset:%3\n\ts_mov_b64 exec, %0"            \
                         : "=&s"(sv) : "v"(a_key), "v"(zero), "n"(((B3) == 2 ? 0 : (B3) + 1) * 8) : "memory", "scc"); \
        }                                                                                                             \
        unsigned gk, ra;                                                                                              \
        f4v wrec;                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"                                                          \
                     "ds_read_b32 %0, %3 offset:%5\n\t"                                                               \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                       \
                     "v_and_b32 %2, 15, %0\n\t"                                                                       \
                     "v_lshl_add_u32 %2, %2, 4, %4\n\t"                                                               \
                     "ds_read_b128 %1, %2 offset:%6\n\t"                                                              \
                     "s_waitcnt lgkmcnt(0)"                                                                           \
                     : "=&v"(gk), "=&v"(wrec), "=&v"(ra) : "v"(a_key), "v"(a_rec0), "n"((B3) * 8), "n"((BUF) * 256)   \
                     : "memory");                                                                                     \
        cx = wrec.x; cy = wrec.y; cz = wrec.z;                                                                        \
        if (W0) {                                                                                                     \
            if (lane == 0) out[OUT_I] = (int)(0x0fffffffu - (gk >> 4));                                               \
        }                                                                                                             \
    }

template <int PPT, bool W0>      // W0: the body of wave 0 (it also clears the next key slot and stores the sampled index)
__device__ __forceinline__ void cellg2_body(const float4 *__restrict__ rec_in, int N, int NP, int M, int *__restrict__ idx_out,
                                            u64 *s_gkey, float (*s_wxyz)[16][4]) {
    constexpr int NW = 16;
    constexpr int VW = PPT < 32 ? PPT : 32;
    typedef float fvec32 __attribute__((ext_vector_type(VW)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int *out = idx_out + (size_t)blockIdx.x * M;
    const float4 *rec = rec_in + (size_t)blockIdx.x * NP + wave * 64 + lane;      // slot k of this wave: rec[k * 1024]

    fvec32 md0, md1;       // min-distance of point `lane` of slot k: md0[k] (k < 32) / md1[k - 32]
    float blo0 = 0.f, blo1 = 0.f, blo2 = 0.f, bhi0 = 0.f, bhi1 = 0.f, bhi2 = 0.f;
    float bx = 0.f, by = 0.f, bz = 0.f;
    unsigned bmax = 0u, bidx = 0x80000000u;
#pragma unroll 1
    for (int k = 0; k < PPT; ++k) {
        const bool real = (k * NW + wave) * 64 + lane < N;
        const float4 r = rec[k * 1024];
        if (k < 32) md0[k & (VW - 1)] = real ? __builtin_inff() : 0.f; else md1[k & (VW - 1)] = real ? __builtin_inff() : 0.f;
        float lo[3] = {real ? r.x : 3.0e38f, real ? r.y : 3.0e38f, real ? r.z : 3.0e38f};
        float hi[3] = {real ? r.x : -3.0e38f, real ? r.y : -3.0e38f, real ? r.z : -3.0e38f};
#pragma unroll
        for (int d = 0; d < 3; ++d)
            for (int off = 32; off >= 1; off >>= 1) {
                const float a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
                lo[d] = a < lo[d] ? a : lo[d];
                hi[d] = b > hi[d] ? b : hi[d];
            }
        const bool any_real = __ballot(real) != 0ull;
        if (lane == k) {
            blo0 = lo[0]; blo1 = lo[1]; blo2 = lo[2];
            bhi0 = hi[0]; bhi1 = hi[1]; bhi2 = hi[2];
            bmax = any_real ? 0x7f800000u : 0u;
        }
    }
    float cx, cy, cz;
    {   // the first sample is original index 0: find its record (one lane of one wave owns it)
        if (tid < 3) s_gkey[tid] = 0ull;
        if (tid < 2 * 16 * 4) (&s_wxyz[0][0][0])[tid] = 0.f;
        __syncthreads();
        for (int k = 0; k < PPT; ++k) {
            const float4 r = rec[k * 1024];
            if (__builtin_bit_cast(unsigned, r.w) == 0xffffffffu) {
                s_wxyz[0][0][0] = r.x; s_wxyz[0][0][1] = r.y; s_wxyz[0][0][2] = r.z;
            }
        }
        __syncthreads();
        cx = s_wxyz[0][0][0]; cy = s_wxyz[0][0][1]; cz = s_wxyz[0][0][2];
        __syncthreads();
        if (tid < 4) s_wxyz[0][0][tid] = 0.f;
        __syncthreads();
    }
    if (tid == 0) out[0] = 0;
    const unsigned a_key = lds_off(&s_gkey[0]);
    const unsigned a_rec_mine = lds_off(&s_wxyz[0][wave][0]);
    const unsigned a_rec0 = lds_off(&s_wxyz[0][0][0]);
    int kb = 0;
    unsigned wk_hi = 0u, klo = (unsigned)wave, khi = 0u, pending = 1u;
    int i = 1;                                // step i: record buffer i & 1, key slot i % 3
    for (; i + 6 <= M; i += 6) {
        CELLG2_STEP(1, 1, i)
        CELLG2_STEP(0, 2, i + 1)
        CELLG2_STEP(1, 0, i + 2)
        CELLG2_STEP(0, 1, i + 3)
        CELLG2_STEP(1, 2, i + 4)
        CELLG2_STEP(0, 0, i + 5)
    }
    if (i < M) CELLG2_STEP(1, 1, i)
    if (i + 1 < M) CELLG2_STEP(0, 2, i + 1)
    if (i + 2 < M) CELLG2_STEP(1, 0, i + 2)
    if (i + 3 < M) CELLG2_STEP(0, 1, i + 3)
    if (i + 4 < M) CELLG2_STEP(1, 2, i + 4)
}
#undef CELLG2_STEP
#undef CELLG2_SLOW
#undef CELLG2_UPD

template <int PPT>      // slots per wave: 64 (up to 65 536 points) or 16 (up to 16 384 points)
__global__ __launch_bounds__(1024) void fps_cellg2_kernel(const float4 *__restrict__ rec_in, int N, int NP, int M,
                                                          int *__restrict__ idx_out) {
    __shared__ u64 s_gkey[3];
    __shared__ __attribute__((aligned(16))) float s_wxyz[2][16][4];
    if (threadIdx.x < 64) cellg2_body<PPT, true>(rec_in, N, NP, M, idx_out, s_gkey, s_wxyz);
    else cellg2_body<PPT, false>(rec_in, N, NP, M, idx_out, s_gkey, s_wxyz);
}

template <int THREADS, int PPT>
void launch_bucket(const float *xyz, const int *perm, int B, int N, int M, int *idx, hipStream_t st) {
    hipLaunchKernelGGL((fps_bucket_kernel<THREADS, PPT>), dim3(B), dim3(THREADS), 0, st, xyz, perm, N, M, idx);
}

}  // namespace

namespace sad {

// 16 384 < N <= 65 536: workspace = perm[B*N] ints, then B * 65 536 float4 records (16-byte aligned).
int launch_fps_cellg(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&fps_sort_kernel), 80 * 1024);
    const int NP = N <= 16384 ? 16384 : 65536;
    int *perm = (int *)workspace;
    const size_t off = (((size_t)B * N * sizeof(int)) + 15) & ~(size_t)15;
    float4 *rec = (float4 *)((unsigned char *)workspace + off);
    hipLaunchKernelGGL(fps_sort_kernel, dim3(B), dim3(SORT_T), sizeof(int) * SORT_CELLS, st, xyz, N, perm);
    hipLaunchKernelGGL(fps_records_kernel, dim3(64, B), dim3(256), 0, st, xyz, perm, N, NP, rec);
    if (get_option(OPT_FPS_VARIANT) == 6) {      // the first form of the kernel
        if (NP == 16384) hipLaunchKernelGGL((fps_cellg_kernel<16>), dim3(B), dim3(1024), 0, st, rec, N, NP, M, idx);
        else hipLaunchKernelGGL((fps_cellg_kernel<64>), dim3(B), dim3(1024), 0, st, rec, N, NP, M, idx);
    } else {
        if (NP == 16384) hipLaunchKernelGGL((fps_cellg2_kernel<16>), dim3(B), dim3(1024), 0, st, rec, N, NP, M, idx);
        else hipLaunchKernelGGL((fps_cellg2_kernel<64>), dim3(B), dim3(1024), 0, st, rec, N, NP, M, idx);
    }
    return check_launch("sad_fps_f32 (cell, global records)");
}

// Called by sad_fps_f32 when a workspace (B*N ints) is available and N <= 16384.
int launch_fps_bucket(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&fps_sort_kernel), 80 * 1024);
    int *perm = (int *)workspace;
    hipLaunchKernelGGL(fps_sort_kernel, dim3(B), dim3(SORT_T), sizeof(int) * SORT_CELLS, st, xyz, N, perm);
    if (int e = check_launch("sad_fps_f32 (sort)")) return e;
    if (get_option(OPT_FPS_VARIANT) != 3) {   // default: cell-bucket kernel (3 = wave/thread-bucket kernel, 6 = first form of the cell kernel)
        // geometry = waves * 100 + slots; fps_threads option overrides (for sweeps)
        const int nb = (N + 63) / 64;                       // buckets
        int geo = get_option(OPT_FPS_THREADS);
        if (geo < 100) geo = nb > 128 ? 1616 : (nb > 32 ? 816 : (nb > 16 ? 408 : 404));
        const int nw = geo / 100, ppt = geo % 100;
        if (nw * ppt < nb || !(nw == 1 || nw == 2 || nw == 4 || nw == 8 || nw == 16) ||
            !(ppt == 4 || ppt == 8 || ppt == 16 || ppt == 32))
            return fail(SAD_EINVAL, "sad_fps_f32: cell geometry %d cannot hold %d buckets", geo, nb);
        if (get_option(OPT_FPS_VARIANT) == 7) {                 // third form: several samples per round
#define SAD_CELL3(NW_, PPT_) case NW_ * 100 + PPT_: launch_cell3<NW_, PPT_>(xyz, perm, B, N, M, idx, st); break;
            switch (geo) {
                SAD_CELL3(16, 4) SAD_CELL3(16, 8) SAD_CELL3(16, 16)
                SAD_CELL3(8, 4) SAD_CELL3(8, 8) SAD_CELL3(8, 16) SAD_CELL3(8, 32)
                SAD_CELL3(4, 4) SAD_CELL3(4, 8) SAD_CELL3(4, 16) SAD_CELL3(4, 32)
                default: return fail(SAD_EINVAL, "sad_fps_f32: cell geometry %d not built for the third form", geo);
            }
#undef SAD_CELL3
            return check_launch("sad_fps_f32 (cell, several samples per round)");
        }
        if (get_option(OPT_FPS_VARIANT) != 6 && nw >= 2) {      // second form of the cell kernel (lane-direct publish); 6 = the first form
#define SAD_CELL2(NW_, PPT_) case NW_ * 100 + PPT_: launch_cell2<NW_, PPT_>(xyz, perm, B, N, M, idx, st); break;
            switch (geo) {
                SAD_CELL2(16, 4) SAD_CELL2(16, 8) SAD_CELL2(16, 16)
                SAD_CELL2(8, 4) SAD_CELL2(8, 8) SAD_CELL2(8, 16) SAD_CELL2(8, 32)
                SAD_CELL2(4, 4) SAD_CELL2(4, 8) SAD_CELL2(4, 16) SAD_CELL2(4, 32)
                SAD_CELL2(2, 8) SAD_CELL2(2, 16) SAD_CELL2(2, 32)
                default: return fail(SAD_EINVAL, "sad_fps_f32: cell geometry %d not built", geo);
            }
#undef SAD_CELL2
            return check_launch("sad_fps_f32 (cell, lane-direct)");
        }
#define SAD_CELL(NW_, PPT_) case NW_ * 100 + PPT_: launch_cell<NW_, PPT_>(xyz, perm, B, N, M, idx, st); break;
        switch (geo) {
            SAD_CELL(16, 4) SAD_CELL(16, 8) SAD_CELL(16, 16)
            SAD_CELL(8, 4) SAD_CELL(8, 8) SAD_CELL(8, 16) SAD_CELL(8, 32)
            SAD_CELL(4, 4) SAD_CELL(4, 8) SAD_CELL(4, 16) SAD_CELL(4, 32)
            SAD_CELL(2, 8) SAD_CELL(2, 16) SAD_CELL(2, 32)
            SAD_CELL(1, 16) SAD_CELL(1, 32)
            default: return fail(SAD_EINVAL, "sad_fps_f32: cell geometry %d not built", geo);
        }
#undef SAD_CELL
        return check_launch("sad_fps_f32 (cell)");
    }
    if (N <= 2048) {
        const int ppt = (N + 255) / 256;
        if (ppt <= 2) launch_bucket<256, 2>(xyz, perm, B, N, M, idx, st);
        else if (ppt <= 4) launch_bucket<256, 4>(xyz, perm, B, N, M, idx, st);
        else launch_bucket<256, 8>(xyz, perm, B, N, M, idx, st);
    } else {
        // Fewer, fatter waves: the per-step chain (publish, barrier, 16-key arg-max, coordinate
        // read) is executed by every wave, so with most waves skipping their update the step time
        // is that chain times the waves sharing a SIMD.  fps_threads option: 1024 / 512 / 256.
        const int th = get_option(OPT_FPS_THREADS);
        if (th == 256 && N <= 8192) {
            if (N <= 4096) launch_bucket<256, 16>(xyz, perm, B, N, M, idx, st);
            else launch_bucket<256, 32>(xyz, perm, B, N, M, idx, st);
        } else if (th == 512 || th == 256) {
            if (N <= 4096) launch_bucket<512, 8>(xyz, perm, B, N, M, idx, st);
            else if (N <= 8192) launch_bucket<512, 16>(xyz, perm, B, N, M, idx, st);
            else launch_bucket<512, 32>(xyz, perm, B, N, M, idx, st);
        } else {
            const int ppt = (N + 1023) / 1024;
            if (ppt <= 4) launch_bucket<1024, 4>(xyz, perm, B, N, M, idx, st);
            else if (ppt <= 8) launch_bucket<1024, 8>(xyz, perm, B, N, M, idx, st);
            else launch_bucket<1024, 16>(xyz, perm, B, N, M, idx, st);
        }
    }
    return check_launch("sad_fps_f32 (bucket)");
}

}  // namespace sad
