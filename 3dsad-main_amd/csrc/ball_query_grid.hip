// Grid-pruned ball query for gfx950 (SPEC.md §3): same answers as the sequential scan, but each
// centroid only tests the points of the 27 grid cells around it.  No reference source exists
// (/root/reference/README.md:1-2).
//
// Why: the brute-force kernel performs N*M pair tests (201 M per 16 384-point scene at SA1) for
// 0.77 MB of compulsory bytes per radius — ~87 tests per byte, VALU-bound at <1 % of the HBM
// roofline.  A uniform grid with cell edge > r_max cuts the tests by ~100x.
//
// How index order survives pruning: candidates arrive in cell order, not index order.  A centroid with
// at most 128 candidates (99.95 % of a KITTI-shaped scene) sorts them by index in registers (one or two
// keys per lane, bitonic network) and compacts the accepted ones per radius with a ballot; a denser one
// sets one bit per accepted index in an LDS bitmap (N bits per radius, one set per workgroup under a
// lock) and scans it in ascending bit order with a wave-wide popcount prefix.  Either way the first
// `nsample` accepted indices come out in ascending index order, exactly what SPEC.md §3's scan produces.
//
// Exactness: the accept test is the same sad::d2f(point, centroid) < r*r.  Pruning is conservative:
// cell = floor((x - x0) * inv) is a monotone function of x in binary32, the cell edge is r_max *
// 1.001 (or larger), so a point within r_max of the centroid lies at most one cell away per axis.
#include "common.h"

namespace {

constexpr int GRID_MAXC = 32768;   // cells per scene (LDS histogram: 128 KB of the CU's 160 KB; the build runs one workgroup per CU)
// The cell edge starts at 1.001 r_max and grows by this factor until the grid fits GRID_MAXC cells.  Round 1-3 doubled it
// with 16 384 cells: a KITTI-shaped scene (70 x 80 x 2 m) at r_max = 0.8 ended with 1.6 m cells — 27 cells = a 4.8 m cube,
// ~70 candidates per centroid of which ~3 are accepted — where 0.8 m cells (26 400 of them) give ~17.  Any edge >= r_max is
// exact (header comment); the candidate count goes with its square.
constexpr float GRID_GROW = 1.18920712f;   // 2^(1/4)
constexpr int BUILD_T = 1024;

struct GridHdr {      // 16 floats / ints at the start of each scene's workspace block
    float x0, y0, z0, inv;
    int gx, gy, gz, ncell;
    float invz;           // cell scale along z: `inv`, or 0 when the z split was given up (every point and centroid in layer 0)
    int pad[7];
};

__host__ __device__ inline size_t scene_ws_bytes(int N) {
    return sizeof(GridHdr) + sizeof(int) * (size_t)(GRID_MAXC + 16) + sizeof(float4) * (size_t)N;
}

__device__ __forceinline__ int cell_coord(float x, float x0, float inv, int g) {
    float t = (x - x0) * inv;
    t = t < -2.f ? -2.f : t;
    const float hi = (float)(g + 1);
    t = t > hi ? hi : t;
    return (int)floorf(t);
}

// ---- build: bounding box, cell histogram, exclusive scan, scatter of (x,y,z,idx) records ---------
// One 1024-thread workgroup per scene.  Scenes of up to 16 384 points keep their points in REGISTERS (PTS
// points per thread, read once, coalesced) through all four phases — the three passes over global memory of
// the first version (bounding box, histogram, scatter) made the SA1 build 37 us, a third of the query it
// serves; larger scenes (N <= 65 536) walk global memory as before (PTS = 0).
template <int PTS>
__global__ __launch_bounds__(BUILD_T) void grid_build_kernel(const float *__restrict__ xyz, int N,
                                                             float cs_min, char *__restrict__ ws) {
    extern __shared__ int hist[];               // GRID_MAXC + 64 ints
    __shared__ float red[6][16];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = xyz + (size_t)blockIdx.x * N * 3;
    char *base = ws + (size_t)blockIdx.x * scene_ws_bytes(N);
    GridHdr *hdr = reinterpret_cast<GridHdr *>(base);
    int *cell_start = reinterpret_cast<int *>(base + sizeof(GridHdr));
    float4 *rec = reinterpret_cast<float4 *>(base + sizeof(GridHdr) + sizeof(int) * (size_t)(GRID_MAXC + 16));
    constexpr int NP = PTS > 0 ? PTS : 1;
    float px[NP], py[NP], pz[NP];
    if constexpr (PTS > 0) {
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            const int j = tid + k * BUILD_T;
            const int jj = j < N ? j : N - 1;
            px[k] = p[jj * 3]; py[k] = p[jj * 3 + 1]; pz[k] = p[jj * 3 + 2];
        }
    }
    auto for_points = [&](auto &&fn) {          // fn(j, x, y, z) for this thread's points
        if constexpr (PTS > 0) {
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                const int j = tid + k * BUILD_T;
                if (j < N) fn(j, px[k], py[k], pz[k]);
            }
        } else {
            for (int j = tid; j < N; j += BUILD_T) fn(j, p[j * 3], p[j * 3 + 1], p[j * 3 + 2]);
        }
    };

    // 1. bounding box
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for_points([&](int, float x, float y, float z) {
        lo[0] = x < lo[0] ? x : lo[0]; hi[0] = x > hi[0] ? x : hi[0];
        lo[1] = y < lo[1] ? y : lo[1]; hi[1] = y > hi[1] ? y : hi[1];
        lo[2] = z < lo[2] ? z : lo[2]; hi[2] = z > hi[2] ? z : hi[2];
    });
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
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d)
        for (int w = 0; w < 16; ++w) {
            lo[d] = red[d][w] < lo[d] ? red[d][w] : lo[d];
            hi[d] = red[3 + d][w] > hi[d] ? red[3 + d][w] : hi[d];
        }
    // 2. grid geometry (identical in every thread): grow the cell edge by 2^(1/4) per trial until the grid fits
    // Non-finite coordinates are undefined behaviour for the RESULT (SPEC.md §3), never for memory:
    // an extent that is not a finite non-negative number (Inf / NaN input, or no finite point at all)
    // selects a one-cell grid — every query then tests every point, like the scan kernel — and the
    // growth loop is bounded (128 trials = a factor of 2^32 on the edge: a radius / extent ratio beyond that
    // also takes the one-cell grid, which is still exact), so the kernel neither spins nor leaves its tables.
    bool sane = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float e = hi[d] - lo[d];
        sane = sane && (e >= 0.f) && (e < 1.0e30f);
    }
    sane = sane && (cs_min > 0.f) && (cs_min < 1.0e30f);
    float cs = cs_min;
    bool flat = false;
    int gx = 1, gy = 1, gz = 1;
    for (int it = 0; sane; ++it) {
        const float inv = 1.0f / cs;
        const float fx = (hi[0] - lo[0]) * inv, fy = (hi[1] - lo[1]) * inv, fz = (hi[2] - lo[2]) * inv;
        if (it >= 128 || !(fx < 2.0e9f) || !(fy < 2.0e9f) || !(fz < 2.0e9f)) {
            if (it >= 128) { sane = false; break; }
            cs = cs * 2.0f;                 // still far too fine for an int cell count
            continue;
        }
        gx = (int)fx + 1;
        gy = (int)fy + 1;
        gz = (int)fz + 1;
        if ((long long)gx * gy * gz <= GRID_MAXC) break;
        // flat scenes (lidar: a few cells high): give up the z split before coarsening x and y — one cell in z keeps the
        // candidate count of the fine x-y grid (its three z layers were all visited anyway); 128 x 128 x 3 cells of a
        // 102 m nuScenes-shaped scene at r_max = 0.8 do not fit, 128 x 128 x 1 do
        if (gz <= 4 && (long long)gx * gy <= GRID_MAXC) { gz = 1; flat = true; break; }
        cs = cs * GRID_GROW;
    }
    if (!sane) {
        gx = gy = gz = 1;
        lo[0] = lo[1] = lo[2] = 0.f;
    }
    const float inv = sane ? 1.0f / cs : 0.f;
    const float invz = flat ? 0.f : inv;
    const int ncell = gx * gy * gz;
    if (tid == 0) {
        hdr->x0 = lo[0]; hdr->y0 = lo[1]; hdr->z0 = lo[2]; hdr->inv = inv;
        hdr->gx = gx; hdr->gy = gy; hdr->gz = gz; hdr->ncell = ncell;
        hdr->invz = invz;
    }
    // 3. histogram (only the cells this grid has, rounded up to whole threads of the scan, are ever touched)
    for (int c = tid; c < BUILD_T * ((ncell + BUILD_T - 1) / BUILD_T); c += BUILD_T) hist[c] = 0;
    __syncthreads();
    auto cell_of = [&](float x, float y, float z) {
        int ix = cell_coord(x, lo[0], inv, gx), iy = cell_coord(y, lo[1], inv, gy), iz = cell_coord(z, lo[2], invz, gz);
        ix = ix < 0 ? 0 : (ix > gx - 1 ? gx - 1 : ix);
        iy = iy < 0 ? 0 : (iy > gy - 1 ? gy - 1 : iy);
        iz = iz < 0 ? 0 : (iz > gz - 1 ? gz - 1 : iz);
        return (iz * gy + iy) * gx + ix;
    };
    for_points([&](int, float x, float y, float z) { atomicAdd(&hist[cell_of(x, y, z)], 1); });
    __syncthreads();
    // 4. exclusive scan over the cells of THIS grid (ncell <= GRID_MAXC): thread t owns cells [t*cpt, (t+1)*cpt)
    const int cpt = (ncell + BUILD_T - 1) / BUILD_T;       // <= GRID_MAXC / BUILD_T = 32
    int sum = 0;
    for (int k = 0; k < cpt; ++k) sum += hist[tid * cpt + k];
    int incl = sum;                             // inclusive scan of `sum` over the wave
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    int run = wbase + incl - sum;
    for (int k = 0; k < cpt; ++k) {
        const int c = tid * cpt + k;
        const int v = hist[c];
        hist[c] = run;                          // becomes the running write offset of the cell
        cell_start[c] = run;                    // (cells in [ncell, BUILD_T * cpt) are empty: start = N; none beyond is ever read)
        run += v;
    }
    if (tid == BUILD_T - 1) cell_start[BUILD_T * cpt] = run;   // = N (covers cell_start[ncell] when ncell is a multiple of cpt)
    __syncthreads();
    // 5. scatter records (order inside a cell is irrelevant: the query restores index order)
    for_points([&](int j, float x, float y, float z) {
        const int pos = atomicAdd(&hist[cell_of(x, y, z)], 1);
        rec[pos] = make_float4(x, y, z, __int_as_float(j));
    });
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (the shuffle version is six dependent
// ds_bpermute round trips)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);    // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);    // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- query -------------------------------------------------------------------------------------
struct GQParams {
    float radii[SAD_MAX_RADII];
    int nsample[SAD_MAX_RADII];
    int32_t *idx[SAD_MAX_RADII];
    int32_t *cnt[SAD_MAX_RADII];   // optional: accepted points per centroid, capped at nsample
    int no_sort;                   // A/B knob (bq_variant = 1): always the bitmap path
};

constexpr int GQ_WAVES = 4;
constexpr int GQ_CPW = 4;     // centroids per wave, processed one after the other

// ---- sort path: a centroid with at most 64 candidates (the common case on lidar-density scenes) needs no bitmap --------
// One candidate per lane: key = (index << 4) | (bit r set iff d2 < radius_r^2), empty lanes = 0xFFFFFFF0.  A 64-lane bitonic
// sort by key (indices are unique) puts the candidates in ascending index order; for every radius the accepted ones are
// then compacted with a ballot + prefix popcount: slot = number of accepted lanes below.  No LDS memory, no atomics, no
// data-dependent loops: 21 exchange stages against the bitmap path's ~100 instructions per radius.
// The kernel is bound by VALU issue (rocprofv3: 1 650 vector instructions per wave, the SIMD's vector unit ~90 % busy at five
// waves), so the network is the one that needs the fewest of them: every merge starts with a MIRROR exchange (lane i against
// lane k-1-i of its block) and continues with half-cleaners (i against i^d) — all blocks ascend, the lane that keeps the
// minimum is always the one with bit d clear, and the six lane masks are compile-time constants handed to v_cndmask as scalar
// operands (a direction-dependent network needs 21 masks: the compiler kept them in spilled SGPRs, two v_readlane per stage).
// The partner comes through the LDS crossbar (ds_swizzle / ds_bpermute: no vector instruction): three VALU per stage.
template <int X>
__device__ __forceinline__ unsigned swz_xor(unsigned v) {                     // value of lane ^ X, X < 32 (bit-mask mode: and 0x1F, or 0, xor X)
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1F);
}
template <int D>
__device__ __forceinline__ unsigned long long low_mask() {                    // lanes with bit D clear
    return D == 1 ? 0x5555555555555555ull : D == 2 ? 0x3333333333333333ull : D == 4 ? 0x0F0F0F0F0F0F0F0Full
         : D == 8 ? 0x00FF00FF00FF00FFull : D == 16 ? 0x0000FFFF0000FFFFull : 0x00000000FFFFFFFFull;
}
template <int D>
__device__ __forceinline__ unsigned exchange(unsigned v, unsigned p) {        // lanes with bit D clear keep the minimum
    const unsigned lo = v < p ? v : p, hi = v < p ? p : v;
    unsigned r;
    const unsigned long long m = low_mask<D>();
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "s"(m));
    return r;
}
template <int D>
__device__ __forceinline__ unsigned clean_from(unsigned v) {                  // half-cleaners D, D/2 .. 1
    if constexpr (D >= 1) {
        v = exchange<D>(v, swz_xor<D>(v));
        v = clean_from<D / 2>(v);
    }
    return v;
}
template <int K>
__device__ __forceinline__ unsigned merge_upto32(unsigned v) {                // merges of blocks 2 .. K (K <= 32)
    if constexpr (K >= 2) {
        v = merge_upto32<K / 2>(v);
        v = exchange<K / 2>(v, swz_xor<K - 1>(v));                           // mirror inside the block of K
        v = clean_from<K / 4>(v);
    }
    return v;
}
// mir63 = 4 * (63 - lane), x32 = 4 * (lane ^ 32): ds_bpermute addresses
__device__ __forceinline__ unsigned bitonic_sort64(unsigned v, int mir63) {
    v = merge_upto32<32>(v);
    v = exchange<32>(v, (unsigned)__builtin_amdgcn_ds_bpermute(mir63, (int)v));
    return clean_from<16>(v);
}
// The same sort when only lanes 0 .. n-1 hold real keys and every other lane the empty key 0xFFFFFFF0 (which sorts last and is equal to
// itself): once the blocks of the smallest power of two >= n are sorted, lanes 0 .. n-1 are in their final order and the wider merges could
// only move empty keys among empty keys — they are skipped (n is wave-uniform: a scalar branch).  n <= 16: 10 of the 21 stages; n <= 32: 15.
__device__ __forceinline__ unsigned bitonic_sort64n(unsigned v, int mir63, int n) {
    v = merge_upto32<16>(v);
    if (n > 16) {
        v = exchange<16>(v, swz_xor<31>(v));                                    // merge of the 32-lane blocks (as merge_upto32<32>)
        v = clean_from<8>(v);
        if (n > 32) {
            v = exchange<32>(v, (unsigned)__builtin_amdgcn_ds_bpermute(mir63, (int)v));
            v = clean_from<16>(v);
        }
    }
    return v;
}
// 128 keys, two per lane: element e = lane + 64 * register
__device__ __forceinline__ void bitonic_sort128(unsigned &v0, unsigned &v1, int mir63, int x32) {
    v0 = bitonic_sort64(v0, mir63);
    v1 = bitonic_sort64(v1, mir63);
    const unsigned p0 = (unsigned)__builtin_amdgcn_ds_bpermute(mir63, (int)v1), p1 = (unsigned)__builtin_amdgcn_ds_bpermute(mir63, (int)v0);
    v0 = v0 < p0 ? v0 : p0;                                                   // element e against 127 - e
    v1 = v1 < p1 ? p1 : v1;
    v0 = exchange<32>(v0, (unsigned)__builtin_amdgcn_ds_bpermute(x32, (int)v0));
    v1 = exchange<32>(v1, (unsigned)__builtin_amdgcn_ds_bpermute(x32, (int)v1));
    v0 = clean_from<16>(v0);
    v1 = clean_from<16>(v1);
}

__device__ __forceinline__ float readlane_f(float v, int l) {                  // (the builtin takes ints: a float argument would be CONVERTED)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// (eight waves per SIMD = at most 64 registers: an FPS workgroup of another batch leaves exactly 64 per SIMD on its CU, so this
// kernel still finds room on the 96 CUs three sampling chains hold; at 65 registers the pipelined step was 0.6 % (f32) / 1.5 % (bf16)
// slower)
template <int NR>
__global__ __launch_bounds__(GQ_WAVES * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void grid_query_kernel(const float *__restrict__ new_xyz,
                                                                   const char *__restrict__ ws,
                                                                   GQParams prm, int N, int M, int B, int nbx, int nsets) {
    extern __shared__ unsigned bm_all[];        // nsets x (NR * (NWP + 64) words (zero between centroids) + the lock word)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroups go to the 8 XCDs round-robin; all centroid blocks of scene
    // xcd + 8*i run on one XCD, so its L2 holds that scene's grid records once
    const int slot = blockIdx.x >> 3;
    const int b = (blockIdx.x & 7) + 8 * (slot / nbx);
    const int bx = slot - (slot / nbx) * nbx;
    const int NW = (N + 31) >> 5;               // bitmap words
    int wshift = 0;                             // words per lane in the scan: 2^wshift (<= 32 for N <= 65536)
    while ((64 << wshift) < NW) ++wshift;
    const int WPL = 1 << wshift;
    const int NWP = WPL * 64;                   // padded words per bitmap
    // Bitmap sets are shared under a lock: only centroids with more than 128 candidates use one (0.05 % of a KITTI-shaped
    // scene), and a set per wave (27 KB per workgroup at N = 16 384) capped the kernel at five waves per SIMD.  ONE set per
    // workgroup (round 3) made the four waves of a workgroup take turns on dense scenes, where most centroids come here
    // (20 m x 20 m scenes: SA1 query 0.205 -> 0.369 ms); with a set per TWO waves (`nsets` = 2 whenever two sets still leave
    // room for eight workgroups per CU) a wave shares its lock with one sibling.
    const int SETW = NR * (NWP + 64) + 1;       // words of a set, its lock word last
    unsigned *bm = bm_all + (nsets == 2 ? (wave >> 1) * SETW : 0);
    unsigned *lock = bm + NR * (NWP + 64);
    // second level: dm[r*64 + l] has bit k set iff word l*WPL + k of bitmap r is non-zero, so the
    // scan touches only the words that received a bit (cost ~ accepted points, not N)
    unsigned *dm = bm + NR * NWP;
    // (the grid is padded to 8 * ceil(B / 8) scenes: the guard comes before anything is read through `base`, whose block
    // lies past the end of the workspace for the padding scenes)
    if (b >= B) return;                         // (workgroup-uniform)
    const char *base = ws + (size_t)b * scene_ws_bytes(N);
    const GridHdr *hdr = reinterpret_cast<const GridHdr *>(base);
    const int *cell_start = reinterpret_cast<const int *>(base + sizeof(GridHdr));
    const float4 *rec = reinterpret_cast<const float4 *>(base + sizeof(GridHdr) + sizeof(int) * (size_t)(GRID_MAXC + 16));
    const float x0 = hdr->x0, y0 = hdr->y0, z0 = hdr->z0, inv = hdr->inv, invz = hdr->invz;
    const int gx = hdr->gx, gy = hdr->gy, gz = hdr->gz;

    // Only the lock words are initialised here; a bitmap set is zeroed by the first wave that takes it (under its lock): the
    // sort paths never touch the sets, and zeroing 14 KB per workgroup cost every wave ~60 vector instructions of its ~700.
    if (threadIdx.x < (unsigned)nsets) bm_all[threadIdx.x * SETW + NR * (NWP + 64)] = 0u;
    __syncthreads();
    bool set_clean = false;                     // this wave has zeroed (or already used and cleaned) its set

    // The wave's GQ_CPW = 4 centroids are independent: their dependent memory round trips (centroid -> cell starts -> first
    // records) are issued for all of them before any is processed, and the bookkeeping in front of the records is done for
    // all four at once: lane 16 c + r holds run r (of nine: three x-adjacent cells each) of centroid c.
    static_assert(GQ_CPW == 4, "the prologue lays four centroids out in the four rows of 16 lanes");
    const int mir63 = 4 * (63 - lane), x32 = 4 * (lane ^ 32);
    const int pcen = lane >> 4, prun = lane & 15;
    float r2[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) r2[r] = prm.radii[r] * prm.radii[r];
    const int m0 = (bx * GQ_WAVES + wave) * GQ_CPW;
    const int mq = m0 + pcen < M ? m0 + pcen : M - 1;
    const float *qp = new_xyz + ((size_t)b * M + mq) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    int rsv = 0, rlv = 0;                       // start / length of this lane's run
    {
        const int ix = cell_coord(qx, x0, inv, gx), iy = cell_coord(qy, y0, inv, gy), iz = cell_coord(qz, z0, invz, gz);
        const int dy = prun % 3 - 1, dz = prun / 3 - 1;
        const int yy = iy + dy, zz = iz + dz;
        const int xlo = ix - 1 < 0 ? 0 : ix - 1, xhi = ix + 1 > gx - 1 ? gx - 1 : ix + 1;
        if (prun < 9 && m0 + pcen < M && yy >= 0 && yy < gy && zz >= 0 && zz < gz && xlo <= xhi) {     // (no runs for a centroid past the end)
            const int c0 = (zz * gy + yy) * gx;
            rsv = cell_start[c0 + xlo];
            rlv = cell_start[c0 + xhi + 1] - rsv;
        }
    }
    // offsets of the runs inside their centroid's candidate list: inclusive scan over each row of 16 lanes (DPP row shifts)
    int incl = rlv;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
    const int offv = incl - rlv;                // candidate i of the centroid lies in run r iff offv[r] <= i < offv[r] + rlv[r]
    const int delv = rsv - offv;                // ... and is record i + delv[r]
    const int T0 = __builtin_amdgcn_readlane(incl, 15), T1 = __builtin_amdgcn_readlane(incl, 31);
    const int T2 = __builtin_amdgcn_readlane(incl, 47), T3 = __builtin_amdgcn_readlane(incl, 63);
    // record of candidate i (per lane) of centroid c (wave-uniform): the last run whose offset is <= i
    auto record_of = [&](int c, int i) -> int {
        int src = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int o = __builtin_amdgcn_readlane(offv, 16 * c + r), d = __builtin_amdgcn_readlane(delv, 16 * c + r);
            src = i >= o ? i + d : src;
        }
        return src;
    };
    // ---- packed path (round 4): the four centroids of the wave in ONE sort ------------------------------------------------
    // The pair / single paths below sort every CANDIDATE of a centroid (two to four sorts and compactions per wave, and a
    // nine-step record walk per candidate list: 1 180 vector instructions per wave, rocprofv3) although on lidar-density
    // scenes only ~15 % of the candidates are accepted by any radius.  Here
    //  * the runs are expanded through LDS: lane 16 c + r (run r of centroid c) writes the record numbers of its run into
    //    the wave's table tab[c][offset + i] — a few fire-and-forget ds_write, the four centroids side by side — and the
    //    candidate lists are then read back one candidate per lane (no record walk, every record load issued at once);
    //  * a candidate is KEPT only if some radius accepts it: ballot + prefix count give it its place behind the ones kept so
    //    far, a forward ds_permute moves its key (centroid << 28 | index << 4 | accept bits) there;
    //  * when the four centroids keep at most 64 points together (the rule on KITTI- and nuScenes-shaped scenes) ONE 64-lane
    //    sort orders all of them, and per radius a ballot restricted to the lane's own centroid gives the slot.
    // Waves with a centroid of more than 64 candidates, or more than 64 kept points, take the older paths below.
    const int Tmax = max(max(T0, T1), max(T2, T3));
    if (!prm.no_sort && Tmax <= 64) {
        int *tab = reinterpret_cast<int *>(bm_all + nsets * SETW) + wave * 256;      // [4 centroids][64 candidates]
        int ml = rlv;                           // longest run of the wave
        ml = max(ml, __builtin_amdgcn_update_dpp(0, ml, 0xB1, 0xF, 0xF, true));
        ml = max(ml, __builtin_amdgcn_update_dpp(0, ml, 0x4E, 0xF, 0xF, true));
        ml = max(ml, __builtin_amdgcn_update_dpp(0, ml, 0x141, 0xF, 0xF, true));
        ml = max(ml, __builtin_amdgcn_update_dpp(0, ml, 0x140, 0xF, 0xF, true));
        const int maxlen = max(max(__builtin_amdgcn_readlane(ml, 0), __builtin_amdgcn_readlane(ml, 16)),
                               max(__builtin_amdgcn_readlane(ml, 32), __builtin_amdgcn_readlane(ml, 48)));
        int *trow = tab + pcen * 64 + offv;
        for (int it = 0; it < maxlen; ++it)     // (wave-uniform trip count; offv + rlv <= T <= 64)
            if (it < rlv) trow[it] = rsv + it;
        unsigned packed = 0xFFFFFFF0u;          // lane p: the p-th kept key (empty: sorts last, accepted by no radius)
        int nacc = 0;                           // keys kept so far (wave-uniform)
        bool fits = true;
        // (same wave: the table reads below are ordered behind the writes by the LDS queue and the compiler's lgkmcnt wait)
        const bool pair0 = T0 <= 32 && T1 <= 32, pair1 = T2 <= 32 && T3 <= 32;
        const int half = lane >> 5, hi = lane & 31;
        // record numbers and records of all passes first (independent loads: one round trip for the wave), then the keys
        bool val[4];
        float4 prs[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool paired = p ? pair1 : pair0;
            const int Ta = p ? T2 : T0, Tb = p ? T3 : T1;
            if (paired) {
                val[2 * p] = hi < (half ? Tb : Ta);
                prs[2 * p] = rec[val[2 * p] ? tab[(2 * p + half) * 64 + hi] : 0];
                val[2 * p + 1] = false;
                prs[2 * p + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                val[2 * p] = lane < Ta;
                prs[2 * p] = rec[val[2 * p] ? tab[(2 * p) * 64 + lane] : 0];
                val[2 * p + 1] = lane < Tb;
                prs[2 * p + 1] = rec[val[2 * p + 1] ? tab[(2 * p + 1) * 64 + lane] : 0];
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool paired = p ? pair1 : pair0;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                if (paired && h2) continue;     // (wave-uniform)
                const int q = 2 * p + h2;
                if (!paired && (q == 0 ? T0 : q == 1 ? T1 : q == 2 ? T2 : T3) == 0) continue;
                const float4 pr = prs[q];
                const int ca = paired ? 2 * p : q, cb = paired ? 2 * p + 1 : q;
                const float cxa = readlane_f(qx, 16 * ca), cya = readlane_f(qy, 16 * ca), cza = readlane_f(qz, 16 * ca);
                float cx = cxa, cy = cya, cz = cza;
                unsigned cidx = (unsigned)ca;
                if (paired) {
                    const float cxb = readlane_f(qx, 16 * cb), cyb = readlane_f(qy, 16 * cb), czb = readlane_f(qz, 16 * cb);
                    cx = half ? cxb : cxa; cy = half ? cyb : cya; cz = half ? czb : cza;
                    cidx = half ? (unsigned)cb : (unsigned)ca;
                }
                const float d = sad::d2f(pr.x, pr.y, pr.z, cx, cy, cz);
                unsigned acc = 0u;
#pragma unroll
                for (int r = 0; r < NR; ++r) acc |= d < r2[r] ? 1u << r : 0u;
                const bool ok = val[q] && acc != 0u;
                const unsigned long long bal = __ballot(ok);
                const int nok = __builtin_popcountll(bal);
                if (nacc + nok > 64) fits = false;
                if (fits) {                     // (wave-uniform)
                    const unsigned key = (cidx << 28) | ((unsigned)__float_as_int(pr.w) << 4) | acc;
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    // forward permute: lane nacc + rank receives this key; the lanes that keep nothing send theirs to lane
                    // nacc + nok (mod 64), outside [nacc, nacc + nok) — whatever lands outside that range is discarded
                    const int dst = ok ? nacc + rank : nacc + nok;
                    const unsigned got = (unsigned)__builtin_amdgcn_ds_permute(dst << 2, (int)key);
                    packed = (unsigned)(lane - nacc) < (unsigned)nok ? got : packed;
                    nacc += nok;
                }
            }
        }
        if (fits) {
            const unsigned key = bitonic_sort64n(packed, mir63, nacc);      // (the kept keys sit in lanes 0 .. nacc-1)
            const int jidx = (int)((key >> 4) & 0x00FFFFFFu);
            const unsigned cl = key >> 28;      // the lane's centroid (15: an empty lane)
            // the lanes of each centroid (contiguous after the sort), and per lane the mask of its own centroid's lanes
            unsigned long long seg[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) seg[c] = __ballot(cl == (unsigned)c);
            unsigned mlo = 0u, mhi = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mlo = cl == (unsigned)c ? (unsigned)seg[c] : mlo;
                mhi = cl == (unsigned)c ? (unsigned)(seg[c] >> 32) : mhi;
            }
            const unsigned cm = cl < 4u ? cl : 0u;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int S = prm.nsample[r];
                int32_t *out0 = prm.idx[r] + ((size_t)b * M + m0) * S;
                const bool ok = (key >> r) & 1u;
                const unsigned long long bal = __ballot(ok);
                const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32) & mhi, __builtin_amdgcn_mbcnt_lo((unsigned)bal & mlo, 0u));
                if (ok && slot < (unsigned)S) out0[cm * (unsigned)S + slot] = jidx;
                // SPEC.md §3 padding: the remaining slots repeat the first (lowest) accepted index; none accepted: zeros
                int tot[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (m0 + c >= M) { tot[c] = 0; continue; }      // (wave-uniform)
                    const unsigned long long bc = bal & seg[c];
                    const int total = __builtin_popcountll(bc);
                    int first = 0;
                    if (bc) first = __builtin_amdgcn_readlane(jidx, __builtin_ctzll(bc));
#ifndef SAD_BQ_NOPAD            // (measurement build: the padding stores of the packed path left out — what they cost; consumers that take `cnt` never read them)
                    if (lane >= total && lane < S) out0[c * S + lane] = first;       // (nsample <= 64: one store covers the row)
#endif
                    tot[c] = total < S ? total : S;
                }
                if (prm.cnt[r] && lane < 4 && m0 + lane < M)
                    prm.cnt[r][(size_t)b * M + m0 + lane] = lane == 0 ? tot[0] : lane == 1 ? tot[1] : lane == 2 ? tot[2] : tot[3];
            }
            return;
        }
    }

    // Two neighbouring centroids with at most 32 candidates each (most pairs of a lidar-density scene: mean 20) share ONE pass:
    // lanes 0..31 carry the first, lanes 32..63 the second — keys, a 32-lane sort per half and the compaction cost the same
    // instructions for two centroids as for one.
    const bool pair0 = T0 <= 32 && T1 <= 32 && m0 + 1 < M && !prm.no_sort;
    const bool pair1 = T2 <= 32 && T3 <= 32 && m0 + 3 < M && !prm.no_sort;
    const int half = lane >> 5, hi = lane & 31;
    float4 pra[2], prb[2];                      // records of the pair's first / second centroid (paired: both in pra)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool paired = p ? pair1 : pair0;
        const int Ta = p ? T2 : T0, Tb = p ? T3 : T1;
        if (paired) {
            const int sa = record_of(2 * p, hi), sb = record_of(2 * p + 1, hi);
            const bool valid = hi < (half ? Tb : Ta);
            pra[p] = rec[valid ? (half ? sb : sa) : 0];
            prb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            pra[p] = rec[lane < Ta ? record_of(2 * p, lane) : 0];
            prb[p] = rec[lane < Tb ? record_of(2 * p + 1, lane) : 0];
        }
    }
    // ONE copy of the per-pair code (the records rotate through registers): unrolled, the kernel was 160 KB of instructions,
    // more than the instruction cache two CUs share
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        if (m0 + 2 * p >= M) break;             // wave-uniform
        const bool paired = p ? pair1 : pair0;
        const float4 pa = pra[0], pb = prb[0];
        pra[0] = pra[1]; prb[0] = prb[1];
        if (paired) {
            // ---- two centroids, one per half wave ----
            const int ca = 2 * p;
            const int Ta = p ? T2 : T0, Tb = p ? T3 : T1;
            const float cxa = readlane_f(qx, 16 * ca), cya = readlane_f(qy, 16 * ca), cza = readlane_f(qz, 16 * ca);
            const float cxb = readlane_f(qx, 16 * ca + 16), cyb = readlane_f(qy, 16 * ca + 16), czb = readlane_f(qz, 16 * ca + 16);
            const float cx = half ? cxb : cxa, cy = half ? cyb : cya, cz = half ? czb : cza;
            const float d = sad::d2f(pa.x, pa.y, pa.z, cx, cy, cz);
            unsigned acc = 0u;
#pragma unroll
            for (int r = 0; r < NR; ++r) acc |= d < r2[r] ? 1u << r : 0u;
            unsigned key = hi < (half ? Tb : Ta) ? ((unsigned)__float_as_int(pa.w) << 4) | acc : 0xFFFFFFF0u;
            key = merge_upto32<32>(key);        // every exchange stays inside a group of 32 lanes: two independent sorts
            const int jidx = (int)(key >> 4);
            const int m = m0 + ca;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int S = prm.nsample[r];
                int32_t *out = prm.idx[r] + ((size_t)b * M + m) * S;      // (the second centroid's row follows: + S)
                const bool ok = (key >> r) & 1u;
                const unsigned long long bal = __ballot(ok);
                const unsigned blo = (unsigned)bal, bhi = (unsigned)(bal >> 32);
                const int na = __builtin_popcount(blo), nb = __builtin_popcount(bhi);
                const unsigned below = __builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                const unsigned slot = below - (half ? (unsigned)na : 0u);
                const unsigned rowoff = half ? (unsigned)S : 0u;
                if (ok && slot < (unsigned)S) out[rowoff + slot] = jidx;
                // SPEC.md §3 padding: the remaining slots repeat the first (lowest) accepted index; none accepted: zeros
                int fa = 0, fb = 0;
                if (blo) fa = __builtin_amdgcn_readlane(jidx, __builtin_ctz(blo));
                if (bhi) fb = __builtin_amdgcn_readlane(jidx, 32 + __builtin_ctz(bhi));
                const int total = half ? nb : na, first = half ? fb : fa;
                const int p1 = total + hi, p2 = p1 + 32;        // (nsample <= 64: two stores per lane cover [total, S))
                if (p1 < S) out[rowoff + (unsigned)p1] = first;
                if (p2 < S) out[rowoff + (unsigned)p2] = first;
                if (prm.cnt[r] && hi == 0) prm.cnt[r][(size_t)b * M + m + half] = total < S ? total : S;
            }
            continue;
        }
#pragma unroll 1
        for (int h2 = 0; h2 < 2; ++h2) {
            const int c = 2 * p + h2;
            const int m = m0 + c;
            if (m >= M) break;                  // wave-uniform
            const float4 prc = h2 ? pb : pa;
            const int T = h2 ? (p ? T3 : T1) : (p ? T2 : T0);
            const float cx = readlane_f(qx, 16 * c), cy = readlane_f(qy, 16 * c), cz = readlane_f(qz, 16 * c);
            auto make_key = [&](const float4 pr, bool valid) -> unsigned {
                const float d = sad::d2f(pr.x, pr.y, pr.z, cx, cy, cz);
                unsigned acc = 0u;
#pragma unroll
                for (int r = 0; r < NR; ++r) acc |= d < r2[r] ? 1u << r : 0u;
                return valid ? ((unsigned)__float_as_int(pr.w) << 4) | acc : 0xFFFFFFF0u;     // (empty lane: sorts last, accepted by no radius)
            };
            if (T > 64 && T <= 128 && !prm.no_sort) {
                // ---- sort path, two candidates per lane (5 % of the centroids of a KITTI-shaped scene; more than 128: 0.05 %) ----
                const bool valid1 = 64 + lane < T;
                const float4 pr1 = rec[valid1 ? record_of(c, 64 + lane) : 0];
                unsigned k0 = make_key(prc, true), k1 = make_key(pr1, valid1);
                bitonic_sort128(k0, k1, mir63, x32);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int S = prm.nsample[r];                       // (<= 64: one padding store covers the row)
                    int32_t *out = prm.idx[r] + ((size_t)b * M + m) * S;
                    const bool ok0 = (k0 >> r) & 1u, ok1 = (k1 >> r) & 1u;
                    const unsigned long long b0 = __ballot(ok0), b1 = __ballot(ok1);
                    const int n0 = __builtin_popcountll(b0), total = n0 + __builtin_popcountll(b1);
                    const unsigned s0 = __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
                    const unsigned s1 = n0 + __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
                    if (ok0 && s0 < (unsigned)S) out[s0] = (int)(k0 >> 4);
                    if (ok1 && s1 < (unsigned)S) out[s1] = (int)(k1 >> 4);
                    int first = 0;
                    if (b0) first = __builtin_amdgcn_readlane((int)(k0 >> 4), __builtin_ctzll(b0));
                    else if (b1) first = __builtin_amdgcn_readlane((int)(k1 >> 4), __builtin_ctzll(b1));
                    if (lane >= total && lane < S) out[(unsigned)lane] = first;
                    if (prm.cnt[r] && lane == 0) prm.cnt[r][(size_t)b * M + m] = total < S ? total : S;
                }
                continue;
            }
            if (T <= 64 && !prm.no_sort) {
                // ---- sort path, one candidate per lane ----
                unsigned key = make_key(prc, lane < T);
                key = bitonic_sort64n(key, mir63, T);       // (the T candidates sit in lanes 0 .. T-1)
                const int jidx = (int)(key >> 4);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int S = prm.nsample[r];
                    int32_t *out = prm.idx[r] + ((size_t)b * M + m) * S;
                    const bool ok = (key >> r) & 1u;
                    const unsigned long long bal = __ballot(ok);
                    const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    const int total = __builtin_popcountll(bal);
                    if (ok && slot < (unsigned)S) out[slot] = jidx;
                    int first = 0;
                    if (bal) first = __builtin_amdgcn_readlane(jidx, __builtin_ctzll(bal));
                    if (lane >= total && lane < S) out[(unsigned)lane] = first;
                    if (prm.cnt[r] && lane == 0) prm.cnt[r][(size_t)b * M + m] = total < S ? total : S;
                }
                continue;
            }
            // ---- bitmap path (more than 128 candidates, or bq_variant = 1) ----
            if (lane == 0)
                while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(4);
            __threadfence_block();
            if (!set_clean) {                       // (wave-uniform) first use by this wave: whatever LDS held before, or what a
                for (int w = lane; w < NR * (NWP + 64); w += 64) bm[w] = 0u;     // sibling left clean, becomes zero
                __threadfence_block();
                set_clean = true;
            }
            for (int i0 = 0; i0 < T; i0 += 64) {
                const int i = i0 + lane;
                const bool valid = i < T;
                float4 pr = prc;
                if (i0 > 0) pr = rec[valid ? record_of(c, i) : 0];
                const float d = sad::d2f(pr.x, pr.y, pr.z, cx, cy, cz);
                const unsigned j = (unsigned)__float_as_int(pr.w);
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (valid && d < r2[r]) {
                        const unsigned w = j >> 5;
                        atomicOr(&bm[r * NWP + w], 1u << (j & 31));
                        atomicOr(&dm[r * 64 + (w >> wshift)], 1u << (w & (WPL - 1)));
                    }
            }
            // scan each bitmap in ascending bit order; a lane owns WPL consecutive words
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int S = prm.nsample[r];
                int32_t *out = prm.idx[r] + ((size_t)b * M + m) * S;
                unsigned *bw = bm + r * NWP + lane * WPL;
                const unsigned dirty = dm[r * 64 + lane];
                dm[r * 64 + lane] = 0u;
                int cnt = 0;
                for (unsigned dd = dirty; dd; dd &= dd - 1) cnt += __builtin_popcount(bw[__builtin_ctz(dd)]);
                const int inclb = wave_incl_scan(cnt);              // DPP: no LDS round trips on this dependent chain
                int slot = inclb - cnt;                             // exclusive prefix
                const int total = __builtin_amdgcn_readlane(inclb, 63);
                int myfirst = 0;
                for (unsigned dd = dirty; dd; dd &= dd - 1) {
                    const int k = __builtin_ctz(dd);
                    unsigned wd = bw[k];
                    bw[k] = 0u;                                     // leave the bitmap clean
                    const int wbase = (lane * WPL + k) << 5;
                    if (slot == 0 && !myfirst) myfirst = wbase + __builtin_ctz(wd);
                    while (wd && slot < S) {
                        const int bit = __builtin_ctz(wd);
                        wd &= wd - 1;
                        out[slot++] = wbase + bit;
                    }
                    slot += __builtin_popcount(wd);                 // bits beyond nsample
                }
                // SPEC.md §3 padding: the remaining slots repeat the first (lowest) accepted index
                const unsigned long long has = __ballot(cnt != 0);
                int first = 0;
                if (has) first = __builtin_amdgcn_readlane(myfirst, __builtin_ctzll(has));
                if (lane >= total && lane < S) out[(unsigned)lane] = first;
                if (prm.cnt[r] && lane == 0) prm.cnt[r][(size_t)b * M + m] = total < S ? total : S;
            }
            __threadfence_block();                  // the bitmaps are clean again before the next wave takes them
            if (lane == 0) atomicExch(lock, 0u);
        }
    }
}

template <int NR>
void launch_query(const float *new_xyz, const char *ws, const GQParams &prm, int B, int N, int M,
                  hipStream_t st) {
    const int NW = (N + 31) >> 5;
    int WPL = 1;
    while (64 * WPL < NW) WPL <<= 1;
    const size_t set_bytes = sizeof(unsigned) * ((size_t)NR * (WPL * 64 + 64) + 1);
    // (eight workgroups per CU keep their 160 KB / 8 each; bq_variant = 2: one set, the round-3 layout, for A/B runs)
    const int nsets = (2 * set_bytes <= 20 * 1024 && sad::get_option(sad::OPT_BQ_VARIANT) != 2) ? 2 : 1;
    const size_t lds = set_bytes * nsets + sizeof(int) * GQ_WAVES * 4 * 64;      // + the packed path's candidate tables
    const int nbx = (M + GQ_WAVES * GQ_CPW - 1) / (GQ_WAVES * GQ_CPW);
    const long long nwg = 8LL * ((B + 7) / 8) * nbx;
    hipLaunchKernelGGL((grid_query_kernel<NR>), dim3((unsigned)nwg), dim3(GQ_WAVES * 64), lds, st, new_xyz, ws, prm, N, M, B, nbx, nsets);
}

}  // namespace

SAD_API size_t sad_ball_query_grid_workspace_bytes(int B, int N) {
    if (B < 1 || N < 1) return 0;
    return (size_t)B * scene_ws_bytes(N);
}

SAD_API int sad_ball_query_grid_f32(const float *xyz, const float *new_xyz, int n_radii,
                                    const float *radii, const int *nsamples, int32_t *const *idx,
                                    int32_t *const *cnt, int B, int N, int M, void *workspace,
                                    sad_stream_t stream) {
    SAD_REQUIRE(xyz && new_xyz && radii && nsamples && idx && workspace, "sad_ball_query_grid_f32: NULL pointer");
    SAD_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && M >= 1, "sad_ball_query_grid_f32: need B,N,M >= 1");
    SAD_REQUIRE(n_radii >= 1 && n_radii <= SAD_MAX_RADII, "sad_ball_query_grid_f32: n_radii=%d not in 1..%d", n_radii, SAD_MAX_RADII);
    SAD_REQUIRE(N <= 65536, "sad_ball_query_grid_f32: N=%d > 65536 (bitmap would not fit LDS)", N);
    SAD_REQUIRE((uintptr_t)workspace % 16 == 0, "sad_ball_query_grid_f32: workspace must be 16-byte aligned");
    GQParams prm{};
    float rmax = 0.f;
    for (int r = 0; r < n_radii; ++r) {
        SAD_REQUIRE(nsamples[r] >= 1 && nsamples[r] <= 64 && idx[r], "sad_ball_query_grid_f32: bad nsample/idx");
        SAD_REQUIRE(radii[r] > 0.f, "sad_ball_query_grid_f32: radius must be > 0");
        prm.radii[r] = radii[r];
        prm.nsample[r] = nsamples[r];
        prm.idx[r] = idx[r];
        prm.cnt[r] = cnt ? cnt[r] : nullptr;
        rmax = radii[r] > rmax ? radii[r] : rmax;
    }
    prm.no_sort = sad::get_option(sad::OPT_BQ_VARIANT) == 1;
    hipStream_t st = (hipStream_t)stream;
    static std::atomic<uint64_t> attr_done0{0}, attr_done4{0}, attr_done16{0};
    const size_t blds = sizeof(int) * (GRID_MAXC + 64);
    if (N <= 4 * BUILD_T) {
        sad::lds_attr_once(attr_done4, reinterpret_cast<const void *>(&grid_build_kernel<4>), 144 * 1024);
        hipLaunchKernelGGL(grid_build_kernel<4>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
    } else if (N <= 16 * BUILD_T) {
        sad::lds_attr_once(attr_done16, reinterpret_cast<const void *>(&grid_build_kernel<16>), 144 * 1024);
        hipLaunchKernelGGL(grid_build_kernel<16>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
    } else {
        sad::lds_attr_once(attr_done0, reinterpret_cast<const void *>(&grid_build_kernel<0>), 144 * 1024);
        hipLaunchKernelGGL(grid_build_kernel<0>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
    }
#ifdef SAD_GRID_BUILD_TWICE  // measurement build: the grid build launched twice (idempotent: same tables) — what its time costs the pipelined step
    if (N <= 4 * BUILD_T) hipLaunchKernelGGL(grid_build_kernel<4>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
    else if (N <= 16 * BUILD_T) hipLaunchKernelGGL(grid_build_kernel<16>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
    else hipLaunchKernelGGL(grid_build_kernel<0>, dim3(B), dim3(BUILD_T), blds, st, xyz, N, rmax * 1.001f, (char *)workspace);
#endif
    if (int e = sad::check_launch("sad_ball_query_grid_f32 (build)")) return e;
    const char *ws = (const char *)workspace;
    switch (n_radii) {
        case 1: launch_query<1>(new_xyz, ws, prm, B, N, M, st); break;
        case 2: launch_query<2>(new_xyz, ws, prm, B, N, M, st); break;
        case 3: launch_query<3>(new_xyz, ws, prm, B, N, M, st); break;
        default: launch_query<4>(new_xyz, ws, prm, B, N, M, st); break;
    }
    return sad::check_launch("sad_ball_query_grid_f32 (query)");
}
