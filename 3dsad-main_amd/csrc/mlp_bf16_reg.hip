// Register-resident fused group -> shared-MLP -> max-pool in bfloat16 on the gfx950 matrix cores: "geometry 2" of
// sad_mlp_chain_bf16 (SPEC.md §14, BASELINE.json configs[4]).  No reference source exists
// (/root/reference/README.md:1-2 is the whole upstream repository).
//
// mlp_bf16.hip (round 1) keeps a tile's activations in LDS, streams every weight fragment L2 -> VGPR per wave and
// max-pools with one atomic per (row run, channel): at v_mfma_f32_32x32x16_bf16 rates (32 cycles per 32x32x16
// product) it spent ~50 us per 128-row tile of which ~3 us were MFMAs.  Here ONE WAVE carries a tile of 32 packed rows
// through a 3-layer chain and the activations never leave its registers:
//   * a hidden layer runs as D[cout, row] = W . X^T: the row stays on the lane, the 32 output channels of a tile land
//     in the 16 accumulator registers, and ReLU + v_cvt_pk_bf16 turn registers 8s .. 8s+7 into the operand of k-step s
//     of the next layer IN PLACE (no lane movement, no LDS): element j of lane half h is channel 16s + 8(j>>2) + 4h +
//     (j&3) of the tile, so the next layer's weight fragments are stored with their k in that order (pack time);
//   * the last layer runs as D[row, cout] = X . W^T: the channel is on the lane and a lane half holds 16 rows in its 16
//     registers.  Lane r carries packed row pi(r) = 16*((r>>2)&1) + 4*(r>>3) + (r&3) of the tile from the gather on,
//     which makes those 16 rows CONSECUTIVE rows of the packed order (rows 16h .. 16h+15 in register order): max-pooling
//     the rows of a group is a running max across registers with wave-half-uniform group boundaries, and a group's result
//     leaves as one 128-byte store per lane half (an atomic max only for a group that continues in another half-tile);
//   * the weight fragments of a tile form one fixed stream; the waves of a workgroup walk their tiles in lockstep and
//     share the stream through a three-slot LDS ring of 8-fragment stages (one global load per wave and NW fragments, prefetched two stages
//     ahead, never drained: the last stages of an item fetch the first of the next);
//   * the layer-0 operand is gathered straight from the bf16 feature rows in fragment form (16-byte loads), the relative
//     coordinates are formed in binary32 and rounded once (SPEC.md §14).
// Accumulation is binary32 inside the matrix core; SPEC §14 leaves its order free: parity is a stated tolerance.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
using sad::BfRegChain;
using sad::BfRegMulti;

constexpr int WHOLE_BIT = 1 << 30;
using sad::CONT_BIT;            // split pooling (common.h): the row's group began in an earlier tile
using sad::GID_MASK;
#ifndef SAD_BR_RS
#define SAD_BR_RS 8      // (16: sa1 82 vs 72 us, sa2 47 vs 41, cluster 112 vs 108, sa3 112 vs 115 — the smaller ring / image lets more workgroups share a CU)
#endif
#ifndef SAD_BR_PFD
#define SAD_BR_PFD 4
#endif
#ifndef SAD_BR_DMA
#define SAD_BR_DMA 1     // the weight ring is filled by LDS-DMA (global_load_lds_dwordx4), three stages ahead; 0 = through registers, two ahead
#endif
constexpr int RS = SAD_BR_RS;        // fragments (1 KB each) per ring stage
constexpr int RNS = SAD_BR_DMA ? 4 : 3;      // ring slots: being read / complete / (DMA: in flight) / being written
constexpr int RING_F4 = RNS * RS * 64;
constexpr int PFD_DEFAULT = SAD_BR_PFD;   // ring reads run this many fragments ahead of the MFMA that consumes them
#ifndef SAD_BR_F2
#define SAD_BR_F2 2      // (with the LDS-DMA ring: two workgroups of 256 registers, no spills — a spill reload is a vector-memory load whose wait, a
#endif                   // vmcnt(0), drains the DMA issued for three stages ahead; 3 x 168 registers with 12 spilled: +0.3 % / +0.5 % on the KITTI / nuScenes-shaped step)
#ifndef SAD_BR_PFD2
#define SAD_BR_PFD2 2
#endif
// SA3 family, round 4 (weights through registers): three workgroups per CU (168 registers) with reads two fragments ahead — 107 us against 121
// with two workgroups of 207 registers and four ahead on the KITTI-shaped batch (three workgroups at four ahead: 113, 19 registers spilled
// instead of 10).  Round 5 (LDS-DMA ring): two workgroups again, see SAD_BR_F2
__host__ __device__ constexpr int pfd_of(int family) { return family == 2 ? SAD_BR_PFD2 : PFD_DEFAULT; }
// Staged pooled output.  A wave owns 4 KB of LDS: STAGE_F floats = slots x CB channels, slot = ordinal of a group inside
// the tile, CB = 128 / 64 / 32 channels per block for tiles with <= 8 / 16 / 32 groups.  EVERY row of an output tile
// max-combines into its group's slot with one LDS atomic per register (ds_max_u32 on the bit patterns: values are >= +0
// after the ReLU) — no running max, no per-row control flow — and after the last output tile of a block each group
// leaves as one row of CB channels: coalesced 16-byte stores (64 lanes cover 256 / CB groups), an atomic max only for the
// first / last group of a tile when it continues in another tile.  The first version walked the 16 registers with a
// running max and stored 128 bytes per (group end, lane half, output tile): 16 exec-masked branches per output tile cost
// ~2 700 cycles against 512 of MFMAs (tools/probe/bf16_stamps.py), and half-tile atomics ran at the memory pipeline's
// instruction rate.
constexpr int STAGE_F = 1024;        // floats of pooled-output staging per wave
__host__ __device__ constexpr int stage_cb(int no2) { return no2 * 32 < 128 ? no2 * 32 : 128; }

// One wave-wide 16-byte LDS-DMA (1 KB lands at lds_dst + 16 * lane; cdna_hip_programming.md: M0 carries the LDS address and is the compiler's,
// so it is saved and restored in the statement that uses it).  The compiler keeps no count of these loads: they are retired by wait_vm.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N)); }

__device__ __forceinline__ void atomic_max_pos(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned *>(addr), __builtin_bit_cast(unsigned, v));
}

__device__ __forceinline__ bf16x8 as_bf(const float4 v) { return __builtin_bit_cast(bf16x8, v); }

// max(x, lo) for lo >= 0 as ONE v_med3_f32 (x, lo, +inf).  A builtin, not inline asm: the operand is an MFMA result and
// hipcc pads the MFMA -> VALU wait states only for instructions it knows (a v_max in an asm statement read stale
// accumulators on the narrow chains); the `x > lo ? x : lo` form costs a canonicalising v_max in front of the compare.
__device__ __forceinline__ float max_pos(float x, float lo) { return __builtin_amdgcn_fmed3f(x, lo, __builtin_inff()); }
__device__ __forceinline__ float relu1(float x) { return max_pos(x, 0.f); }
__device__ __forceinline__ bf16x8 pack8(const f32x16 &t, int s) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = t[8 * s + j];
        v[j] = (__bf16)relu1(f);
    }
    return v;
}

#ifdef SAD_BR_STAMPS   // measurement build only (tools/probe/bf16_stamps.py): s_memtime sums per phase over a wave's tiles
__device__ unsigned long long g_brst[1024 * 16];
#define SAD_BSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SAD_BACC(i, t1, t0) do { if (blockIdx.x < 256 && lane == 0) g_brst[(blockIdx.x * 4 + wave) * 16 + (i)] += (t1) - (t0); } while (0)
#else
#define SAD_BSTAMP(var)
#define SAD_BACC(i, t1, t0)
#endif

struct Ring {
    float4 *ring;       // [RNS][RS][64 lanes]
    int slot;           // slot of the stage being consumed (wave-uniform)
};

// Pooling bookkeeping of one tile (wave-uniform unless noted)
struct PoolInfo {
    int kind;           // 1: each half of the tile lies inside one group (plain max chain), else general
    int soff[16];       // (per lane) byte offset of the slot of row 16h + i inside the wave's staging area, + 4 * (lane & 31)
    int cbs;            // log2 of the block width CB in channels (5, 6, 7)
    int ngroups;        // groups with live rows in this tile (0: a tile past the end, nothing is stored)
    int g_first;        // group of the tile's first row (groups are consecutive in the packed order)
    bool whole_first, whole_last;   // the first / last group of the tile has no rows in another tile
    bool cont_first;    // split pooling: the tile's first group began in an earlier tile (its rows here pool into the tile's continuation row)
};

// Row-map entries of a wave's NEXT tile, fetched while the current one runs: the gather is a chain of dependent round
// trips (row map -> feature row / coordinates), and with two waves per SIMD nothing hides the first of them
struct NextRows {
    int src, gvq, gvn;  // row_src / row_gid of this lane's row pi(r), row_gid of tile row r (natural order)
};
__device__ __forceinline__ NextRows fetch_rows(const BfRegChain &c, int tile, int total, int lane) {
    const int r = lane & 31;
    const int rho = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    int q = tile * 32 + rho, qt = tile * 32 + r;
    q = q < total ? q : total - 1;                  // rows past the end (and whole tiles past it) repeat the last row and store nothing
    qt = qt < total ? qt : total - 1;
    NextRows n;
    n.src = c.row_src[q];
    n.gvq = c.row_gid[q];
    n.gvn = c.row_gid[qt];
    return n;
}

// STATICW: the chain's whole fragment stream sits in LDS for the lifetime of the workgroup (the narrow first-stage chains:
// 3 - 22 fragments) — no ring, no barriers, the waves of a workgroup are independent; `rs.ring` then points at that image.
// SPLIT: split pooling (bf16 rows + continuation rows, plain stores: common.h BfRegChain) — its own instantiation, so that the
// f32 / atomic-max form keeps its registers (with a run-time flag family 2 spilled 33 registers instead of 13)
template <int KS0, int NO0, int NO1, int NO2, bool VEC0, int NW, bool STATICW, int PFD, bool SPLIT>
__device__ __forceinline__ void br_tile(const BfRegChain &c, const int tile, const float *__restrict__ sbias, const int lane, const int wave,
                                        Ring &rs, const float4 *__restrict__ sbase, const float4 *__restrict__ nbase, float *stage,
                                        const int total, NextRows &rows, const BfRegChain &nc, const int ntile, const int ntotal) {
    constexpr int KS1 = 2 * NO0, KS2 = 2 * NO1;
    constexpr int CB = stage_cb(NO2);                     // widest staged block of this chain (channels)
    constexpr int P0 = NO0 * KS0, P1 = NO1 * KS1, P2 = NO2 * KS2, P = P0 + P1 + P2;
    constexpr int NSTG = (P + RS - 1) / RS;
    constexpr int FPW = RS / NW;
    constexpr bool ROLL2 = (KS2 % RS == 0) && NO2 > 4;      // layer-2 tiles span whole stages: loop over them stays rolled
    const int r = lane & 31, h = lane >> 5;
    SAD_BSTAMP(ts0);
    // ---- rows: lane r carries packed row pi(r) of the tile (its row-map entries were fetched during the previous tile) ----
    const int src = rows.src;
    const int grp = rows.gvq & GID_MASK;
    const int gv_nat = rows.gvn;
    float rel[3];
    {
        const float *pq = c.xyz + (long long)src * 3;
        const float *pc = c.new_xyz + (long long)grp * 3;
        rel[0] = pq[0] - pc[0]; rel[1] = pq[1] - pc[1]; rel[2] = pq[2] - pc[2];
    }
    // ---- layer-0 operand: k-step ks, half h = 16-byte chunk 2 ks + h of [feat(C) | dx dy dz | 0 ...] ----------
    bf16x8 x0[KS0];
    {
        const int C = c.C;
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            bf16x8 v = {};
            const int ch = 2 * ks + h;
            if constexpr (VEC0) {                    // bf16 rows, C % 8 == 0: whole chunks
                const __bf16 *row = reinterpret_cast<const __bf16 *>(c.feat) + (long long)src * c.ld_feat;
                if (ch * 8 < C) v = *reinterpret_cast<const bf16x8 *>(row + ch * 8);
                else if (ch * 8 == C) { v[0] = (__bf16)rel[0]; v[1] = (__bf16)rel[1]; v[2] = (__bf16)rel[2]; }
            } else {                                 // any layout (narrow first stage): element by element
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = ch * 8 + j;
                    float f = 0.f;
                    if (k < C) {
                        f = c.feat_bf16 ? (float)reinterpret_cast<const __bf16 *>(c.feat)[(long long)src * c.ld_feat + k]
                                        : reinterpret_cast<const float *>(c.feat)[(long long)src * c.ld_feat + k];
                    } else if (k < C + 3) {
                        f = k == C ? rel[0] : (k == C + 1 ? rel[1] : rel[2]);
                    }
                    v[j] = (__bf16)f;
                }
            }
            x0[ks] = v;
        }
    }
    rows = fetch_rows(nc, ntile, ntotal, lane);     // (consumed by this wave's next tile)
    const float *sb0 = sbias, *sb1 = sbias + NO0 * 32, *sb2 = sb1 + NO1 * 32;
    // (kernel-argument fields used inside the loops: held in vector registers — left to the compiler they are re-read from
    // the argument segment at every use, and every scalar load drains the queue of LDS reads with an lgkmcnt(0))
    int ldo = c.ld_out, cout_last = c.cout_last, col_off = c.col_off;
    asm volatile("" : "+v"(ldo), "+v"(cout_last), "+v"(col_off));
    float *const c_out = c.out;
    __bf16 *const contp = SPLIT ? reinterpret_cast<__bf16 *>(c.cont) : nullptr;     // (wave-uniform, used once per staged block: left in scalar registers)
    const bool vec_out = (c.ld_out % 4 == 0) && (c.col_off % 4 == 0) && ((reinterpret_cast<uintptr_t>(c.out) & 15) == 0);

    // ---- the ring ---------------------------------------------------------------------------------------------
    float4 *const ring = rs.ring;
    int slot = rs.slot;
    int srel = 0;
    static_assert(FPW == 2 || FPW == 4, "two or four fragments per wave and stage");   // (FPW == 2: T2, T3 stay unused)
    float4 T0, T1, T2, T3;                          // this wave's fragments of stage srel + 2, in flight (named: an array
    T2 = T3 = make_float4(0.f, 0.f, 0.f, 0.f);      // captured by the lambdas below may end up in scratch)
    const unsigned ulane = (unsigned)lane;
    auto stage_begin = [&]() {
        if constexpr (STATICW) return;
#if SAD_BR_DMA
        // this wave's fragments of stage srel + 3 (the last three stages of an item: the first three of the next) straight into the slot
        // that was read in the previous stage — every wave has passed that stage's barrier behind its reads
        const int s3 = srel + 3;
        const float4 *sp = s3 < NSTG ? sbase + (size_t)(s3 * RS + FPW * wave) * 64 : nbase + (size_t)((s3 - NSTG) * RS + FPW * wave) * 64;
        const int ws = slot + 3 >= RNS ? slot + 3 - RNS : slot + 3;
        const unsigned dst = (unsigned)(size_t)ring + (unsigned)(ws * RS + FPW * wave) * 1024u;
#pragma unroll
        for (int e = 0; e < FPW; ++e) glds16(sp + e * 64 + ulane, dst + e * 1024u);
#else
        const int s2 = srel + 2;
        const float4 *sp = s2 < NSTG ? sbase + (size_t)(s2 * RS + FPW * wave) * 64 : nbase + (size_t)((s2 - NSTG) * RS + FPW * wave) * 64;
        T0 = sp[ulane];
        T1 = (sp + 64)[ulane];
        if constexpr (FPW == 4) {
            T2 = (sp + 128)[ulane];
            T3 = (sp + 192)[ulane];
        }
#endif
    };
    auto stage_end = [&]() {
        if constexpr (STATICW) return;
#if SAD_BR_DMA
        // retire the DMA issued one stage ago (stage srel + 2: the next stage reads ahead into it), leave this stage's in flight; anything
        // else this wave has in the queue behind it (gather loads, pooled-row stores) only makes the wait longer, never shorter
        wait_vm<FPW>();
        __syncthreads();                            // (lgkmcnt(0) + s_barrier: the compiler knows of no DMA)
        slot = slot + 1 == RNS ? 0 : slot + 1;
        ++srel;
#else
        const int ws = slot + 2 >= RNS ? slot + 2 - RNS : slot + 2;
        float4 *wp = ring + (ws * RS + FPW * wave) * 64 + lane;
        wp[0] = T0;
        wp[64] = T1;
        if constexpr (FPW == 4) {
            wp[128] = T2;
            wp[192] = T3;
        }
        __syncthreads();
        slot = slot + 1 == RNS ? 0 : slot + 1;
        ++srel;
#endif
    };
    // fragment at position pp of this tile's stream, read while position p0 is being consumed (pp >= p0)
    auto frag_at = [&](int pp, int p0) -> float4 {
        if constexpr (STATICW) return ring[pp * 64 + lane];          // (the image is padded to whole stages: pp < P + RS)
        int sl = slot + (pp / RS - p0 / RS);                          // the stage 0 / 1 ahead of the one being consumed
        sl = sl >= RNS ? sl - RNS : sl;
        return ring[(sl * RS + pp % RS) * 64 + lane];
    };
    float4 a[PFD];
#pragma unroll
    for (int u = 0; u < PFD; ++u) a[u] = frag_at(u, 0);
    // one fragment position: the weight operand of position p, the read PFD positions ahead, stage bookkeeping
#define BR_BEGIN(p) do { if ((p) % RS == 0) stage_begin(); } while (0)
#define BR_NEXT(p) do { a[(p) % PFD] = frag_at((p) + PFD, (p)); \
                        __builtin_amdgcn_sched_barrier(0);       /* the reads stay PFD positions ahead of their MFMAs */ \
                        if ((p) % RS == RS - 1) stage_end(); } while (0)

    bf16x8 x1[KS1];
    // ---- layer 0 ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int o = 0; o < NO0; ++o) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4 *>(sb0 + o * 32 + 8 * g + 4 * h);
            acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            const int p = o * KS0 + ks;
            BR_BEGIN(p);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[p % PFD]), x0[ks], acc, 0, 0, 0);
            BR_NEXT(p);
        }
        x1[2 * o] = pack8(acc, 0);
        x1[2 * o + 1] = pack8(acc, 1);
    }
    SAD_BSTAMP(ts1);
    bf16x8 x2[KS2];
    // ---- layer 1 ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int o = 0; o < NO1; ++o) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4 *>(sb1 + o * 32 + 8 * g + 4 * h);
            acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int p = P0 + o * KS1 + ks;
            BR_BEGIN(p);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[p % PFD]), x1[ks], acc, 0, 0, 0);
            BR_NEXT(p);
        }
        x2[2 * o] = pack8(acc, 0);
        x2[2 * o + 1] = pack8(acc, 1);
    }
    // ---- pooling bookkeeping, computed only now: its 16 slot offsets would otherwise be live through layers 0 and 1
    // (lane t < 32 looks at tile row t in natural order; both halves compute the same) ----
    PoolInfo pi;
    {
        const bool live = tile * 32 + r < total;
        const int gv = gv_nat;
        const int g = gv & GID_MASK;
        const int gprev = __shfl_up(g, 1, 64);
        const bool same_prev = r > 0 && gprev == g;                 // (rows past the end repeat the last row: same group)
        const unsigned startm = ~(unsigned)__ballot(same_prev && lane < 32);      // bit t: row t starts a group (bit 0 set)
        const unsigned livem = (unsigned)__ballot(live && lane < 32);
        pi.kind = (~startm & 0xFFFEFFFEu) == 0xFFFEFFFEu ? 1 : 2;   // (rows past the end count as continuing the last group)
        pi.ngroups = __builtin_amdgcn_readfirstlane(__builtin_popcount(startm & livem));
        pi.g_first = __builtin_amdgcn_readlane(g, 0);
        const int nlive = __builtin_popcount(livem);
        pi.whole_first = (__builtin_amdgcn_readlane(gv, 0) & WHOLE_BIT) != 0;
        pi.whole_last = (__builtin_amdgcn_readlane(gv, nlive > 0 ? nlive - 1 : 0) & WHOLE_BIT) != 0;
        pi.cont_first = (__builtin_amdgcn_readlane(gv, 0) & CONT_BIT) != 0;
        constexpr int CBS_MAX = CB == 128 ? 7 : (CB == 64 ? 6 : 5);
        const int want = pi.ngroups <= 8 ? 7 : (pi.ngroups <= 16 ? 6 : 5);
        pi.cbs = want < CBS_MAX ? want : CBS_MAX;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = __builtin_popcount(startm & (0xFFFFFFFFu >> (31 - (16 * h + i)))) - 1;
            pi.soff[i] = (slot << (pi.cbs + 2)) + 4 * r;
        }
    }
    SAD_BSTAMP(ts2);
    // ---- layer 2: D[row, cout] = X . W^T, pooled over the rows of each group -------------------------------------
    auto l2_tile = [&](const int o, const int p0) {   // p0: position of the tile's first fragment (modulo the stage size when rolled)
        SAD_BSTAMP(tk0);
        f32x16 acc;
        {
            const float bv = sb2[o * 32 + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = bv;
        }
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int p = p0 + ks;
            BR_BEGIN(p);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2[ks], as_bf(a[p % PFD]), acc, 0, 0, 0);
            BR_NEXT(p);
        }
#ifdef SAD_BR_STAMPS
        asm volatile("" : "+v"(acc));
        SAD_BSTAMP(tk1);
        SAD_BACC(6, tk1, tk0);
#endif
        // register i of this lane half = row 16h + i of the tile, channel 32 o + r
#ifdef SAD_BR_ABL       // measurement builds (tools/probe/build_variant.sh): 1 no pooling atomics, 2 no flush, 3 neither (wrong results)
        if (SAD_BR_ABL == 3) { asm volatile("" :: "v"(acc)); return; }
#endif
        if (pi.ngroups > 0) {
            const int nobm = (1 << (pi.cbs - 5)) - 1;               // output tiles per block - 1
            const int ob = o & nobm;
            char *sp = reinterpret_cast<char *>(stage) + 128 * ob;
#ifdef SAD_BR_ABL
            if (SAD_BR_ABL == 1) { asm volatile("" :: "v"(acc)); } else
#endif
            if (pi.kind == 1) {
                // each half of the tile lies inside one group: a plain max chain across the registers, one LDS atomic per half
                float m = relu1(acc[0]);
#pragma unroll
                for (int i = 1; i < 16; ++i) m = max_pos(acc[i], m);      // (m >= 0: the ReLU is part of the max)
                atomicMax(reinterpret_cast<unsigned *>(sp + pi.soff[15]), __builtin_bit_cast(unsigned, m));
            } else {
                // every row goes to its group's slot: 16 LDS atomics, no control flow.  (Measured alternatives, all slower on
                // the KITTI-shaped batch: a running max + an atomic only at the rows that end a run, skipped by scalar or by
                // exec-mask branches — 16 branches per output tile cost more than the 16 - ~5 atomics they save.)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = relu1(acc[i]);
                    atomicMax(reinterpret_cast<unsigned *>(sp + pi.soff[i]), __builtin_bit_cast(unsigned, v));      // ds_max_u32
                }
            }
#ifdef SAD_BR_ABL
            if (SAD_BR_ABL == 2) return;
#endif
            if (ob == nobm) {
                // block complete: every group leaves as one row of CB channels, then its slot is zero again
                const int cbs = pi.cbs;
                const int ch0 = (o >> (cbs - 5)) << cbs;
                const int lpg = 1 << (cbs - 2);                     // lanes per group (16 bytes per lane)
                const int ch = ch0 + 4 * (lane & (lpg - 1));
                if constexpr (SPLIT) {
                    // split pooling: every group of the tile leaves as ONE plain 8-byte store per lane of bf16 (rounding is monotone:
                    // the rounded maximum is the maximum of the rounded values) — the rows of a group that began in an earlier tile
                    // into the tile's continuation row, all others into the group's row; the layer that reads the pooled rows takes
                    // the maximum of the two (mlp_bf16_rows.hip).  (host: cout % 8 == 0, ld_out % 8 == 0, col_off % 8 == 0, 16-byte bases)
                    // (eight channels per lane and 16-byte stores — half the passes — measured 0.6 % SLOWER in the pipelined step: two LDS reads and two clears per lane)
                    __bf16 *const outb = reinterpret_cast<__bf16 *>(c_out);
                    for (int s0 = 0; s0 < pi.ngroups; s0 += 64 >> (cbs - 2)) {   // (wave-uniform)
                        const int sidx = s0 + (lane >> (cbs - 2));
                        float4 *src = reinterpret_cast<float4 *>(stage + (s0 << cbs)) + lane;
                        if (sidx < pi.ngroups) {
                            const float4 v = *src;
                            *src = make_float4(0.f, 0.f, 0.f, 0.f);
                            __bf16 *orow = (sidx == 0 && pi.cont_first) ? contp + (long long)tile * cout_last + ch
                                                                         : outb + (long long)(pi.g_first + sidx) * ldo + col_off + ch;
                            bf16x4 pk;
                            pk[0] = (__bf16)v.x; pk[1] = (__bf16)v.y; pk[2] = (__bf16)v.z; pk[3] = (__bf16)v.w;
                            if (ch + 3 < cout_last) *reinterpret_cast<bf16x4 *>(orow) = pk;
                        }
                    }
                    return;
                }
                for (int s0 = 0; s0 < pi.ngroups; s0 += 64 >> (cbs - 2)) {       // (wave-uniform)
                    const int sidx = s0 + (lane >> (cbs - 2));
                    float4 *src = reinterpret_cast<float4 *>(stage + (s0 << cbs)) + lane;
#if defined(SAD_BR_ABL) && SAD_BR_ABL == 4      // (4: plain stores where a group continues in another tile, 5: no global stores at all)
                    const bool whole = true;
#else
                    const bool whole = (sidx > 0 || pi.whole_first) && (sidx < pi.ngroups - 1 || pi.whole_last);
#endif
                    if (sidx < pi.ngroups && whole) {
                        const float4 v = *src;
                        *src = make_float4(0.f, 0.f, 0.f, 0.f);
#if defined(SAD_BR_ABL) && SAD_BR_ABL == 5
                        asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                        continue;
#endif
                        float *orow = c_out + (long long)(pi.g_first + sidx) * ldo + col_off + ch;
                        if (vec_out && ch + 3 < cout_last) {
                            *reinterpret_cast<float4 *>(orow) = v;
                        } else if (ch < cout_last) {
                            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (ch + k < cout_last) orow[k] = e[k];
                        }
                    }
                }
                // the first / last group of the tile when it continues in another tile: an atomic max per channel, one channel
                // per lane — an instruction then covers 64 consecutive floats (four per lane at a 16-byte stride touched every
                // 64-byte line of the row four times: the atomics cost ~10 % of every dispatch that way)
                auto combine = [&](const int sidx) {
                    float *srow = stage + (sidx << cbs);
                    float *orow = c_out + (long long)(pi.g_first + sidx) * ldo + col_off + ch0;
                    for (int cc = lane; cc < (1 << cbs); cc += 64) {
                        const float v = srow[cc];
                        srow[cc] = 0.f;
                        if (ch0 + cc < cout_last) atomic_max_pos(orow + cc, v);
                    }
                };
#if !defined(SAD_BR_ABL) || SAD_BR_ABL != 4
                if (!pi.whole_first) combine(0);
                if (!pi.whole_last && pi.ngroups > 1) combine(pi.ngroups - 1);
#endif
            }
        }
#ifdef SAD_BR_STAMPS
        { SAD_BSTAMP(tk2); SAD_BACC(7, tk2, tk1); }
#endif
    };
    if constexpr (ROLL2) {
        static_assert(RS % PFD == 0 && KS2 % RS == 0, "a rolled tile must start at the same position modulo the stage and the read queue");
#pragma unroll 1
        for (int o = 0; o < NO2; ++o) l2_tile(o, (P0 + P1) % RS);
    } else {
#pragma unroll
        for (int o = 0; o < NO2; ++o) l2_tile(o, P0 + P1 + o * KS2);
    }
    if (!STATICW && P % RS != 0) stage_end();       // the padded last stage
#undef BR_BEGIN
#undef BR_NEXT
    rs.slot = slot;
#ifdef SAD_BR_STAMPS
    {
        SAD_BSTAMP(ts3);
        SAD_BACC(0, ts1, ts0);      // rows, gather, layer 0
        SAD_BACC(1, ts2, ts1);      // layer 1
        SAD_BACC(2, ts3, ts2);      // layer 2 + pooling
        SAD_BACC(3, 1, 0);          // tiles
        SAD_BACC(4, (unsigned long long)P, 0ull);   // fragments
    }
#endif
}

// Shapes (sad::bfreg_shape_id):  KS0, NO0, NO1, NO2
//  0: <=16 -> 16 -> 16 -> 32 (SA1 narrow)    1: <=16 -> 32 -> 32 -> 64 (SA1 wide)     7: <=16 -> 64 -> 64 -> 128 (configs[0])
//  2: 67 -> 64 -> 64 -> 128 (SA2)            3: 67 -> 64 -> 96 -> 128
//  4: 131 -> 128 -> 128 -> 256 (SA3)         5: 131 -> 128 -> 192 -> 256             6: 131 -> 128 -> 256 -> 256
//  8: 259 -> 256 -> 256 -> 512 (cluster)     9: 259 -> 256 -> 512 -> 1024
template <int FAMILY, int NW, bool SPLIT>
__device__ __forceinline__ void run_br(const BfRegChain &c, int shape, int tile, const float *sb, int lane, int wave, Ring &rs,
                                       const float4 *sbase, const float4 *nbase, float *stage, int total, NextRows &rows,
                                       const BfRegChain &nc, int ntile, int ntotal) {
    if constexpr (FAMILY == 0) {
        if (shape == 0) br_tile<1, 1, 1, 1, false, NW, true, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else if (shape == 1) br_tile<1, 1, 1, 2, false, NW, true, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else br_tile<1, 2, 2, 4, false, NW, true, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
    } else if constexpr (FAMILY == 1) {
        if (shape == 2) br_tile<5, 2, 2, 4, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else br_tile<5, 2, 3, 4, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
    } else if constexpr (FAMILY == 2) {
        if (shape == 4) br_tile<9, 4, 4, 8, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else if (shape == 5) br_tile<9, 4, 6, 8, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else br_tile<9, 4, 8, 8, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
    } else {
        if (shape == 8) br_tile<17, 8, 8, 16, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
        else br_tile<17, 8, 16, 32, true, NW, false, pfd_of(FAMILY), SPLIT>(c, tile, sb, lane, wave, rs, sbase, nbase, stage, total, rows, nc, ntile, ntotal);
    }
}

constexpr int BR_NW = 4;

template <int FAMILY, bool SPLIT>
__global__ __launch_bounds__(BR_NW * 64, FAMILY == 0 ? 4 : (FAMILY == 1 ? 3 : (FAMILY == 2 ? SAD_BR_F2 : 2))) void mlp_bf16_reg_kernel(const BfRegMulti mp) {
    constexpr int NW = BR_NW;
    constexpr int FPW = RS / NW;
    constexpr bool STATICW = FAMILY == 0;
    // [ring: RNS stages x RS fragments x 1 KB | family 0: the stream images of all chains][per chain: biases of the three layers]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *ring = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *stage = smem + (STATICW ? mp.static_f4 : RING_F4) * 4 + wave * STAGE_F;    // pooled-output staging of this wave
    for (int i = lane; i < STAGE_F; i += 64) stage[i] = 0.f;
    float *sbias = smem + (STATICW ? mp.static_f4 : RING_F4) * 4 + NW * STAGE_F;
    static_assert(sad::REG_MAX_CHAINS == 3, "chain selection below is written out for three chains");
    int bo = 0, b1 = 0, b2 = 0;
    for (int ci = 0; ci < mp.n; ++ci) {
        const BfRegChain &c = mp.c[ci];
        if (ci == 1) b1 = bo;
        if (ci == 2) b2 = bo;
        for (int l = 0; l < 3; ++l) {
            for (int i = tid; i < c.np[l]; i += NW * 64) sbias[bo + i] = c.bias[l][i];
            bo += c.np[l];
        }
    }
    // items = NW consecutive tiles of one chain, chain 0 first (heaviest); static round-robin deal
    const int tot0 = mp.c[0].rowtab[0], tot1 = mp.n > 1 ? mp.c[1].rowtab[0] : 1, tot2 = mp.n > 2 ? mp.c[2].rowtab[0] : 1;
    const int t0 = ((tot0 + 31) / 32 + NW - 1) / NW;
    const int t1 = mp.n > 1 ? t0 + ((tot1 + 31) / 32 + NW - 1) / NW : t0;
    const int nitems = mp.n > 2 ? t1 + ((tot2 + 31) / 32 + NW - 1) / NW : t1;
    const unsigned ulane = (unsigned)lane;
    auto stream_of = [&](int it) -> const float4 * {
        const int ci = it < t0 ? 0 : (it < t1 ? 1 : 2);
        return reinterpret_cast<const float4 *>(mp.c[ci].stream);
    };
    int item = (int)blockIdx.x;
    int w1 = 0, w2 = 0;                              // family 0: float4 offsets of the images of chains 1 and 2
    if constexpr (STATICW) {
        int wo = 0;
        for (int ci = 0; ci < mp.n; ++ci) {
            if (ci == 1) w1 = wo;
            if (ci == 2) w2 = wo;
            const int n4 = mp.c[ci].stream_frags * 64;
            const float4 *sp = reinterpret_cast<const float4 *>(mp.c[ci].stream);
            for (int i = tid; i < n4; i += NW * 64) ring[wo + i] = sp[i];
            wo += n4;
        }
    } else if (item < nitems) {   // prologue: stages 0 and 1 of the first item (DMA: 0, 1 and 2)
        const float4 *sp = stream_of(item) + (size_t)(FPW * wave) * 64;
#if SAD_BR_DMA
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int e = 0; e < FPW; ++e) glds16(sp + (size_t)(s * RS + e) * 64 + ulane, (unsigned)(size_t)ring + (unsigned)(s * RS + FPW * wave + e) * 1024u);
        wait_vm<0>();
#else
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < FPW; ++e) ring[(s * RS + FPW * wave + e) * 64 + lane] = (sp + (size_t)(s * RS + e) * 64)[ulane];
#endif
    }
    __syncthreads();                                // (also: the biases are in place)
    Ring rs{ring, 0};
#ifdef SAD_BR_STAMPS
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < 256 && lane == 0)
        for (int i = 0; i < 16; ++i) g_brst[(blockIdx.x * 4 + wave) * 16 + i] = 0;
#endif
    auto chain_of = [&](int it) { return __builtin_amdgcn_readfirstlane(it < t0 ? 0 : (it < t1 ? 1 : 2)); };
    auto tile_of = [&](int it, int ci) { return (it - (ci == 0 ? 0 : (ci == 1 ? t0 : t1))) * NW + wave; };
    auto total_of = [&](int ci) { return ci == 0 ? tot0 : (ci == 1 ? tot1 : tot2); };
    NextRows rows{0, 0, 0};
    if (item < nitems) {
        const int ci = chain_of(item);
        rows = fetch_rows(mp.c[ci], tile_of(item, ci), total_of(ci), lane);
    }
    for (; item < nitems; item += (int)gridDim.x) { // (workgroup-uniform)
        const int ci = chain_of(item);
        const int nxt = item + (int)gridDim.x;
        const int nit = nxt < nitems ? nxt : item;   // (no next item: any valid one, its rows are fetched and dropped)
        const int nci = chain_of(nit);
        // the tile's last two stages fetch the first two of the next item's stream, and the tile the row map of the next tile
        if constexpr (STATICW) rs.ring = ring + (ci == 0 ? 0 : (ci == 1 ? w1 : w2));
        run_br<FAMILY, NW, SPLIT>(mp.c[ci], mp.shape[ci], tile_of(item, ci), sbias + (ci == 0 ? 0 : (ci == 1 ? b1 : b2)), lane, wave, rs,
                           stream_of(item), stream_of(nit), stage, total_of(ci), rows, mp.c[nci], tile_of(nit, nci), total_of(nci));
    }
#if SAD_BR_DMA
    if constexpr (!STATICW) wait_vm<0>();             // (no DMA may land in this workgroup's LDS after it has gone)
#endif
#ifdef SAD_BR_STAMPS
    if (blockIdx.x < 256 && lane == 0) {
        g_brst[(blockIdx.x * 4 + wave) * 16 + 12] = __builtin_amdgcn_s_memtime() - tk0;
        g_brst[(blockIdx.x * 4 + wave) * 16 + 13] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
#endif
}

// fragment stream of one chain in consumption order (see the header): layer 0 in natural k order, layers 1 and 2 in the
// order the previous layer's accumulators provide their k; zero fragments pad the last stage
__global__ __launch_bounds__(256) void bfreg_pack_kernel(const float *__restrict__ W0, const float *__restrict__ W1, const float *__restrict__ W2,
                                                         int cin0, int c0, int c1, int c2, int KS0, int NO0, int NO1, int NO2, int xyz_first,
                                                         long long nfrag, __bf16 *__restrict__ dst) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;      // one bf16 per thread
    if (e >= nfrag * 512) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long long p = e >> 9;
    const int r = lane & 31, h = lane >> 5;
    const int KS1 = 2 * NO0, KS2 = 2 * NO1;
    const long long P0 = (long long)NO0 * KS0, P1 = (long long)NO1 * KS1, P2 = (long long)NO2 * KS2;
    float v = 0.f;
    if (p < P0) {
        const int o = (int)(p / KS0), ks = (int)(p % KS0);
        const int co = 32 * o + r, k = 16 * ks + 8 * h + j;          // internal input order [feat(C) | xyz(3)]
        if (co < c0 && k < cin0) {
            int col = k;
            if (xyz_first) col = k < cin0 - 3 ? k + 3 : k - (cin0 - 3);
            v = W0[(size_t)co * cin0 + col];
        }
    } else if (p < P0 + P1 + P2) {
        const bool l1 = p < P0 + P1;
        const long long pp = l1 ? p - P0 : p - P0 - P1;
        const int KS = l1 ? KS1 : KS2, cin = l1 ? c0 : c1, cout = l1 ? c1 : c2;
        const float *W = l1 ? W1 : W2;
        const int o = (int)(pp / KS), ks = (int)(pp % KS);
        const int co = 32 * o + r;
        const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);   // channel behind element j of half h
        if (co < cout && k < cin) v = W[(size_t)co * cin + k];
    }
    dst[e] = (__bf16)v;
}

struct Shape { int cmax, d1, d2, d3, KS0, NO0, NO1, NO2, family; };
const Shape kShapes[10] = {
    {16, 16, 16, 32, 1, 1, 1, 1, 0},      {16, 32, 32, 64, 1, 1, 1, 2, 0},      {67, 64, 64, 128, 5, 2, 2, 4, 1},     {67, 64, 96, 128, 5, 2, 3, 4, 1},
    {131, 128, 128, 256, 9, 4, 4, 8, 2},  {131, 128, 192, 256, 9, 4, 6, 8, 2},  {131, 128, 256, 256, 9, 4, 8, 8, 2},  {16, 64, 64, 128, 1, 2, 2, 4, 0},
    {259, 256, 256, 512, 17, 8, 8, 16, 3}, {259, 256, 512, 1024, 17, 8, 16, 32, 3}};

}  // namespace

namespace sad {

// dims = {C + 3, C1, C2, C3} of a grouped 3-layer chain -> compiled shape, or -1
int bfreg_shape_id(int L, const int *dims) {
    if (L != 3 || !dims) return -1;
    for (int i = 0; i < 10; ++i) {
        const Shape &s = kShapes[i];
        const bool in_ok = s.KS0 == 1 ? (dims[0] >= 3 && dims[0] <= s.cmax) : dims[0] == s.cmax;
        if (in_ok && dims[1] == s.d1 && dims[2] == s.d2 && dims[3] == s.d3) return i;
    }
    return -1;
}
int bfreg_family(int shape) { return shape >= 0 && shape < 10 ? kShapes[shape].family : -1; }

long long bfreg_stream_frags(int shape) {
    if (shape < 0 || shape >= 10) return 0;
    const Shape &s = kShapes[shape];
    const long long P = (long long)s.NO0 * s.KS0 + (long long)s.NO1 * 2 * s.NO0 + (long long)s.NO2 * 2 * s.NO1;
    return (P + RS - 1) / RS * RS;
}

int bfreg_pack(int shape, const int *dims, int first_has_xyz, const float *const *W, void *dst, hipStream_t st) {
    const Shape &s = kShapes[shape];
    const long long nfrag = bfreg_stream_frags(shape);
    const long long n = nfrag * 512;
    hipLaunchKernelGGL(bfreg_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W[0], W[1], W[2], dims[0], dims[1], dims[2], dims[3],
                       s.KS0, s.NO0, s.NO1, s.NO2, first_has_xyz, nfrag, (__bf16 *)dst);
    return check_launch("sad_mlp_pack_bf16 (stream image)");
}

template <int FAMILY, bool SPLIT>
static int launch_bfreg_family(const BfRegMulti &mp, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_bf16_reg_kernel<FAMILY, SPLIT>), 160 * 1024);
    static std::atomic<int> per_cu{0};
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mlp_bf16_reg_kernel<FAMILY, SPLIT>, BR_NW * 64, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 1;
        }
        pc = nb > 4 ? 4 : nb;
        per_cu.store(pc, std::memory_order_relaxed);
    }
    if (get_option(OPT_MLP_DYN_SLOTS) > 0 && get_option(OPT_MLP_DYN_SLOTS) < pc) pc = get_option(OPT_MLP_DYN_SLOTS);   // A/B knob
    const int cus = sad::device_cus();
    long long grid = (long long)cus * pc;
    const long long cap = mp.max_tiles / BR_NW + mp.n;           // never more workgroups than items could exist
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    hipLaunchKernelGGL((mlp_bf16_reg_kernel<FAMILY, SPLIT>), dim3((unsigned)grid), dim3(BR_NW * 64), lds, st, mp);
    return check_launch("sad_mlp_chain_bf16 (register-resident chain)");
}

int launch_bfreg(const BfRegMulti &mp, hipStream_t st) {
    const int fam = bfreg_family(mp.shape[0]);
    BfRegMulti mq = mp;
    mq.static_f4 = 0;
    size_t lds = 0;
    for (int i = 0; i < mq.n; ++i) {
        lds += sizeof(float) * (size_t)(mq.c[i].np[0] + mq.c[i].np[1] + mq.c[i].np[2]);
        if (bfreg_family(mq.shape[i]) != fam) return fail(SAD_EINVAL, "launch_bfreg: chains of different shape families in one dispatch");
        mq.c[i].stream_frags = (int)bfreg_stream_frags(mq.shape[i]);
        mq.static_f4 += mq.c[i].stream_frags * 64;
    }
    lds += sizeof(float4) * (size_t)(fam == 0 ? mq.static_f4 : RING_F4) + sizeof(float) * (size_t)BR_NW * STAGE_F;
    const bool split = mq.c[0].out_bf16 != 0;
    for (int i = 1; i < mq.n; ++i)
        if ((mq.c[i].out_bf16 != 0) != split) return fail(SAD_EINVAL, "launch_bfreg: split-pooled and f32-pooled chains in one dispatch");
    if (split) {
        switch (fam) {
            case 0: return launch_bfreg_family<0, true>(mq, lds, st);
            case 1: return launch_bfreg_family<1, true>(mq, lds, st);
            case 2: return launch_bfreg_family<2, true>(mq, lds, st);
            default: return launch_bfreg_family<3, true>(mq, lds, st);
        }
    }
    switch (fam) {
        case 0: return launch_bfreg_family<0, false>(mq, lds, st);
        case 1: return launch_bfreg_family<1, false>(mq, lds, st);
        case 2: return launch_bfreg_family<2, false>(mq, lds, st);
        default: return launch_bfreg_family<3, false>(mq, lds, st);
    }
}

}  // namespace sad

#ifdef SAD_BR_STAMPS
extern "C" __attribute__((visibility("default"))) int sad_debug_read_br_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_brst), sizeof(unsigned long long) * 1024 * 16);
}
#endif
