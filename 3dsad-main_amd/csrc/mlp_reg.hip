// Register-resident fused group -> shared-MLP -> max-pool for gfx950 (SPEC.md §6): "geometry 2".
// No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
//
// The tiled kernel of mlp_chain.hip shares a row tile between the waves of a workgroup: its
// activations cross LDS after every layer and all waves meet at a barrier per layer, so the matrix
// pipe idles while tiles are staged, pooled and synchronised (SQ_WAIT_ANY ~ 1/3 of the wave cycles on
// the SA chains).  Here ONE WAVE carries a tile of 32 packed rows through the whole chain and the
// activations never leave its registers:
//
//   * every layer is the transposed product Y^T[oc,row] = W[oc,k] X^T[k,row] on
//     v_mfma_f32_32x32x2_f32, so a layer's result has the row on the lane and the output channels in
//     the 16 accumulator registers: lane (j,h) holds channels 8a+4h+q (a,q = 0..3) of row j;
//   * the next layer wants, for k-step pair e of k-group a, channel 8a+2e+h on lane (j,h): two
//     v_permlane32_swap per k-group turn the accumulator registers into exactly those B operands
//     (swap(r0,r1) -> (k0|k1),(k4|k5); swap(r2,r3) -> (k2|k3),(k6|k7)), in place — no LDS, no barrier;
//   * layer 0's operand comes straight from global memory: lane (j,h) loads 16 bytes of row j's
//     gathered feature row (two lanes cover 32 contiguous bytes) and the same two swaps apply;
//   * layers alternate between "transient" and "persistent": a finished 32-channel tile of layer 0
//     (16 registers) is consumed at once as k-range [32o, 32o+32) of ALL output tiles of layer 1,
//     whose accumulators are the persistent register array; layer 1's accumulators then are layer 2's
//     operands, and each 32-channel tile of layer 2 is pooled and stored as soon as it is complete.
//     k still ascends inside every output's chain (tiles are produced in channel order, bias first),
//     so results stay bit-identical to SPEC.md §6's fmaf chain and to the tiled kernel.
//   * weights stream L2 -> VGPR in the A-fragment order of sad_mlp_pack_f32 (one coalesced 16-byte
//     load per lane feeds four MFMAs), a few k-steps ahead in a register ring.
// Waves are independent (no barrier after the bias copy): the matrix pipe of a SIMD is kept busy by
// whichever of its resident waves has operands ready.  Tiles of 32 packed rows (the global row map of
// rowscan_kernel: padding rows are never computed) are handed out from one atomic work counter shared
// by all chains of a dispatch, heaviest chain first.
//
// Shapes are compiled per chain (k-loops are straight-line code over registers): the KITTI / TINY /
// nuScenes SA chains and BASELINE configs[0]; any other chain runs on the tiled kernel.
#include "common.h"

namespace {

#include "reg_common.h"

#ifndef SAD_REG_RING
#define SAD_REG_RING 4
#endif
constexpr int RING = SAD_REG_RING;   // weight fragments in flight per wave
#ifndef SAD_REG_RING2
#define SAD_REG_RING2 8
#endif
// ... and in the last layer, whose tiles end with stores / atomic max merges: vector-memory operations
// retire in order, so a fragment load issued behind an atomic (in flight ~3000 cycles when every CU
// issues them) is not usable before it; with 8 k-groups fetched BEFORE the epilogue of the previous
// tile, the loads that queue behind its atomics are first needed ~4000 cycles later.
constexpr int RING2 = SAD_REG_RING2;
constexpr int WAVES = 4;         // waves per workgroup (independent; they only share the bias copy)


#ifdef SAD_REG_STAMPS    // measurement build only (tools/probe/reg_stamps.py): s_memtime at the phase boundaries of a tile
__device__ unsigned long long g_stamps[64 * 64];
#define SAD_STAMP(i)                                                                                  \
    do {                                                                                              \
        if (blockIdx.x < 16 && (threadIdx.x & 63) == 0 && tile < 16 * 1024)                            \
            g_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63) * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SAD_STAMP(i)
#endif

// A-fragment array of one layer: fragment k (64 lanes x 16 B) at base[k * 64 + lane], base wave-uniform
struct Frags {
    const float4 *__restrict__ base;
    unsigned lane;      // (unsigned: the zero-extended 32-bit offset is what selects the scalar-base load form)
    __device__ __forceinline__ float4 operator()(int k) const { return (base + (size_t)k * 64)[lane]; }   // scalar add, then the lane slot
    __device__ __forceinline__ Frags at(int k) const { return Frags{base + (size_t)k * 64, lane}; }
};

// acc += W[oc tile, 0 : 8*NT] * X, X given as NT*4 operand registers.  af: this lane's first fragment
// (consecutive k-groups are 64 float4 apart).
template <int NT>
__device__ __forceinline__ f32x16 ktile(f32x16 acc, const Frags af, const float *b) {
    constexpr int D = NT < RING ? NT : RING;
    float4 a[D];
#pragma unroll
    for (int u = 0; u < D; ++u) a[u] = af(u);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc = mma4(acc, a[t % D], b + 4 * t);
        if (t + D < NT) a[t % D] = af(t + D);
        __builtin_amdgcn_sched_barrier(0);   // keep the refill right behind the MFMAs that freed the register
    }
    return acc;
}

// The same with the operands in this wave's private LDS image (lane-linear float4 per k-group, written
// by the wave itself): used where the layer-0 input is too wide to stay in registers next to layer 1's
// accumulators.  LDS latency is covered by reading two k-groups ahead.
template <int NT>
__device__ __forceinline__ f32x16 ktile_lds(f32x16 acc, const Frags af, const float4 *lb) {
    constexpr int D = NT < RING ? NT : RING;
    constexpr int BD = NT < 2 ? NT : 2;
    float4 a[D], b[BD];
#pragma unroll
    for (int u = 0; u < D; ++u) a[u] = af(u);
#pragma unroll
    for (int u = 0; u < BD; ++u) b[u] = lb[u * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bb[4] = {b[t % BD].x, b[t % BD].y, b[t % BD].z, b[t % BD].w};
        acc = mma4(acc, a[t % D], bb);
        if (t + D < NT) a[t % D] = af(t + D);
        if (t + BD < NT) b[t % BD] = lb[(t + BD) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// One tile of 32 packed rows through a 3-layer chain.
//   NT0: k-groups (of 8) of the layer-0 input     NO0: 32-channel output tiles of layer 0
//   NG1: k-groups of layer 1 (<= 4*NO0)           NO1: output tiles of layer 1
//   NG2: k-groups of layer 2 (== 4*NO1)           NO2: output tiles of layer 2
//   IN0_LDS: the layer-0 operands live in the wave's LDS image `lds0` (NT0 * 64 float4) instead of registers
//   PIPE: one never-drained weight-fragment ring for the whole tile (wide chains, few waves per SIMD)
template <int NT0, int NO0, int NG1, int NO1, int NG2, int NO2, bool IN0_LDS = false, bool PIPE = false>
__device__ __forceinline__ void reg_tile(const RegChain &c, const int tile, const float *__restrict__ sbias, const int lane,
                                         float4 *lds0, float *lds_pool, const int pool_floats) {
    constexpr int COUT = NO2 * 32;
    static_assert(NG2 <= 4 * NO1 && NG1 <= 4 * NO0, "layer widths must chain");
    constexpr bool FULL1 = NG1 == 4 * NO0;          // layer 1 reads every channel of layer 0's padded output
    const int j = lane & 31, h = lane >> 5;
    SAD_STAMP(0);
    SAD_STAMP(40 + 2 * (tile / (int)(gridDim.x * WAVES) < 10 ? tile / (int)(gridDim.x * WAVES) : 10));
    const int total = c.rowtab[0];
    int q = tile * 32 + j;
    const bool live = q < total;
    if (!live) q = total - 1;                      // rows past the end repeat the last row and store nothing
#ifdef SAD_REG_NOGATHER     // measurement build: every tile reads the first 32 rows (hot in L1)
    q = j;
#endif
    const int src = c.row_src[q];
    const int gv = c.row_gid[q];
    const int grp = gv & (WHOLE_BIT - 1);
    const bool whole = (gv & WHOLE_BIT) != 0;

    // ---- layer-0 operands: [dx dy dz 0 | f0 f1 ...], k-group t = channels 8t .. 8t+7 ----------------
    float in0[IN0_LDS ? 4 : NT0 * 4];
    {
        const float *pf = c.feat + (long long)src * c.ld_feat;
#pragma unroll
        for (int t = 0; t < NT0; ++t) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c.cpr > 0) {                       // 16-byte chunks: lane half h takes chunk 2t - 1 + h of the feature row
                const int ch = 2 * t - 1 + h;
                if (ch >= 0 && ch < c.cpr) v = *reinterpret_cast<const float4 *>(pf + 4 * ch);
            } else if (c.C == 1 && t == 0 && h == 1) {
                v.x = pf[0];
            }
            if (t == 0 && h == 0) {
                const float *pq = c.xyz + (long long)src * 3;
                const float *pc = c.new_xyz + (long long)grp * 3;
                v = make_float4(pq[0] - pc[0], pq[1] - pc[1], pq[2] - pc[2], 0.f);
            }
            if constexpr (IN0_LDS) {
                to_operands(v.x, v.y, v.z, v.w, in0);
                lds0[t * 64 + lane] = make_float4(in0[0], in0[1], in0[2], in0[3]);   // (own lanes only: no barrier needed)
            } else {
                to_operands(v.x, v.y, v.z, v.w, in0 + 4 * t);
            }
        }
    }
    const float *l0 = c.packed + c.off[0], *l1 = c.packed + c.off[1], *l2 = c.packed + c.off[2];
    // wave-uniform fragment bases (SGPR pairs) + the lane's 16-byte slot: loads use the scalar-base addressing
    // form, so stepping through the fragment stream costs scalar adds and no address VGPRs
    const Frags f0{reinterpret_cast<const float4 *>(l0 + NO0 * 32), (unsigned)lane};
    const Frags f1{reinterpret_cast<const float4 *>(l1 + NO1 * 32), (unsigned)lane};
    const Frags f2{reinterpret_cast<const float4 *>(l2 + NO2 * 32), (unsigned)lane};
    const float *sb0 = sbias, *sb1 = sbias + NO0 * 32, *sb2 = sb1 + NO1 * 32;
    const int key = live ? grp + 1 : 0;
    float *orow = c.out + (long long)grp * c.ld_out + c.col_off;
    constexpr int NI = NO1 * 4;
    const int nkey = __shfl_down(key, 1, 64);
    const bool tail = live && (j == 31 || nkey != key);     // last row of its group inside this tile
    Stage sg;
    {
        const int pkey = __shfl_up(key, 1, 64);
        const bool head = live && (j == 0 || pkey != key);
        const unsigned heads = (unsigned)__ballot(head);                 // low 32 bits: lanes of half 0
        sg.tails = (unsigned)__ballot(tail);
        sg.ngroups = __builtin_popcount(heads);
        sg.slot = __builtin_popcount(heads & (0xFFFFFFFFu >> (31 - j))) - 1;
        sg.lds = sg.ngroups * COUT <= pool_floats ? lds_pool : nullptr;   // (wave-uniform)
    }                     // (output tile, k-group) items of layer 1 fed by one tile of layer 0

    f32x16 acc1[NO1];
#pragma unroll
    for (int o1 = 0; o1 < NO1; ++o1) acc1[o1] = bias_tile(sb1 + o1 * 32, h);
    float in2[NG2 * 4];

    if constexpr (PIPE) {
        // ---- one weight-fragment stream for the whole tile, never drained -------------------------------
        // The fragment loads of a tile form a fixed list: per layer-0 tile o its NT0 fragments, then the NI
        // fragments of layer 1 it feeds; then NG2 fragments per layer-2 tile.  Fragment g lives in ring slot
        // g % RING and is fetched RING items ahead of its use ACROSS block boundaries (a drained ring costs
        // one exposed L2 round trip, ~1 us, at the head of every k-loop: 30 % of a 16-group block).
            static_assert(FULL1 && NG2 % RING == 0 && NG2 >= RING, "piped variant: full-width layers");
        constexpr int D = RING;
        constexpr int PER = NT0 + NI, TOT = NO0 * PER, R0 = TOT % D;
        auto item = [&](int g) -> float4 {                 // (g is a compile-time constant after unrolling)
            const int o = g / PER, r = g % PER;
            if (r < NT0) return f0(o * NT0 + r);
            const int i = r - NT0;
            return f1((i >> 2) * NG1 + 4 * o + (i & 3));
        };
        float4 ring[D];
#pragma unroll
        for (int u = 0; u < D; ++u) ring[u] = item(u);
        SAD_STAMP(1);
        f32x16 t;
        float bt[16];
        float4 lb[2];                                      // LDS operand ring (IN0_LDS)
#pragma unroll
        for (int g = 0; g < TOT; ++g) {
            const int o = g / PER, r = g % PER;
            if (r == 0) {
                t = bias_tile(sb0 + o * 32, h);
                if constexpr (IN0_LDS) {
                    lb[0] = lds0[lane];
                    if (NT0 > 1) lb[1] = lds0[64 + lane];
                }
            }
            if (r < NT0) {
                if constexpr (IN0_LDS) {
                    const float bb[4] = {lb[r % 2].x, lb[r % 2].y, lb[r % 2].z, lb[r % 2].w};
                    t = mma4(t, ring[g % D], bb);
                    if (r + 2 < NT0) lb[r % 2] = lds0[(r + 2) * 64 + lane];
                } else {
                    t = mma4(t, ring[g % D], in0 + 4 * r);
                }
            } else {
                const int i = r - NT0;
                acc1[i >> 2] = mma4(acc1[i >> 2], ring[g % D], bt + 4 * (i & 3));
            }
            ring[g % D] = g + D < TOT ? item(g + D) : f2(g + D - TOT);   // (layer-2 fragment i' = g + D - TOT: slot (R0 + i') % D)
            __builtin_amdgcn_sched_barrier(0);             // keep each refill right behind the MFMAs that freed its slot
            if (r == 0 && o < 4) SAD_STAMP(2 + 2 * o);
            if (r == NT0 - 1) {
                SAD_STAMP(3 + 2 * o);
                t = relu16(t);
#pragma unroll
                for (int a = 0; a < 4; ++a) to_operands(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3], bt + 4 * a);
            }
        }
#pragma unroll
        for (int o1 = 0; o1 < NO1; ++o1) {
            const f32x16 u = relu16(acc1[o1]);
#pragma unroll
            for (int a = 0; a < 4; ++a) to_operands(u[4 * a], u[4 * a + 1], u[4 * a + 2], u[4 * a + 3], in2 + 16 * o1 + 4 * a);
        }
        SAD_STAMP(10);
        const PoolMasks pm = pool_masks(key);
        SAD_STAMP(11);
        // layer 2: a ring of RING2 slots; fragment i of a tile lives in slot i % RING2.  The first RING of
        // tile 0 were fetched by the loop above (into ring[(R0 + i) % RING]), the rest are fetched here.
        constexpr int D2 = NG2 % RING2 == 0 ? RING2 : RING;
        float4 ring2[D2];
#pragma unroll
        for (int i = 0; i < D2; ++i) ring2[i] = i < D ? ring[(R0 + i) % D] : f2(i);
#pragma unroll 1
        for (int o = 0; o < NO2; ++o) {
            f32x16 t2 = bias_tile(sb2 + o * 32, h);
            const Frags cur = f2.at(o * NG2);
            const Frags nxt = f2.at((o + 1 < NO2 ? o + 1 : o) * NG2);   // (last tile: harmless re-read)
#pragma unroll
            for (int i = 0; i < NG2; ++i) {
                t2 = mma4(t2, ring2[i % D2], in2 + 4 * i);
                ring2[i % D2] = i + D2 < NG2 ? cur(i + D2) : nxt(i + D2 - NG2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (o < 8) SAD_STAMP(12 + 2 * o);
            if (sg.lds) pool_stage<COUT>(relu16(t2), pm, tail, sg, o, h);
            else pool_store(relu16(t2), pm, tail, whole, orow, o, h, c);
            if (o < 8) SAD_STAMP(13 + 2 * o);
        }
        if (sg.lds) stage_flush<COUT>(sg, grp, whole, lane, c);
        SAD_STAMP(30);
        SAD_STAMP(41 + 2 * (tile / (int)(gridDim.x * WAVES) < 10 ? tile / (int)(gridDim.x * WAVES) : 10));
    } else {
        // ---- simple variant (narrow chains: few MFMAs per block, latency is covered by many waves per SIMD) ----
#pragma unroll 1
        for (int o = 0; o < NO0; ++o) {
            f32x16 t = bias_tile(sb0 + o * 32, h);
            if constexpr (IN0_LDS) t = ktile_lds<NT0>(t, f0.at(o * NT0), lds0 + lane);
            else t = ktile<NT0>(t, f0.at(o * NT0), in0);
            t = relu16(t);
            float bt[16];
#pragma unroll
            for (int a = 0; a < 4; ++a) to_operands(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3], bt + 4 * a);
            // k-groups 4o .. 4o+3 of every output tile of layer 1
            constexpr int D = NI < RING ? NI : RING;
            float4 ar[D];
            const int gmax = FULL1 ? 4 : NG1 - 4 * o;  // k-groups of this tile that layer 1 reads (>= 1)
            auto frag = [&](int i) {                   // item i = (output tile i / 4, k-group a = i % 4)
                const int a = i & 3;
                const int g = 4 * o + (a < gmax ? a : gmax - 1);
                return f1((i >> 2) * NG1 + g);
            };
#pragma unroll
            for (int u = 0; u < D; ++u) ar[u] = frag(u);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if ((i & 3) < gmax) acc1[i >> 2] = mma4(acc1[i >> 2], ar[i % D], bt + 4 * (i & 3));
                if (i + D < NI) ar[i % D] = frag(i + D);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // layer 1's accumulators -> layer 2's operands (in place)
#pragma unroll
        for (int o1 = 0; o1 < NO1; ++o1) {
            const f32x16 t = relu16(acc1[o1]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (4 * o1 + a < NG2) to_operands(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3], in2 + 16 * o1 + 4 * a);
        }
        const PoolMasks pm = pool_masks(key);
#pragma unroll 1
        for (int o = 0; o < NO2; ++o) {
            f32x16 t = bias_tile(sb2 + o * 32, h);
            t = ktile<NG2>(t, f2.at(o * NG2), in2);
            if (sg.lds) pool_stage<COUT>(relu16(t), pm, tail, sg, o, h);
            else pool_store(relu16(t), pm, tail, whole, orow, o, h, c);
        }
        if (sg.lds) stage_flush<COUT>(sg, grp, whole, lane, c);
    }
}

// Shape table (sad::reg_shape_id): NT0, NO0, NG1, NO1, NG2, NO2
//  0: C+3 <= 7 -> 16 -> 16 -> 32    (SA1 narrow branches)        1: <= 7 -> 32 -> 32 -> 64   (SA1 wide branch)
//  2: 67 -> 64 -> 64 -> 128         (SA2)                        3: 67 -> 64 -> 96 -> 128
//  4: 131 -> 128 -> 128 -> 256      (SA3)                        5: 131 -> 128 -> 192 -> 256
//  6: 131 -> 128 -> 256 -> 256                                   7: <= 7 -> 64 -> 64 -> 128  (BASELINE configs[0])
template <int FAMILY>
__device__ __forceinline__ void run_tile(const RegChain &c, int shape, int tile, const float *sb, int lane, float4 *lds0,
                                         float *lp, int pf) {
    if constexpr (FAMILY == 0) {
        if (shape == 0) reg_tile<1, 1, 2, 1, 2, 1>(c, tile, sb, lane, nullptr, lp, pf);
        else reg_tile<1, 1, 4, 1, 4, 2>(c, tile, sb, lane, nullptr, lp, pf);
    } else if constexpr (FAMILY == 1) {
        if (shape == 2) reg_tile<9, 2, 8, 2, 8, 4, true, true>(c, tile, sb, lane, lds0, lp, pf);
        else if (shape == 3) reg_tile<9, 2, 8, 3, 12, 4, true, true>(c, tile, sb, lane, lds0, lp, pf);
        else reg_tile<1, 2, 8, 2, 8, 4, false, true>(c, tile, sb, lane, nullptr, lp, pf);
    } else {
        if (shape == 4) reg_tile<17, 4, 16, 4, 16, 8, true, true>(c, tile, sb, lane, lds0, lp, pf);
        else if (shape == 5) reg_tile<17, 4, 16, 6, 24, 8, true, true>(c, tile, sb, lane, lds0, lp, pf);
        else reg_tile<17, 4, 16, 8, 32, 8, true, true>(c, tile, sb, lane, lds0, lp, pf);
    }
}

// per-wave LDS (in float4): the layer-0 operand image of families 1 / 2 (dead once layer 0 is done, then reused as
// the pooled-output staging buffer); family 0 has no image and gets a staging buffer of its own (16 slots x 64 ch)
__host__ __device__ constexpr int in0_f4(int family) { return family == 2 ? 17 * 64 : (family == 1 ? 9 * 64 : 256); }

template <int FAMILY>
__device__ __forceinline__ void reg_body(const RegMulti &mp) {
    // [families 1, 2: per wave a private image of the layer-0 operands, 9 / 17 k-groups x 64 lanes x 16 B] then
    // per chain the bias blocks of the three layers
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef SAD_REG_STAMPS
    if (blockIdx.x < 16 && (threadIdx.x & 63) == 0) g_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63) * 64 + 63] = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[62] = gridDim.x;
#endif
    constexpr int IN0_F4 = in0_f4(FAMILY);
    float *sbias = smem + WAVES * IN0_F4 * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    float4 *lds0 = reinterpret_cast<float4 *>(smem) + (tid >> 6) * IN0_F4;
    static_assert(sad::REG_MAX_CHAINS == 3, "chain selection below is written out for three chains");
    int bo = 0, b1 = 0, b2 = 0;                   // bias offsets of chains 1 and 2 (scalars: no indexed local arrays)
    for (int ci = 0; ci < mp.n; ++ci) {
        const RegChain &c = mp.c[ci];
        if (ci == 1) b1 = bo;
        if (ci == 2) b2 = bo;
        for (int l = 0; l < 3; ++l) {
            for (int i = tid; i < c.np[l]; i += WAVES * 64) sbias[bo + i] = c.packed[c.off[l] + i];
            bo += c.np[l];
        }
    }
    __syncthreads();          // the only barrier: from here on the waves of the workgroup are independent
    // tiles of all chains form one work list (chain 0 first = heaviest first)
    const int t0 = (mp.c[0].rowtab[0] + 31) / 32;
    const int t1 = mp.n > 1 ? t0 + (mp.c[1].rowtab[0] + 31) / 32 : t0;
    const int nitems = mp.n > 2 ? t1 + (mp.c[2].rowtab[0] + 31) / 32 : t1;
    // Static round-robin hand-out: tiles of one chain cost the same, and a returning atomic on ONE address
    // is served at ~30 ns each chip-wide (measured: a shared work counter capped SA1 at 33 ns per tile).
    const int nwaves = gridDim.x * WAVES;
    for (int item = blockIdx.x * WAVES + (tid >> 6); item < nitems; item += nwaves) {
        const int ci = __builtin_amdgcn_readfirstlane(item < t0 ? 0 : (item < t1 ? 1 : 2));
        const int tile = item - (ci == 0 ? 0 : (ci == 1 ? t0 : t1));
        run_tile<FAMILY>(mp.c[ci], mp.shape[ci], tile, sbias + (ci == 0 ? 0 : (ci == 1 ? b1 : b2)), lane, lds0,
                         reinterpret_cast<float *>(lds0), IN0_F4 * 4);
    }
}

template <int FAMILY>
__global__ __launch_bounds__(WAVES * 64, 2) void mlp_reg_kernel(const RegMulti mp) {   // at least two waves per SIMD: <= 256 registers
    reg_body<FAMILY>(mp);
}
// the narrow SA1 chains are latency / vector-ALU bound: six waves per SIMD instead of five (<= 80 registers)
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(6, 6))) void mlp_reg_kernel_narrow(const RegMulti mp) {
    reg_body<0>(mp);
}
template <int FAMILY>
constexpr auto reg_kernel_of() {
    if constexpr (FAMILY == 0) return &mlp_reg_kernel_narrow;
    else return &mlp_reg_kernel<FAMILY>;
}

template <int FAMILY>
int launch_family(const RegMulti &mp, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(reg_kernel_of<FAMILY>()), 160 * 1024);
    static std::atomic<int> per_cu{0};      // resident workgroups per CU (occupancy query once per process: same on every device of a node)
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reg_kernel_of<FAMILY>(), WAVES * 64, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 2;
        }
        pc = nb > 8 ? 8 : nb;
        per_cu.store(pc, std::memory_order_relaxed);
    }
    const int cus = sad::device_cus();
    if (sad::get_option(sad::OPT_MLP_DYN_SLOTS) > 0 && sad::get_option(sad::OPT_MLP_DYN_SLOTS) < pc) pc = sad::get_option(sad::OPT_MLP_DYN_SLOTS);   // A/B knob
    long long grid = (long long)cus * pc;
    const long long cap = (mp.max_tiles + WAVES - 1) / WAVES;      // never more waves than tiles could exist
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    hipLaunchKernelGGL(reg_kernel_of<FAMILY>(), dim3((unsigned)grid), dim3(WAVES * 64), lds, st, mp);
    return sad::check_launch("sad_mlp_chain_f32 (register-resident chain)");
}

}  // namespace

namespace sad {

int reg_shape_id(int L, const int *kp, const int *np) {
    if (L != 3) return -1;
    struct S { int kp0, np0, kp1, np1, kp2, np2; };
    static const S tab[8] = {
        {8, 32, 16, 32, 16, 32},   {8, 32, 32, 32, 32, 64},     {72, 64, 64, 64, 64, 128},    {72, 64, 64, 96, 96, 128},
        {136, 128, 128, 128, 128, 256}, {136, 128, 128, 192, 192, 256}, {136, 128, 128, 256, 256, 256}, {8, 64, 64, 64, 64, 128}};
    for (int i = 0; i < 8; ++i)
        if (kp[0] == tab[i].kp0 && np[0] == tab[i].np0 && kp[1] == tab[i].kp1 && np[1] == tab[i].np1 && kp[2] == tab[i].kp2 &&
            np[2] == tab[i].np2)
            return i;
    return -1;
}

int reg_family(int shape) { return shape <= 1 ? 0 : (shape <= 3 || shape == 7 ? 1 : 2); }

int launch_reg(const RegMulti &mp, hipStream_t st) {
    size_t lds = 0;
    for (int i = 0; i < mp.n; ++i) lds += sizeof(float) * (size_t)(mp.c[i].np[0] + mp.c[i].np[1] + mp.c[i].np[2]);
    const int fam = reg_family(mp.shape[0]);
    lds += sizeof(float4) * (size_t)WAVES * in0_f4(fam);
    for (int i = 1; i < mp.n; ++i)
        if (reg_family(mp.shape[i]) != fam) return fail(SAD_EINVAL, "launch_reg: chains of different families in one dispatch");
    if (fam == 0) return launch_family<0>(mp, lds, st);
    if (fam == 1) return launch_family<1>(mp, lds, st);
    return launch_family<2>(mp, lds, st);
}

}  // namespace sad

#ifdef SAD_REG_STAMPS
extern "C" __attribute__((visibility("default"))) int sad_debug_read_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 64);
}
#endif
