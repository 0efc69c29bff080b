// Layer-streamed fused group -> shared-MLP -> max-pool for gfx950 (SPEC.md §6): "geometry 3", for chains
// that are WIDE and have FEW rows (the size-adaptive cluster layer: 259 -> 256 -> 512 -> 1024 on ~50 000
// packed rows per step).  No reference source exists (/root/reference/README.md:1-2).
//
// Why another kernel.  Such a chain has ~1 500 row tiles of 32 rows and 46 MFLOP per tile: whatever owns a
// whole tile (a workgroup of the tiled kernel, a wave of the register-resident kernel) is too coarse a unit
// for 1 024 SIMDs — the tiled kernel ran 3.07 "rounds" of workgroups as 4 (77 % of its own speed) with one
// workgroup per CU (104 KB of LDS), all waves in lockstep.  At f32 matrix rates (64 FLOP/clk/SIMD) a layer
// needs 2*K flops per activation float, so writing a layer's activations to memory and reading them back
// (~200 MB per step for both cluster branches, L2 / Infinity-Cache resident) is cheap next to the arithmetic.
// So each layer is its own launch and the unit of work is (32-row tile) x (OCG = 4 output tiles of 32
// channels): 12 560 items for the last layer of cluster.b1, dealt round-robin to persistent waves, every wave
// independent, no barrier, no LDS round trip:
//   * B operand (activations): lane (j,h) loads 16 bytes [8g + 4h, +4) of row j from the row-major input —
//     the gathered feature row (layer 0, through the row map) or the previous layer's output — and two
//     v_permlane32_swap turn them into the four operands of k-group g (as in csrc/mlp_reg.hip);
//   * A operand (weights): fragment order of sad_mlp_pack_f32, one coalesced 16-byte load per lane and
//     k-group for each of the item's four output tiles; both streams run two k-groups ahead;
//   * four accumulators share every B operand (16 MFMAs per k-group);
//   * k ascends from the bias, so every output is SPEC.md §6's fmaf chain bit for bit;
//   * hidden layers store relu(acc) row-major (16 bytes per lane); the last layer pools the rows of each
//     group (DPP segmented max) and writes through the wave's LDS staging buffer: one coalesced store per
//     whole group, 256-byte-contiguous atomic max for groups that continue in another tile.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WHOLE_BIT = 1 << 30;
constexpr int OCG = 4;            // output tiles (of 32 channels) per work item
constexpr int LWAVES = 4;         // waves per workgroup (independent)
constexpr int SLOTS = 16;         // pooled-output staging slots per wave (groups ending in one tile)
constexpr int CW = OCG * 32;      // channels per item

using sad::LayerJob;
using sad::LayerMulti;

#ifdef SAD_LAYER_STAMPS   // measurement build only (tools/probe/layer_stamps.py)
__device__ unsigned long long g_lstamps[3 * 64 * 8];   // [launch % 3][wave slot][start tick, start real, end tick, end real, items, k-loop ticks, -, -]
__device__ unsigned g_llaunch;
__device__ unsigned long long g_lall[4096 * 4];        // last launch, every wave: start real, end real, items, SIMD/CU id
#endif

struct Swapped { float lo, hi; };
__device__ __forceinline__ Swapped swap32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];   // (copy first: bit_cast on r[1] directly reads element 0 with this hipcc)
    return {__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}
// (c0..c3 | c4..c7) in v -> operands of the four MFMAs of the k-group: out[e] = (c_{2e} | c_{2e+1})
__device__ __forceinline__ void to_operands(const float4 v, float *out) {
    const Swapped s01 = swap32(v.x, v.y), s23 = swap32(v.z, v.w);
    out[0] = s01.lo; out[1] = s23.lo; out[2] = s01.hi; out[3] = s23.hi;
}
__device__ __forceinline__ f32x16 mma4(f32x16 acc, const float4 a, const float *b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);
    return acc;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, true); }
struct PoolMasks { int m[5]; };
__device__ __forceinline__ PoolMasks pool_masks(int key) {
    PoolMasks pm;
    pm.m[0] = dpp_i<0x111, 0xF>(key) == key ? -1 : 0;
    pm.m[1] = dpp_i<0x112, 0xF>(key) == key ? -1 : 0;
    pm.m[2] = dpp_i<0x114, 0xF>(key) == key ? -1 : 0;
    pm.m[3] = dpp_i<0x118, 0xF>(key) == key ? -1 : 0;
    pm.m[4] = dpp_i<0x142, 0xA>(key) == key ? -1 : 0;
    return pm;
}
// segmented inclusive max-scan over the 32 rows (values >= 0), all 16 registers step by step (see mlp_reg.hip)
__device__ __forceinline__ f32x16 seg_max16(f32x16 t, const PoolMasks &pm) {
    int x[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float f = t[g];
        x[g] = __builtin_bit_cast(int, f);
    }
#define SAD_STEP(CTRL, RM, K)                                       \
    _Pragma("unroll") for (int g = 0; g < 16; ++g) {                \
        const int u = dpp_i<CTRL, RM>(x[g]) & pm.m[K];              \
        x[g] = u > x[g] ? u : x[g];                                 \
    }
    SAD_STEP(0x111, 0xF, 0)
    SAD_STEP(0x112, 0xF, 1)
    SAD_STEP(0x114, 0xF, 2)
    SAD_STEP(0x118, 0xF, 3)
    SAD_STEP(0x142, 0xA, 4)
#undef SAD_STEP
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = __builtin_bit_cast(float, x[g]);
    return t;
}
__device__ __forceinline__ void atomic_max_pos(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned *>(addr), __builtin_bit_cast(unsigned, v));
}

// One work item: rows [32*rt, 32*rt + 32) x output tiles [OCG*og, OCG*og + OCG) of one layer.
// (GATHER / LAST are compile-time and the k-loop body is branch-free: with a branch or a predicated load in
// the loop the compiler drains every load — s_waitcnt vmcnt(0) — at the top of each k-group, which exposes a
// full L2 round trip per k-group: measured 2.3x slower.)
template <bool GATHER, bool LAST>
__device__ __forceinline__ void layer_item(const LayerJob &jb, const int rt, const int og, const int lane, float *stage) {
    const int j = lane & 31, h = lane >> 5;
    const int total = jb.rowtab[0];
    int q = rt * 32 + j;
    const bool live = q < total;
    if (!live) q = total - 1;                       // rows past the end repeat the last row and store nothing
    int src = q, gv = 0;
    if (GATHER || LAST) gv = jb.row_gid[q];
    if (GATHER) src = jb.row_src[q];
    const int grp = gv & (WHOLE_BIT - 1);
    const bool whole = (gv & WHOLE_BIT) != 0;
    // loads use the scalar-base form (SGPR pair + 32-bit lane offset): stepping through the k-groups costs scalar
    // adds, not per-lane 64-bit address arithmetic (the lane part is fixed per item; buffers are < 4 GB)
    const unsigned xoff = (unsigned)src * (unsigned)jb.ldx * 4u;        // byte offset of this lane's row
    float4 rel = make_float4(0.f, 0.f, 0.f, 0.f);
    if (GATHER && h == 0) {
        const float *pq = jb.xyz + (long long)src * 3;
        const float *pc = jb.new_xyz + (long long)grp * 3;
        rel = make_float4(pq[0] - pc[0], pq[1] - pc[1], pq[2] - pc[2], 0.f);
    }
    const int KG = jb.kg;
    // activations of k-group g for this lane (16 bytes; zero outside the row)
    const char *xb = reinterpret_cast<const char *>(jb.x);
    auto ldb = [&](int g) -> float4 {
        g = g < KG ? g : KG - 1;
        if constexpr (GATHER) {                     // [dx dy dz 0 | f0 f1 ...]: half h holds chunk 2g - 1 + h of the feature row
            const int ch = 2 * g - 1 + h;
            const int cc = ch < 0 ? 0 : (ch < jb.cpr ? ch : jb.cpr - 1);      // always a valid address; selected below
            return *reinterpret_cast<const float4 *>(xb + (size_t)(xoff + 16u * (unsigned)cc));
        } else {
            return *reinterpret_cast<const float4 *>((xb + (size_t)g * 32) + (size_t)(xoff + 16u * (unsigned)h));
        }
    };
    // ... and the selection, applied when the k-group is USED (applied at load time it would wait for the load there)
    auto fixb = [&](float4 v, int g) -> float4 {
        if constexpr (GATHER) {
            const int ch = 2 * g - 1 + h;
            const bool ok = ch >= 0 && ch < jb.cpr;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            const bool first = g == 0 && h == 0;
            v.x = first ? rel.x : v.x; v.y = first ? rel.y : v.y; v.z = first ? rel.z : v.z; v.w = first ? rel.w : v.w;
        }
        return v;
    };
    const float4 *fr = reinterpret_cast<const float4 *>(jb.packed + jb.off + jb.np) + (size_t)(og * OCG) * KG * 64;   // wave-uniform
    const unsigned ulane = (unsigned)lane;
    auto lda = [&](int oc, int g) -> float4 { return (fr + ((size_t)oc * KG + (g < KG ? g : KG - 1)) * 64)[ulane]; };

    f32x16 acc[OCG];
    {
        const float *bias = jb.packed + jb.off + (og * OCG) * 32;
#pragma unroll
        for (int oc = 0; oc < OCG; ++oc)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + oc * 32 + 8 * a + 4 * h);
                acc[oc][4 * a + 0] = bv.x; acc[oc][4 * a + 1] = bv.y; acc[oc][4 * a + 2] = bv.z; acc[oc][4 * a + 3] = bv.w;
            }
    }
    // Weights (L2-resident) run two k-groups ahead, activations FOUR: a row tile's activations are touched for the
    // first time by this XCD here (Infinity Cache / HBM latency), and with two k-groups of cover identical items
    // finished anywhere between 300 and 576 us depending on the CU's distance to the data (measured).
#ifndef SAD_LAYER_AD
#define SAD_LAYER_AD 2
#endif
    constexpr int AD = SAD_LAYER_AD;                // k-groups of weight fragments in flight (2 or 4)
    float4 bq[4], aq[AD][OCG];
#pragma unroll
    for (int u = 0; u < 4; ++u) bq[u] = ldb(u);
#pragma unroll
    for (int u = 0; u < AD; ++u)
#pragma unroll
        for (int oc = 0; oc < OCG; ++oc) aq[u][oc] = lda(oc, u);
#ifdef SAD_LAYER_STAMPS
    const unsigned long long k0 = __builtin_amdgcn_s_memtime();
#endif
    // Issue order inside a k-group: the four MFMAs of ONE output tile (a dependent chain: 64-cycle latency =
    // 64-cycle issue), then at once the reload of that tile's weight fragment — a vector-memory instruction costs
    // tens of issue cycles, which hide in the shadow of the following MFMAs only if the loads are spread between
    // them (all five loads bunched behind the sixteen MFMAs left ~200 of 1 200 cycles per k-group exposed:
    // measured).  The operands of the NEXT k-group are prepared (two lane swaps) behind the second output tile.
    const int KG4 = KG & ~3;
    float ops[4];
    to_operands(fixb(bq[0], 0), ops);
#pragma unroll 1
    for (int g = 0; g < KG4; g += 4) {              // branch-free body; reloads past the end are clamped (harmless re-reads)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float nops[4];
#pragma unroll
            for (int oc = 0; oc < OCG; ++oc) {
                acc[oc] = mma4(acc[oc], aq[u % AD][oc], ops);
                aq[u % AD][oc] = lda(oc, g + AD + u);
                if (oc == 0) bq[u] = ldb(g + 4 + u);                       // (its old content became `ops` one k-group ago)
                if (oc == 1) to_operands(fixb(bq[(u + 1) & 3], g + u + 1), nops);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ops[e] = nops[e];
        }
    }
    for (int u = 0; u < (KG & 3); ++u) {            // up to three trailing k-groups (slots hold groups KG4 + u)
        const float4 bv = u == 0 ? bq[0] : (u == 1 ? bq[1] : bq[2]);
        to_operands(fixb(bv, KG4 + u), ops);
#pragma unroll
        for (int oc = 0; oc < OCG; ++oc) {
            const float4 av = u < 2 ? (u == 0 ? aq[0][oc] : aq[1][oc]) : (AD == 4 && u == 2 ? aq[2 % AD][oc] : lda(oc, KG4 + u));
            acc[oc] = mma4(acc[oc], av, ops);
        }
    }
#ifdef SAD_LAYER_STAMPS
    if (blockIdx.x < 16 && lane == 0) {
        // (the last MFMA's result is consumed below; this stamp sits right behind the issue of the k-loop)
        g_lstamps[((g_llaunch % 3) * 64 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63)) * 8 + 5] += __builtin_amdgcn_s_memtime() - k0;
        g_lstamps[((g_llaunch % 3) * 64 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63)) * 8 + 4] += 1;
    }
#endif
    if (jb.relu) {
#pragma unroll
        for (int oc = 0; oc < OCG; ++oc)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[oc][g] = acc[oc][g] > 0.f ? acc[oc][g] : 0.f;
    }
    if constexpr (!LAST) {
        // hidden layer: row-major output, 16 bytes per lane and k-group (padded channels are exact zeros)
        if (live) {
            float *yrow = jb.y + (long long)q * jb.ldy + (og * OCG) * 32 + 4 * h;
#pragma unroll
            for (int oc = 0; oc < OCG; ++oc)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    *reinterpret_cast<float4 *>(yrow + oc * 32 + 8 * a) = make_float4(acc[oc][4 * a], acc[oc][4 * a + 1], acc[oc][4 * a + 2], acc[oc][4 * a + 3]);
        }
        return;
    } else {
    // last layer: max over the rows of each group, staged in LDS, written once per group
    const int key = live ? grp + 1 : 0;
    const PoolMasks pm = pool_masks(key);
    const int nkey = __shfl_down(key, 1, 64), pkey = __shfl_up(key, 1, 64);
    const bool tail = live && (j == 31 || nkey != key);
    const bool head = live && (j == 0 || pkey != key);
    const unsigned heads = (unsigned)__ballot(head), tails = (unsigned)__ballot(tail);
    const int ngroups = __builtin_popcount(heads);
    const int slot = __builtin_popcount(heads & (0xFFFFFFFFu >> (31 - j))) - 1;
    const bool staged = ngroups <= SLOTS;          // (wave-uniform)
    const int ch0 = og * CW;
#pragma unroll
    for (int oc = 0; oc < OCG; ++oc) {
        const f32x16 t = seg_max16(acc[oc], pm);
        if (!tail) continue;
        if (staged) {
            float *d = stage + slot * CW + oc * 32 + 4 * h;
#pragma unroll
            for (int a = 0; a < 4; ++a) *reinterpret_cast<float4 *>(d + 8 * a) = make_float4(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3]);
        } else {                                    // more groups end in this tile than slots: direct
            float *o = jb.out + (long long)grp * jb.ld_out + jb.col_off + ch0 + oc * 32 + 4 * h;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ch0 + oc * 32 + 8 * a + 4 * h + e >= jb.cout_last) continue;
                    if (whole) o[8 * a + e] = t[4 * a + e];
                    else atomic_max_pos(o + 8 * a + e, t[4 * a + e]);
                }
        }
    }
    if (staged) {
        unsigned rem = tails;
        for (int s = 0; s < ngroups; ++s) {
            const int p = __builtin_ctz(rem);
            rem &= rem - 1;
            const int g = __builtin_amdgcn_readlane(grp, p);
            const bool w = __builtin_amdgcn_readlane((int)whole, p) != 0;
            float *orow = jb.out + (long long)g * jb.ld_out + jb.col_off + ch0;
            const float *sp = stage + s * CW;
#pragma unroll
            for (int k = 0; k < CW / 64; ++k) {
                const int ch = lane + 64 * k;
                if (ch0 + ch < jb.cout_last) {
                    if (w) orow[ch] = sp[ch];
                    else atomic_max_pos(orow + ch, sp[ch]);
                }
            }
        }
    }
    }
}

__global__ __launch_bounds__(LWAVES * 64, 2) void mlp_layer_kernel(const LayerMulti lm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // per wave: SLOTS x CW staging floats
    const int lane = threadIdx.x & 63;
    float *stage = smem + (threadIdx.x >> 6) * (SLOTS * CW);
    // items: job 0 first (heaviest), og fastest so that consecutive items share the activation rows
    const int i0 = ((lm.j[0].rowtab[0] + 31) / 32) * lm.j[0].nog;
    const int nitems = lm.n > 1 ? i0 + ((lm.j[1].rowtab[0] + 31) / 32) * lm.j[1].nog : i0;
    // Static round-robin hand-out.  (A shared work counter does not scale: returning atomics on ONE address are
    // served at ~30 ns each chip-wide, and a layer has up to 13 000 items — 0.4 ms of counter traffic; measured.)
    // Items of a layer cost the same per chain, so round-robin leaves every SIMD within one item of the mean.
    const int nwaves = gridDim.x * LWAVES;
#ifdef SAD_LAYER_STAMPS
    const int sl = ((g_llaunch % 3) * 64 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63)) * 8;
    if (blockIdx.x < 16 && lane == 0) {
        g_lstamps[sl + 0] = __builtin_amdgcn_s_memtime(); g_lstamps[sl + 1] = __builtin_amdgcn_s_memrealtime();
        g_lstamps[sl + 4] = 0; g_lstamps[sl + 5] = 0;
    }
#endif
#ifdef SAD_LAYER_STAMPS
    if (lane == 0 && blockIdx.x * LWAVES + (threadIdx.x >> 6) < 4096) {
        g_lall[(blockIdx.x * LWAVES + (threadIdx.x >> 6)) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        g_lall[(blockIdx.x * LWAVES + (threadIdx.x >> 6)) * 4 + 2] = (nitems - (blockIdx.x * LWAVES + (threadIdx.x >> 6)) + nwaves - 1) / nwaves;
    }
#endif
    for (int item = blockIdx.x * LWAVES + (threadIdx.x >> 6); item < nitems; item += nwaves) {
        const int ji = __builtin_amdgcn_readfirstlane(item < i0 ? 0 : 1);
        const int it = item - (ji ? i0 : 0);
        const int nog = lm.j[ji].nog;
        const LayerJob &jb = lm.j[ji];
        if (jb.gather) {
            if (jb.last) layer_item<true, true>(jb, it / nog, it % nog, lane, stage);
            else layer_item<true, false>(jb, it / nog, it % nog, lane, stage);
        } else {
            if (jb.last) layer_item<false, true>(jb, it / nog, it % nog, lane, stage);
            else layer_item<false, false>(jb, it / nog, it % nog, lane, stage);
        }
    }
#ifdef SAD_LAYER_STAMPS
    if (blockIdx.x < 16 && lane == 0) { g_lstamps[sl + 2] = __builtin_amdgcn_s_memtime(); g_lstamps[sl + 3] = __builtin_amdgcn_s_memrealtime(); }
    if (lane == 0) {
        const int w = blockIdx.x * LWAVES + (threadIdx.x >> 6);
        if (w < 4096) {
            g_lall[w * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_lall[w * 4 + 3] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#endif
}

}  // namespace

namespace sad {

int launch_layers(const LayerMulti &lm, long long max_items, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_layer_kernel), 160 * 1024);
    const size_t lds = sizeof(float) * (size_t)LWAVES * SLOTS * CW;
    static std::atomic<int> per_cu{0};
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mlp_layer_kernel, LWAVES * 64, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 2;
        }
        pc = nb > 4 ? 4 : nb;
        per_cu.store(pc, std::memory_order_relaxed);
    }
    if (get_option(OPT_MLP_DYN_SLOTS) > 0 && get_option(OPT_MLP_DYN_SLOTS) < pc) pc = get_option(OPT_MLP_DYN_SLOTS);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    (void)hipGetLastError();
    long long grid = (long long)cus * pc;
    const long long cap = (max_items + LWAVES - 1) / LWAVES;
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    hipLaunchKernelGGL(mlp_layer_kernel, dim3((unsigned)grid), dim3(LWAVES * 64), lds, st, lm);
#ifdef SAD_LAYER_STAMPS
    {
        static unsigned launch = 0;
        ++launch;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_llaunch), &launch, sizeof launch, 0, hipMemcpyHostToDevice, st);
    }
#endif
    return check_launch("sad_mlp_chain_f32 (layer-streamed chain)");
}

}  // namespace sad

#ifdef SAD_LAYER_STAMPS
extern "C" __attribute__((visibility("default"))) int sad_debug_read_layer_all(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lall), sizeof(unsigned long long) * 4096 * 4);
}
extern "C" __attribute__((visibility("default"))) int sad_debug_read_layer_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lstamps), sizeof(unsigned long long) * 3 * 64 * 8);
}
#endif
