// Layer-streamed fused group -> shared-MLP -> max-pool for gfx950 (SPEC.md §6): "geometry 3", for chains
// that are WIDE and have FEW rows (the size-adaptive cluster layer: 259 -> 256 -> 512 -> 1024 on ~50 000
// packed rows per step).  No reference source exists (/root/reference/README.md:1-2).
//
// Why another kernel.  Such a chain has ~1 500 row tiles of 32 rows and 46 MFLOP per tile: whatever owns a
// whole tile through the whole chain (a workgroup of the tiled kernel, a wave of the register-resident kernel)
// is too coarse a unit for 1 024 SIMDs — the tiled kernel ran 3.07 "rounds" of workgroups as 4 with one
// workgroup per CU (104 KB of LDS), all waves in lockstep.  At f32 matrix rates (64 FLOP/clk/SIMD) a layer
// needs 2*K flops per activation float, so writing a layer's activations to memory and reading them back
// (~200 MB per step for both cluster branches, L2 / Infinity-Cache resident) is cheap next to the arithmetic.
// So each layer is its own launch — a plain GEMM with a fused prologue (gather through the row map) and
// epilogue (bias + ReLU; the last layer pools the rows of each group) — and the unit of work is a block of
// 128 rows x 128 output channels: 3 128 items for the last layer of cluster.b1, dealt round-robin to
// persistent workgroups.  k ascends from the bias inside every output's chain, so results are SPEC.md §6's
// fmaf chains bit for bit (tests/test_gpu_mlp.py::test_layer_streamed_chain_parity).
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace {

// MFMA operand shuffles (swap32 / to_operands / mma4), the two-operation DPP segmented max-scan (pool_masks / seg_max16)
// and atomic_max_pos are the register-resident kernels' (reg_common.h): one copy of each
#include "reg_common.h"

constexpr int LWAVES = 4;         // waves per workgroup (independent)
constexpr int SLOTS = 16;         // pooled-output staging slots per wave and row tile (groups ending in the tile)

using sad::LayerJob;
using sad::LayerMulti;

// ---- workgroup-cooperative layer GEMM --------------------------------------------------------------------
// Measured on this chip (tools/probe/kloop_probe.hip, kloop2_probe.hip): a wave-wide 16-byte global load costs the
// SIMD about one MFMA slot (~64 cycles) of issue — MFMA loops fed by global loads reach 113 / 96 / 92 TFLOP/s (one
// wave per SIMD) with 1 / 4 / 5 loads per 16 MFMAs, and more waves per SIMD do not recover it (140 / 114 / 107 with
// three) — while ds_read_b128 is almost free next to MFMAs: the same loop fed from LDS runs at 129 / 138 / 152
// TFLOP/s with 1 / 2 / 3 waves per SIMD.  So every MFMA operand comes from LDS here, and global memory is read
// once per WORKGROUP: a block of 128 rows x 128 output channels per item, k in chunks of KC k-groups staged in
// two LDS stages (weights copied as they are packed, activations turned into operand order by the loading wave),
// wave (wy, wx) of the 2 x 2 grid computes row tiles {2wy, 2wy+1} x output tiles {2wx, 2wx+1}: 16 MFMAs per
// k-group from four ds_read_b128, and 8 MFMAs per global load instead of 3.2.
#ifndef SAD_LAYER_KC
#define SAD_LAYER_KC 2
#endif
constexpr int KC = SAD_LAYER_KC;                    // k-groups per LDS stage (2 or 4)
constexpr int STAGE_F4 = KC * 8 * 64;               // float4 per stage: KC x (4 weight + 4 activation fragments) x 64 lanes

__device__ __forceinline__ int job_rows(const LayerJob &jb) { return jb.rowtab ? jb.rowtab[0] : jb.rows; }

// Row bookkeeping of one 32-row tile for this lane
struct RowInfo {
    int q, grp;
    bool live, whole;
    unsigned xoff;
    float4 rel;
};
template <bool GATHER>
__device__ __forceinline__ RowInfo row_info(const LayerJob &jb, int rt, int lane, bool need_gid) {
    RowInfo r;
    const int j = lane & 31, h = lane >> 5;
    const int total = job_rows(jb);
    r.q = rt * 32 + j;
    r.live = r.q < total;
    if (!r.live) r.q = total - 1;                   // rows past the end repeat the last row and store nothing
    int src = r.q, gv = 0;
    if (GATHER || need_gid) gv = jb.row_gid[r.q];
    if (GATHER) src = jb.row_src[r.q];
    r.grp = gv & (WHOLE_BIT - 1);
    r.whole = (gv & WHOLE_BIT) != 0;
    r.xoff = (unsigned)src * (unsigned)jb.ldx * 4u;
    r.rel = make_float4(0.f, 0.f, 0.f, 0.f);
    if (GATHER && h == 0) {
        const float *pq = jb.xyz + (long long)src * 3;
        const float *pc = jb.new_xyz + (long long)r.grp * 3;
        r.rel = make_float4(pq[0] - pc[0], pq[1] - pc[1], pq[2] - pc[2], 0.f);
    }
    return r;
}

struct Chunk { float4 a0, a1, a2, a3, b0, b1, b2, b3; };   // (named members: an array captured by lambdas ended up in scratch)
static_assert(KC == 4 || KC == 2, "Chunk holds up to four activation fragments");

// What one wave contributes to chunk c of an item's fill: its quarter of the weight fragments (k-group f / 4 of the chunk,
// output tile f % 4 for f = wave * KC + i) and the activations of its row tile (`fr`) for the chunk's KC k-groups
template <bool GATHER>
__device__ __forceinline__ Chunk load_chunk_of(const LayerJob &jb, const unsigned xoff, const int ob, const int c, const int wave, const int lane) {
    const int h = lane >> 5;
    const int KG = jb.kg;
    const char *xb = reinterpret_cast<const char *>(jb.x);
    const float4 *wf = reinterpret_cast<const float4 *>(jb.packed + jb.off + jb.np) + (size_t)(ob * 4) * KG * 64;   // wave-uniform
    const unsigned ulane = (unsigned)lane;
    float4 ga[4], gb[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const int f = wave * KC + i;
        const int gk = c * KC + (f >> 2);
        ga[i] = (wf + ((size_t)(f & 3) * KG + (gk < KG ? gk : KG - 1)) * 64)[ulane];
    }
#pragma unroll
    for (int u = 0; u < KC; ++u) {
        int g = c * KC + u;
        g = g < KG ? g : KG - 1;
        if constexpr (GATHER) {                     // [dx dy dz 0 | f0 f1 ...]: half h holds chunk 2g - 1 + h of the feature row
            const int ch = 2 * g - 1 + h;
            const int cc = ch < 0 ? 0 : (ch < jb.cpr ? ch : jb.cpr - 1);
            gb[u] = *reinterpret_cast<const float4 *>(xb + (size_t)(xoff + 16u * (unsigned)cc));
        } else {
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF >= 3      // measurement build (wrong results): no activation loads at all (a fused chain reads them from LDS)
            gb[u] = make_float4(0.25f, 0.5f, 0.75f, 1.f);
#elif defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF == 2    // every activation load hits row 0 (cache hits: the issue cost stays, the bytes go)
            gb[u] = *reinterpret_cast<const float4 *>((xb + (size_t)g * 32) + (size_t)(16u * (unsigned)h));
#else
            gb[u] = *reinterpret_cast<const float4 *>((xb + (size_t)g * 32) + (size_t)(xoff + 16u * (unsigned)h));
#endif
        }
    }
    return Chunk{ga[0], ga[1], ga[KC > 2 ? 2 : 0], ga[KC > 2 ? 3 : 0], gb[0], gb[1], gb[KC > 2 ? 2 : 0], gb[KC > 2 ? 3 : 0]};
}

// Chunk 0 of a workgroup's NEXT item, fetched while the current item's last chunk is multiplied: without it every item
// starts with the exposed latency of its row bookkeeping and first loads (two dependent round trips for a gather layer) —
// 6 % of an item of the 512-deep last cluster layer, a third of an item of the gather layer
struct Prefetched {        // (plain words only: with a RowInfo and its bools inside, the object was kept in scratch)
    Chunk ck;
    float4 rel;             // relative coordinates of this lane's row (gather layers)
    unsigned xoff;          // byte offset of its input row
    int valid;
    int next;               // the workgroup's next item (>= the item count: none)
};
// Where the next item comes from.  Static deal: item + gridDim.x.  Queues (common.h, ItemQueue): thread 0 pulled a
// position of its XCD's queue when the current item started and hands the item to the workgroup through `slot` (LDS) one
// barrier before the last chunk; an empty own queue ends the prefetched sequence (the kernel's loop then looks at the
// other queues, latency exposed: only at the tail of a launch).
struct NextItem {
    int ji;                 // job of the workgroup's next item
    int rb, ob;
};
struct NextSource {
    int dyn;
    int pulled;             // thread 0: the position pulled at the start of the item
    int stat;               // static deal: the next item
    int own;                // this workgroup's queue
};
// three ints behind the two fill stages: the published next item, the items of job 0, the items of both jobs (kept in LDS:
// as scalar registers they were spilled inside the chunk loop)
constexpr int META_ITEM = 0, META_I0 = 1, META_N = 2;

template <bool GATHER, bool LAST>
__device__ __forceinline__ Prefetched gemm_item(const LayerMulti &lm, const LayerJob &jb, const int rb, const int ob, float4 *lds,
                                                const Prefetched pin, const NextSource ns) {
    Prefetched pre;                                 // (by value in, by value out: a reference parameter kept the object in scratch)
    pre.valid = 0;
    pre.next = 0x7FFFFFFF;
    int *meta = reinterpret_cast<int *>(lds + 2 * STAGE_F4);
    auto publish_next = [&]() {                     // (thread 0, before the barrier in front of the last chunk)
        if (ns.dyn && threadIdx.x == 0) meta[META_ITEM] = ns.pulled * 8 + ns.own;
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wy = wave >> 1, wx = wave & 1;
    const int KG = jb.kg;
    const int NC = (KG + KC - 1) / KC;              // chunks (a partial last chunk is padded with zero activations)
    // this wave FILLS: weight fragments of k-group (chunk*KC + wave) for the 4 output tiles, and the activation
    // operands of row tile rb*4 + wave for the KC k-groups of the chunk
    unsigned fr_xoff;
    float4 fr_rel;
    if (pin.valid) {
        fr_xoff = pin.xoff;
        fr_rel = pin.rel;
    } else {
        const RowInfo fr = row_info<GATHER>(jb, rb * 4 + wave, lane, false);
        fr_xoff = fr.xoff;
        fr_rel = fr.rel;
    }
    auto load_chunk = [&](int c) -> Chunk { return load_chunk_of<GATHER>(jb, fr_xoff, ob, c, wave, lane); };
    auto store_chunk = [&](const Chunk ck, int c, float4 *st) {     // st: [kg_l][8 fragments: 4 weights, 4 activations][64 lanes]
        const float4 ga[4] = {ck.a0, ck.a1, ck.a2, ck.a3}, gb[4] = {ck.b0, ck.b1, ck.b2, ck.b3};
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            const int f = wave * KC + i;
            st[((f >> 2) * 8 + (f & 3)) * 64 + lane] = ga[i];
        }
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            const int g = c * KC + u;
            float4 v = gb[u];
            if constexpr (GATHER) {
                const int ch = 2 * g - 1 + h;
                const bool ok = ch >= 0 && ch < jb.cpr;
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                const bool first = g == 0 && h == 0;
                v.x = first ? fr_rel.x : v.x; v.y = first ? fr_rel.y : v.y; v.z = first ? fr_rel.z : v.z; v.w = first ? fr_rel.w : v.w;
            }
            if (g >= KG) v = make_float4(0.f, 0.f, 0.f, 0.f);      // padding k-group of the last chunk: x = 0 leaves every chain unchanged
            float ops[4];
            to_operands(v.x, v.y, v.z, v.w, ops);
            st[(u * 8 + 4 + wave) * 64 + lane] = make_float4(ops[0], ops[1], ops[2], ops[3]);
        }
    };
    f32x16 acc[2][2];                               // [row tile][output tile]
    {
        const float *bias = jb.packed + jb.off + (ob * 4 + 2 * wx) * 32;
#pragma unroll
        for (int oc = 0; oc < 2; ++oc)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + oc * 32 + 8 * a + 4 * h);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[rt][oc][4 * a + 0] = bv.x; acc[rt][oc][4 * a + 1] = bv.y; acc[rt][oc][4 * a + 2] = bv.z; acc[rt][oc][4 * a + 3] = bv.w;
                }
            }
    }
    __syncthreads();                                // the previous item's readers (and its pooled-output staging) are done
    Chunk nxt;
    if (pin.valid) nxt = pin.ck;
    else nxt = load_chunk(0);
    store_chunk(nxt, 0, lds);
    if (NC == 1) publish_next();
    __syncthreads();
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF == 4
    float4 wnx[2], xnx[2];
    wnx[0] = wnx[1] = xnx[0] = xnx[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        float4 *cur = lds + (c & 1) * STAGE_F4;
        if (c + 1 < NC) {
            nxt = load_chunk(c + 1);                // global loads in flight while this chunk computes
        } else {                                    // last chunk: the first chunk of the workgroup's next item instead
          const int nitem = __builtin_amdgcn_readfirstlane(ns.dyn ? meta[META_ITEM] : ns.stat);     // (wave-uniform: scalar decode and branches)
          const int mi0 = __builtin_amdgcn_readfirstlane(meta[META_I0]);
          pre.next = nitem;
          if (nitem < __builtin_amdgcn_readfirstlane(meta[META_N])) {
            const int nji = nitem < mi0 ? 0 : 1;
            const int nit = nitem - (nji ? mi0 : 0);
            const int nnog = nji ? lm.j[1].nog : lm.j[0].nog;
            NextItem nx;
            nx.ji = nji;
            nx.rb = nit / nnog;
            nx.ob = nit % nnog;
            // (one expansion per job, no run-time index into the kernel argument and no closure over `pre`: either keeps it in scratch)
#define SAD_FETCH_NEXT(NJ)                                                                         \
    do {                                                                                           \
        if ((NJ).gather) {                                                                         \
            const RowInfo nf = row_info<true>((NJ), nx.rb * 4 + wave, lane, false);                \
            pre.xoff = nf.xoff;                                                                    \
            pre.rel = nf.rel;                                                                      \
            pre.ck = load_chunk_of<true>((NJ), nf.xoff, nx.ob, 0, wave, lane);                     \
        } else {                                                                                   \
            const RowInfo nf = row_info<false>((NJ), nx.rb * 4 + wave, lane, false);               \
            pre.xoff = nf.xoff;                                                                    \
            pre.rel = nf.rel;                                                                      \
            pre.ck = load_chunk_of<false>((NJ), nf.xoff, nx.ob, 0, wave, lane);                    \
        }                                                                                          \
    } while (0)
            if (nx.ji == 0) SAD_FETCH_NEXT(lm.j[0]);
            else SAD_FETCH_NEXT(lm.j[1]);
#undef SAD_FETCH_NEXT
            pre.valid = 1;
          }
        }
        // one k-group = 2 weight + 2 activation fragments from LDS -> 16 MFMAs; reads run one k-group ahead
        float4 wa[2][2], xa[2][2];
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF == 4      // measurement build (racy, wrong results): the chunk's first fragments were read BEFORE the barrier (what a three-stage ring would allow)
        if (c == 0) {
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wa[0][i] = cur[(0 * 8 + 2 * wx + i) * 64 + lane];
            xa[0][i] = cur[(0 * 8 + 4 + 2 * wy + i) * 64 + lane];
        }
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF == 4
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) { wa[0][i] = wnx[i]; xa[0][i] = xnx[i]; }
        }
#endif
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            if (u + 1 < KC) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wa[(u + 1) & 1][i] = cur[((u + 1) * 8 + 2 * wx + i) * 64 + lane];
                    xa[(u + 1) & 1][i] = cur[((u + 1) * 8 + 4 + 2 * wy + i) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // the next k-group's LDS reads stay AHEAD of this k-group's MFMAs
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float ops[4] = {xa[u & 1][rt].x, xa[u & 1][rt].y, xa[u & 1][rt].z, xa[u & 1][rt].w};
#pragma unroll
                for (int oc = 0; oc < 2; ++oc) acc[rt][oc] = mma4(acc[rt][oc], wa[u & 1][oc], ops);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < NC) store_chunk(nxt, c + 1, lds + ((c + 1) & 1) * STAGE_F4);
        if (c + 2 == NC) publish_next();
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF == 4
        {
            const float4 *nx = lds + ((c + 1) & 1) * STAGE_F4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wnx[i] = nx[(0 * 8 + 2 * wx + i) * 64 + lane];
                xnx[i] = nx[(0 * 8 + 4 + 2 * wy + i) * 64 + lane];
            }
        }
#endif
        __syncthreads();
    }
    if (jb.relu) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[rt][oc][g] = acc[rt][oc][g] > 0.f ? acc[rt][oc][g] : 0.f;
    }
    // ---- epilogue: this wave's 2 row tiles x 2 output tiles --------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const RowInfo ri = row_info<false>(jb, rb * 4 + 2 * wy + rt, lane, LAST);
        const int j = lane & 31;
        if constexpr (!LAST) {
            // hidden layer: row-major output, 16 bytes per lane (padded channels are exact zeros)
#if defined(SAD_LAYER_WHATIF) && SAD_LAYER_WHATIF >= 1      // measurement build: the hidden activations are not stored (one sentinel row keeps the code alive)
            if (ri.live && ri.q == 0x7FFFFFF0) {
#else
            if (ri.live) {
#endif
                float *yrow = jb.y + (long long)ri.q * jb.ldy + (ob * 4 + 2 * wx) * 32 + 4 * h;
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        *reinterpret_cast<float4 *>(yrow + oc * 32 + 8 * a) =
                            make_float4(acc[rt][oc][4 * a], acc[rt][oc][4 * a + 1], acc[rt][oc][4 * a + 2], acc[rt][oc][4 * a + 3]);
            }
        } else {
            // last layer: max over the rows of each group; results are collected in this wave's slice of the (now
            // free) fill stages and written once per group: a coalesced store, or a 256-byte atomic max for a group
            // that continues in another tile
            float *stage = reinterpret_cast<float *>(lds) + wave * (SLOTS * 64);
            const int key = ri.live ? ri.grp + 1 : 0;
            const PoolMasks pm = pool_masks(key);
            const int nkey = __shfl_down(key, 1, 64), pkey = __shfl_up(key, 1, 64);
            const bool tail = ri.live && (j == 31 || nkey != key);
            const bool head = ri.live && (j == 0 || pkey != key);
            const unsigned heads = (unsigned)__ballot(head), tails = (unsigned)__ballot(tail);
            const int ngroups = __builtin_popcount(heads);
            const int slot = __builtin_popcount(heads & (0xFFFFFFFFu >> (31 - j))) - 1;
            const bool staged = ngroups <= SLOTS;  // (wave-uniform)
            const int ch0 = (ob * 4 + 2 * wx) * 32;
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) {
                const f32x16 t = seg_max16(acc[rt][oc], pm);
                if (!tail) continue;
                if (staged) {
                    float *d = stage + slot * 64 + oc * 32 + 4 * h;
#pragma unroll
                    for (int a = 0; a < 4; ++a) *reinterpret_cast<float4 *>(d + 8 * a) = make_float4(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3]);
                } else {                            // more groups end in this tile than slots: direct
                    float *o = jb.out + (long long)ri.grp * jb.ld_out + jb.col_off + ch0 + oc * 32 + 4 * h;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (ch0 + oc * 32 + 8 * a + 4 * h + e >= jb.cout_last) continue;
                            if (ri.whole) o[8 * a + e] = t[4 * a + e];
                            else atomic_max_pos(o + 8 * a + e, t[4 * a + e]);
                        }
                }
            }
            if (staged) {
                unsigned rem = tails;
                for (int s = 0; s < ngroups; ++s) {
                    const int p = __builtin_ctz(rem);
                    rem &= rem - 1;
                    const int g = __builtin_amdgcn_readlane(ri.grp, p);
                    const bool w = __builtin_amdgcn_readlane((int)ri.whole, p) != 0;
                    float *orow = jb.out + (long long)g * jb.ld_out + jb.col_off + ch0;
                    const float *sp = stage + s * 64;
                    if (ch0 + lane < jb.cout_last) {
                        if (w) orow[lane] = sp[lane];
                        else atomic_max_pos(orow + lane, sp[lane]);
                    }
                }
            }
        }
    }
    return pre;
}

// (two workgroups per CU by LDS; amdgpu_waves_per_eu tells the register allocator so — left alone it squeezes the
// kernel into 128 registers for four waves per SIMD and parks the prefetched fragments in scratch)
__global__ __launch_bounds__(LWAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_layer_kernel(const LayerMulti lm) {
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];     // two fill stages (the first doubles as pooled-output staging)
    // items = (block of 128 rows) x (block of 128 output channels), job 0 first (heaviest); the channel block runs
    // fastest, so the workgroups that share a row block (and its activations in L2) run at about the same time.
    // Items are pulled from the launching stream's per-XCD queues (common.h, ItemQueue): under an FPS kernel of another
    // stream some of these workgroups only start when others finish, and with a static round-robin they still owned a full
    // share of the items (cluster dispatch 757 us alone, 1 069 us beside one FPS kernel).  lm.queue == NULL: static deal.
    int *meta = reinterpret_cast<int *>(lds4 + 2 * STAGE_F4);
    const int i0 = ((job_rows(lm.j[0]) + 127) / 128) * lm.j[0].nog;
    const int nitems = lm.n > 1 ? i0 + ((job_rows(lm.j[1]) + 127) / 128) * lm.j[1].nog : i0;
    const int dyn = lm.queue != nullptr;
    const sad::ItemQueue Q = sad::itemq_init(lm.queue, dyn ? 8 : 0);
    int item = blockIdx.x;
    if (threadIdx.x == 0) {
        meta[META_I0] = i0;
        meta[META_N] = nitems;
        if (dyn) meta[META_ITEM] = sad::itemq_item(Q, Q.own, sad::itemq_pull(Q));
    }
    __syncthreads();
    if (dyn) item = meta[META_ITEM];
    Prefetched pre;
    pre.valid = 0;
    while (true) {
        if (item >= nitems) {
            if (!dyn) break;
            __syncthreads();                        // (every thread has read the published item)
            if (threadIdx.x == 0) meta[META_ITEM] = sad::itemq_steal(Q, nitems);
            __syncthreads();
            item = meta[META_ITEM];
            if (item >= nitems) break;
            pre.valid = 0;
        }
        const int ji = item < i0 ? 0 : 1;
        const int it = item - (ji ? i0 : 0);
        const LayerJob &jb = lm.j[ji];
        const int nog = jb.nog;
        NextSource ns;
        ns.dyn = dyn;
        ns.pulled = 0;
        if (dyn && threadIdx.x == 0) ns.pulled = sad::itemq_pull(Q);      // (returning atomic: in flight while the item is computed)
        ns.stat = item + (int)gridDim.x;
        ns.own = Q.own;
        if (jb.gather) {
            if (jb.last) pre = gemm_item<true, true>(lm, jb, it / nog, it % nog, lds4, pre, ns);
            else pre = gemm_item<true, false>(lm, jb, it / nog, it % nog, lds4, pre, ns);
        } else {
            if (jb.last) pre = gemm_item<false, true>(lm, jb, it / nog, it % nog, lds4, pre, ns);
            else pre = gemm_item<false, false>(lm, jb, it / nog, it % nog, lds4, pre, ns);
        }
        item = pre.next;
    }
    if (dyn && threadIdx.x == 0) sad::itemq_done(Q, (int)gridDim.x);      // the last workgroup out re-arms the queues for the next launch
}

}  // namespace

namespace sad {

int *stream_item_queue(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int *> queues;     // (never freed: 1 KB per stream that ever launched a layer)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    auto it = queues.find({dev, st});
    if (it != queues.end()) return it->second;
    int *q = nullptr;
    // (the zero fill is enqueued on the launching stream: torch's streams do not synchronise with the null stream, and the
    // first launch that pulls from the queue must see the zeros.  Entries are keyed by the raw stream handle and live as long
    // as the process: a destroyed stream whose handle is reused inherits a queue that its last launch re-armed, i.e. zeros)
    if (hipMalloc(&q, sizeof(int) * ITEMQ_INTS) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemsetAsync(q, 0, sizeof(int) * ITEMQ_INTS, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(q);
        return nullptr;
    }
    queues[{dev, st}] = q;
    return q;
}

int launch_layers(const LayerMulti &lm_in, long long max_items, hipStream_t st) {
    LayerMulti lm = lm_in;
    lm.queue = get_option(OPT_MLP_LAYER_QUEUE) == 1 ? stream_item_queue(st) : nullptr;
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_layer_kernel), 160 * 1024);
    const size_t lds = sizeof(float4) * ((size_t)2 * STAGE_F4 + 1);     // + the three meta ints
    static std::atomic<int> per_cu{0};
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mlp_layer_kernel, LWAVES * 64, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 2;
        }
        pc = nb > 2 ? 2 : nb;       // (three fit when the allocator stays under 170 registers, and run 20 % slower: 876 vs 728 us on cluster.b1)
        per_cu.store(pc, std::memory_order_relaxed);
    }
    if (get_option(OPT_MLP_DYN_SLOTS) > 0 && get_option(OPT_MLP_DYN_SLOTS) < pc) pc = get_option(OPT_MLP_DYN_SLOTS);
    const int cus = sad::device_cus();
    long long grid = (long long)cus * pc;
    const long long cap = max_items;
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    hipLaunchKernelGGL(mlp_layer_kernel, dim3((unsigned)grid), dim3(LWAVES * 64), lds, st, lm);
    return check_launch("sad_mlp_chain_f32 (layer-streamed chain)");
}

}  // namespace sad

