// Register-resident fused group -> shared-MLP -> max-pool with a workgroup-shared weight ring: "geometry 4"
// (SPEC.md §6).  No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
//
// mlp_reg.hip keeps a 32-row tile in ONE wave's registers through a 3-layer chain and streams the weight
// fragments L2 -> VGPR: one wave-wide 16-byte global load per four MFMAs.  Measured on this chip
// (tools/probe/kloop_probe.hip, kloop2_probe.hip) such a load costs the SIMD about one MFMA slot of issue, however
// many waves are resident, while ds_read_b128 is almost free next to MFMAs — so the k-loops of the wide SA3 chains
// sit near 105 TFLOP/s of 157.  Here the four waves of a workgroup carry four tiles of the SAME chain through the
// chain in lockstep and share every weight fragment through LDS:
//   * the fragment order of a tile is a fixed list (mlp_reg.hip's PIPE variant); sad_mlp_pack_f32 stores a copy of
//     the weights in exactly that order (the "stream image", 1 KB per fragment, padded to whole stages);
//   * a stage = 8 consecutive fragments; wave w fetches fragments 2w, 2w+1 of stage s+2 at the top of stage s and
//     writes them to ring slot (s+2) mod 3 at its end, followed by the stage's only barrier; every wave then reads
//     all 8 fragments of a stage from LDS: one global load per 16 MFMAs instead of one per 4;
//   * the ring is never drained: the last two stages of a tile fetch the first two of the workgroup's next tile;
//   * the layer-0 operands (too wide for registers next to layer 1's accumulators, and 4 x 17 KB of LDS images
//     would leave one workgroup per CU) are re-gathered from the feature rows for each of layer 0's output tiles:
//     17 loads per pass, 1 per 26 MFMAs over the tile.
// Everything else — operand shuffles, transient / persistent layer alternation, DPP pooling, staged output, bit-exact
// fmaf chains in ascending k — is mlp_reg.hip's (reg_common.h).  Work items (4 tiles of one chain) are dealt
// round-robin to persistent workgroups.
#include "common.h"

namespace {

#include "reg_common.h"

constexpr int WAVES = 4;
constexpr int S = 8;                 // fragments per stage
constexpr int NS = 3;                // ring slots (stages): being read / complete / being written
constexpr int FPW = S / WAVES;       // fragments each wave fetches per stage
constexpr int RING_F4 = NS * S * 64; // float4 in the ring
// (work items are pulled from the per-XCD queues of common.h, one item ahead)
// pooled-output staging per wave (floats): 8 slots of a tile's COUT channels (groups ending inside one tile)
__host__ __device__ constexpr int pool_floats(int family) { return family == 2 ? 8 * 256 : 8 * 128; }

#ifdef SAD_COOP_STAMPS   // measurement build only (tools/probe/coop_stamps.py): s_memtime at the phase boundaries of a tile
__device__ unsigned long long g_cstamps[64 * 16];
__device__ unsigned long long g_call[2048 * 4];     // per workgroup: start / end (s_memrealtime), items, hardware id
#define SAD_CSTAMP(i)                                                                                              \
    do {                                                                                                           \
        if (blockIdx.x < 16 && lane == 0) g_cstamps[(blockIdx.x * 4 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define SAD_CACC(i, t0)                                                                                            \
    do {                                                                                                           \
        if (blockIdx.x < 16 && lane == 0) g_cstamps[(blockIdx.x * 4 + wave) * 16 + (i)] += __builtin_amdgcn_s_memtime() - (t0); \
    } while (0)
#else
#define SAD_CSTAMP(i)
#define SAD_CACC(i, t0)
#endif

struct RingState {
    float4 *ring;        // workgroup-shared: [NS][S][64 lanes]
    int slot;            // ring slot of the stage being consumed (wave-uniform)
};

// One tile of 32 packed rows through a 3-layer chain (full-width layers: NG1 = 4 NO0, NG2 = 4 NO1), weights from
// the ring.  `sbase`: this chain's stream image; `nbase`: the stream image of the workgroup's next item.
template <int NT0, int NO0, int NO1, int NO2, int PF>
__device__ __forceinline__ void coop_tile(const RegChain &c, const int tile, const float *__restrict__ sbias, const int lane, const int wave,
                                          RingState &rs, const float4 *__restrict__ sbase, const float4 *__restrict__ nbase,
                                          float *lds_pool, const sad::ItemQueue &Q, int &grabbed) {
    constexpr int COUT = NO2 * 32;
    constexpr int NI = NO1 * 4, NG2 = NO1 * 4;
    constexpr int PER = NT0 + NI, TOT = NO0 * PER, P = TOT + NO2 * NG2;
    constexpr int NSTG = (P + S - 1) / S;
    constexpr bool ROLL2 = NG2 % S == 0;   // layer-2 tiles span whole stages: the loop over them can stay rolled
    const int j = lane & 31, h = lane >> 5;
    SAD_CSTAMP(0);
    const int total = c.rowtab[0];
    int q = tile * 32 + j;
    const bool live = q < total;
    if (!live) q = total - 1;                      // rows past the end (and whole tiles past it) repeat the last row and store nothing
    const int src = c.row_src[q];
    const int gv = c.row_gid[q];
    const int grp = gv & (WHOLE_BIT - 1);
    const bool whole = (gv & WHOLE_BIT) != 0;
    const char *pf = reinterpret_cast<const char *>(c.feat + (long long)src * c.ld_feat);
    float4 rel = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h == 0) {
        const float *pq = c.xyz + (long long)src * 3;
        const float *pc = c.new_xyz + (long long)grp * 3;
        rel = make_float4(pq[0] - pc[0], pq[1] - pc[1], pq[2] - pc[2], 0.f);
    }
    const int cpr = c.cpr;
    // layer-0 operand of k-group r: lane half h takes 16-byte chunk 2r - 1 + h of the feature row ([dx dy dz 0 | f0 f1 ...])
    auto xload = [&](int r) -> float4 {
        const int ch = 2 * r - 1 + h;
        const int cc = ch < 0 ? 0 : (ch < cpr ? ch : cpr - 1);
        return *reinterpret_cast<const float4 *>(pf + 16 * cc);
    };
    auto xfix = [&](float4 v, int r, float *ops) {
        const int ch = 2 * r - 1 + h;
        const bool ok = ch >= 0 && ch < cpr;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        if (r == 0) {
            v.x = h == 0 ? rel.x : v.x; v.y = h == 0 ? rel.y : v.y; v.z = h == 0 ? rel.z : v.z; v.w = h == 0 ? rel.w : v.w;
        }
        to_operands(v.x, v.y, v.z, v.w, ops);
    };
    const float *sb0 = sbias, *sb1 = sbias + NO0 * 32, *sb2 = sb1 + NO1 * 32;
    const int key = live ? grp + 1 : 0;
    float *orow = c.out + (long long)grp * c.ld_out + c.col_off;
    const int nkey = __shfl_down(key, 1, 64);
    const bool tail = live && (j == 31 || nkey != key);
    Stage sg;
    {
        const int pkey = __shfl_up(key, 1, 64);
        const bool head = live && (j == 0 || pkey != key);
        const unsigned heads = (unsigned)__ballot(head);
        sg.tails = (unsigned)__ballot(tail);
        sg.ngroups = __builtin_popcount(heads);
        sg.slot = __builtin_popcount(heads & (0xFFFFFFFFu >> (31 - j))) - 1;
        sg.lds = sg.ngroups * COUT <= PF ? lds_pool : nullptr;    // (wave-uniform)
    }

    // ---- the ring --------------------------------------------------------------------------------------------
    float4 *const ring = rs.ring;
    int slot = rs.slot;
    int srel = 0;                                   // stage of this tile being consumed
    float4 T0, T1;                                  // this wave's two fragments of stage srel + 2, in flight
    const unsigned ulane = (unsigned)lane;
    auto stage_begin = [&]() {
        const int s2 = srel + 2;
        const float4 *sp = s2 < NSTG ? sbase + (size_t)(s2 * S + FPW * wave) * 64 : nbase + (size_t)((s2 - NSTG) * S + FPW * wave) * 64;
        T0 = sp[ulane];
        T1 = (sp + 64)[ulane];
    };
    auto stage_end = [&]() {
        const int ws = slot + 2 >= NS ? slot + 2 - NS : slot + 2;
        float4 *wp = ring + (ws * S + FPW * wave) * 64 + lane;
        wp[0] = T0;
        wp[64] = T1;
        __syncthreads();
        slot = slot + 1 == NS ? 0 : slot + 1;
        ++srel;
    };
    // fragment at position p of this tile (compile-time p % S = m, d = stages ahead of the one being consumed)
    auto ring_read = [&](int m, int d) -> float4 {
        int sl = slot + d;
        sl = sl >= NS ? sl - NS : sl;
        return ring[(sl * S + m) * 64 + lane];
    };
    float4 a[2];
    a[0] = ring_read(0, 0);
    a[1] = ring_read(1, 0);

    f32x16 acc1[NO1];
#pragma unroll
    for (int o1 = 0; o1 < NO1; ++o1) acc1[o1] = bias_tile(sb1 + o1 * 32, h);
    float in2[NG2 * 4];
    // layer-0 operand loads: three k-groups ahead, across the layer-1 blocks
    constexpr int NX = NO0 * NT0;
    float4 xq[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) xq[u] = xload(u % NT0);
    {
        f32x16 t;
        float bt[16];
#pragma unroll
        for (int p = 0; p < TOT; ++p) {
            const int o = p / PER, r = p % PER, m = p % S;
            if (m == 0) stage_begin();
            // the pull for the item after the next: a returning atomic, issued BEHIND the stage's fragment loads — vector
            // memory operations return in order, so in front of them it would hold up the first stage of every item
            if (p == 0 && threadIdx.x == 0 && Q.nq) grabbed = sad::itemq_item(Q, Q.own, sad::itemq_pull(Q));
            if (r == 0) t = bias_tile(sb0 + o * 32, h);
            if (r < NT0) {
                const int n = o * NT0 + r;
                float ops[4];
                xfix(xq[n % 3], r, ops);
                t = mma4(t, a[p % 2], ops);
#ifdef SAD_COOP_NOGATHER     // measurement build (wrong results): the feature rows are gathered for layer 0's FIRST output tile only
                if (n + 3 < NT0) xq[n % 3] = xload((n + 3) % NT0);
#else
                if (n + 3 < NX) xq[n % 3] = xload((n + 3) % NT0);
#endif
            } else {
                const int i = r - NT0;
                acc1[i >> 2] = mma4(acc1[i >> 2], a[p % 2], bt + 4 * (i & 3));
            }
            a[p % 2] = ring_read((p + 2) % S, (p + 2) / S - p / S);
            __builtin_amdgcn_sched_barrier(0);
            if (r == NT0 - 1) {
                t = relu16(t);
#pragma unroll
                for (int g = 0; g < 4; ++g) to_operands(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], bt + 4 * g);
            }
            if (m == S - 1) stage_end();
        }
    }
    SAD_CSTAMP(1);
#pragma unroll
    for (int o1 = 0; o1 < NO1; ++o1) {
        const f32x16 u = relu16(acc1[o1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) to_operands(u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3], in2 + 16 * o1 + 4 * g);
    }
    const PoolMasks pm = pool_masks(key);
    SAD_CSTAMP(2);
#ifdef SAD_COOP_STAMPS
    if (blockIdx.x < 16 && lane == 0) { g_cstamps[(blockIdx.x * 4 + wave) * 16 + 5] = 0; g_cstamps[(blockIdx.x * 4 + wave) * 16 + 6] = 0; }
#endif
    auto l2_tile = [&](const int o, const int p0) {   // p0: position of the tile's first fragment (modulo the stage size when rolled)
#ifdef SAD_COOP_STAMPS
        const unsigned long long tk = __builtin_amdgcn_s_memtime();
#endif
        f32x16 t2 = bias_tile(sb2 + o * 32, h);
#pragma unroll
        for (int i = 0; i < NG2; ++i) {
            const int p = p0 + i;
            const int m = p % S;
            if (m == 0) stage_begin();
            t2 = mma4(t2, a[p % 2], in2 + 4 * i);
            a[p % 2] = ring_read((p + 2) % S, (p + 2) / S - p / S);
            __builtin_amdgcn_sched_barrier(0);
            if (m == S - 1) stage_end();
        }
        SAD_CACC(5, tk);
#ifdef SAD_COOP_STAMPS
        const unsigned long long tp = __builtin_amdgcn_s_memtime();
#endif
        if (sg.lds) pool_stage<COUT>(relu16(t2), pm, tail, sg, o, h);
        else pool_store(relu16(t2), pm, tail, whole, orow, o, h, c);
        SAD_CACC(6, tp);
    };
    if constexpr (ROLL2) {
#pragma unroll 1
        for (int o = 0; o < NO2; ++o) l2_tile(o, TOT);
    } else {
#pragma unroll
        for (int o = 0; o < NO2; ++o) l2_tile(o, TOT + o * NG2);
    }
    SAD_CSTAMP(3);
    if (P % S != 0) stage_end();                   // the padded last stage
    if (sg.lds) stage_flush<COUT>(sg, grp, whole, lane, c);
    SAD_CSTAMP(4);
    rs.slot = slot;
}

// Shapes (sad::reg_shape_id).  Family 1 (SA2): 2: 67 -> 64 -> 64 -> 128, 3: 67 -> 64 -> 96 -> 128.
// Family 2 (SA3): 4: 131 -> 128 -> 128 -> 256, 5: 131 -> 128 -> 192 -> 256, 6: 131 -> 128 -> 256 -> 256.
template <int FAMILY>
__device__ __forceinline__ void run_coop(const RegChain &c, int shape, int tile, const float *sb, int lane, int wave, RingState &rs,
                                         const float4 *sbase, const float4 *nbase, float *lp, const sad::ItemQueue &Q, int &grabbed) {
    constexpr int PF = pool_floats(FAMILY);
    if constexpr (FAMILY == 1) {
        if (shape == 2) coop_tile<9, 2, 2, 4, PF>(c, tile, sb, lane, wave, rs, sbase, nbase, lp, Q, grabbed);
        else coop_tile<9, 2, 3, 4, PF>(c, tile, sb, lane, wave, rs, sbase, nbase, lp, Q, grabbed);
    } else {
        if (shape == 4) coop_tile<17, 4, 4, 8, PF>(c, tile, sb, lane, wave, rs, sbase, nbase, lp, Q, grabbed);
        else if (shape == 5) coop_tile<17, 4, 6, 8, PF>(c, tile, sb, lane, wave, rs, sbase, nbase, lp, Q, grabbed);
        else coop_tile<17, 4, 8, 8, PF>(c, tile, sb, lane, wave, rs, sbase, nbase, lp, Q, grabbed);
    }
}

template <int FAMILY>
__global__ __launch_bounds__(WAVES * 64, FAMILY == 2 ? 2 : 3) void mlp_coop_kernel(const RegMulti mp) {
    // [ring: NS stages x 8 fragments x 1 KB][per wave: pooled-output staging][per chain: biases]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *ring = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PF = pool_floats(FAMILY);
    float *lds_pool = smem + RING_F4 * 4 + wave * PF;
    float *sbias = smem + RING_F4 * 4 + WAVES * PF;
    static_assert(sad::REG_MAX_CHAINS == 3, "chain selection below is written out for three chains");
    int bo = 0, b1 = 0, b2 = 0;
    for (int ci = 0; ci < mp.n; ++ci) {
        const RegChain &c = mp.c[ci];
        if (ci == 1) b1 = bo;
        if (ci == 2) b2 = bo;
        for (int l = 0; l < 3; ++l) {
            for (int i = tid; i < c.np[l]; i += WAVES * 64) sbias[bo + i] = c.packed[c.off[l] + i];
            bo += c.np[l];
        }
    }
    // items = 4 consecutive tiles of one chain, chain 0 first (heaviest)
    const int t0 = ((mp.c[0].rowtab[0] + 31) / 32 + WAVES - 1) / WAVES;
    const int t1 = mp.n > 1 ? t0 + ((mp.c[1].rowtab[0] + 31) / 32 + WAVES - 1) / WAVES : t0;
    const int nitems = mp.n > 2 ? t1 + ((mp.c[2].rowtab[0] + 31) / 32 + WAVES - 1) / WAVES : t1;
    int *s_next = reinterpret_cast<int *>(sbias + ((bo + 3) & ~3));      // [0], [1]: item indices handed from thread 0 to the workgroup
    const sad::ItemQueue Q = sad::itemq_init(mp.counter, mp.nq);
    // Test knob mlp_steal_after = v > 0 (eight queues only): the own queue counts as empty after v - 1 items, items are
    // then taken from the OTHER queues one at a time with nothing prefetched — consecutive items of a workgroup change chain
    // wherever the queues do, so the ring refill below runs at every such change instead of only at the tail of a dispatch.
    // When nothing is left elsewhere the own queue is used after all: every item is taken exactly once under any placement.
    const bool steal_mode = mp.steal_after > 0 && mp.nq == 8;
    int own_budget = mp.steal_after - 1;            // (thread 0)
    auto take = [&]() -> int {
        int it = nitems;
        if (own_budget > 0) {
            it = sad::itemq_item(Q, Q.own, sad::itemq_pull(Q));
            --own_budget;
        }
        if (it >= nitems) it = sad::itemq_steal(Q, nitems);
        if (it >= nitems) it = sad::itemq_item(Q, Q.own, sad::itemq_pull(Q));
        return it < nitems ? it : nitems;
    };
    if (tid == 0) {
        sad::itemq_claim(Q, mp.check_id);           // (test knob mlp_check_inuse; nothing when check_id == 0)
        if (mp.nq == 0) {                           // A/B knob (mlp_static=1): static round-robin deal
            s_next[0] = (int)blockIdx.x;
            s_next[1] = (int)(blockIdx.x + gridDim.x);
        } else if (steal_mode) {
            s_next[0] = take();
            s_next[1] = nitems;
        } else {
            const int pos = sad::itemq_pull(Q, 2);  // the first two items with one atomic
            int first = sad::itemq_item(Q, Q.own, pos), second = sad::itemq_item(Q, Q.own, pos + 1);
            if (first >= nitems) first = sad::itemq_steal(Q, nitems);
            s_next[0] = first;
            s_next[1] = second;
        }
    }
    __syncthreads();                                // (also: the biases are in place)
    int item = s_next[0], nxt = s_next[1];
    __syncthreads();                                // s_next is written again at the end of the first item
#ifdef SAD_COOP_STAMPS
    if (blockIdx.x < 2048 && tid == 0) {
        g_call[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        g_call[blockIdx.x * 4 + 1] = 0;
        g_call[blockIdx.x * 4 + 2] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_call[blockIdx.x * 4 + 3] = ((unsigned long long)(xcc & 0xF) << 32) | hw;
    }
    if (blockIdx.x < 16 && lane == 0) {
        g_cstamps[(blockIdx.x * 4 + wave) * 16 + 8] = __builtin_amdgcn_s_memtime();
        g_cstamps[(blockIdx.x * 4 + wave) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
        g_cstamps[(blockIdx.x * 4 + wave) * 16 + 12] = 0;
    }
#endif
    const unsigned ulane = (unsigned)lane;
    auto stream_of = [&](int it) -> const float4 * {
        const int ci = it < t0 ? 0 : (it < t1 ? 1 : 2);
        const RegChain &c = mp.c[ci];
        return reinterpret_cast<const float4 *>(c.packed + c.stream_off);
    };
    if (item < nitems) {   // prologue: stages 0 and 1 of the first item
        const float4 *sp = stream_of(item) + (size_t)(FPW * wave) * 64;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < FPW; ++e) ring[(s * S + FPW * wave + e) * 64 + lane] = (sp + (size_t)(s * S + e) * 64)[ulane];
    }
    __syncthreads();
    RingState rs{ring, 0};
    const sad::ItemQueue Qoff{Q.q, 0, Q.own};       // (steal mode: the tile pulls nothing ahead)
    while (item < nitems) {                         // (workgroup-uniform)
        int grabbed = 0;                            // the item after the next: pulled inside the tile, in flight while it runs
        const int ci = __builtin_amdgcn_readfirstlane(item < t0 ? 0 : (item < t1 ? 1 : 2));
        const int tg = item - (ci == 0 ? 0 : (ci == 1 ? t0 : t1));
        // the tile's last two stages fetch the first two of the stream it expects to run next
        const float4 *expect = stream_of(nxt < nitems ? nxt : item);
        run_coop<FAMILY>(mp.c[ci], mp.shape[ci], tg * WAVES + wave, sbias + (ci == 0 ? 0 : (ci == 1 ? b1 : b2)), lane, wave, rs, stream_of(item),
                 expect, lds_pool, steal_mode ? Qoff : Q, grabbed);
        if (tid == 0) {
            if (mp.nq == 0) grabbed = nxt + (int)gridDim.x;
            else if (steal_mode) grabbed = take();
            else if (grabbed >= nitems && nxt >= nitems) grabbed = sad::itemq_steal(Q, nitems);   // own queue empty and nothing in hand
            s_next[0] = grabbed;
        }
        __syncthreads();
        item = nxt;
        nxt = s_next[0];                            // (rewritten only after the next item's many barriers)
        if (item >= nitems) {                       // (the item in hand was past the end, the stolen one is not)
            item = nxt;
            nxt = nitems;
        }
        if (item < nitems && stream_of(item) != expect) {
            // an item taken from another queue may belong to another chain than the one whose first stages the ring
            // holds (tail of a dispatch only): fill the two ring slots again (the barrier above closed every read)
            const float4 *sp = stream_of(item) + (size_t)(FPW * wave) * 64;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int sl = rs.slot + s >= NS ? rs.slot + s - NS : rs.slot + s;
#pragma unroll
                for (int e = 0; e < FPW; ++e) ring[(sl * S + FPW * wave + e) * 64 + lane] = (sp + (size_t)(s * S + e) * 64)[ulane];
            }
            if (tid == 0 && mp.nq == 8) atomicAdd(Q.q + sad::ITEMQ_REFILLS, 1);   // test instrumentation (common.h)
            __syncthreads();
        }
#ifdef SAD_COOP_STAMPS
        if (blockIdx.x < 16 && lane == 0) g_cstamps[(blockIdx.x * 4 + wave) * 16 + 12] += 1;
#endif
    }
    if (tid == 0 && mp.nq) sad::itemq_done(Q, (int)gridDim.x);    // the last workgroup out re-arms the queues for the next launch
#ifdef SAD_COOP_STAMPS
    if (blockIdx.x < 2048 && tid == 0) g_call[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x < 16 && lane == 0) {
        g_cstamps[(blockIdx.x * 4 + wave) * 16 + 9] = __builtin_amdgcn_s_memtime();
        g_cstamps[(blockIdx.x * 4 + wave) * 16 + 11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace

namespace sad {

bool coop_shape(int shape) { return shape >= 2 && shape <= 6; }

// positions of a tile's fragment stream (padded to whole stages); 0 = no stream image for this shape
long long coop_stream_frags(int shape, const int *kp, const int *np) {
    if (!coop_shape(shape)) return 0;
    const int NT0 = kp[0] / 8, NO0 = np[0] / 32, NO1 = np[1] / 32, NO2 = np[2] / 32;
    const long long P = (long long)NO0 * (NT0 + 4 * NO1) + (long long)NO2 * 4 * NO1;
    return (P + S - 1) / S * S;
}

template <int FAMILY>
static int launch_coop_family(const RegMulti &mp, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_coop_kernel<FAMILY>), 160 * 1024);
    static std::atomic<int> per_cu{0};
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mlp_coop_kernel<FAMILY>, WAVES * 64, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 2;
        }
        constexpr int want = FAMILY == 2 ? 2 : 3;
        pc = nb > want ? want : nb;
        per_cu.store(pc, std::memory_order_relaxed);
    }
    if (get_option(OPT_MLP_DYN_SLOTS) > 0 && get_option(OPT_MLP_DYN_SLOTS) < pc) pc = get_option(OPT_MLP_DYN_SLOTS);   // A/B knob
    const int cus = sad::device_cus();
    long long grid = (long long)cus * pc;
    const long long cap = mp.max_tiles / WAVES + mp.n;          // never more workgroups than items could exist
    if (grid > cap) grid = cap < 1 ? 1 : cap;
    RegMulti mq = mp;
    if (grid < 64) mq.nq = 1;                                   // (a small grid may leave XCDs without a workgroup: one queue)
    if (get_option(OPT_MLP_STATIC) == 1) mq.nq = 0;             // A/B knob: static round-robin deal
    mq.steal_after = mq.nq == 8 ? get_option(OPT_MLP_STEAL_AFTER) : 0;   // test knobs (see the kernel / common.h)
    static std::atomic<int> dispatch_id{0};
    mq.check_id = mq.nq == 8 && get_option(OPT_MLP_CHECK_INUSE) ? 1 + (dispatch_id.fetch_add(1, std::memory_order_relaxed) & 0x3FFFFFFF) : 0;
    hipLaunchKernelGGL((mlp_coop_kernel<FAMILY>), dim3((unsigned)grid), dim3(WAVES * 64), lds, st, mq);
    return check_launch("sad_mlp_chain_f32 (cooperative register-resident chain)");
}

int launch_coop(const RegMulti &mp, hipStream_t st) {
    const int fam = reg_family(mp.shape[0]);
    size_t lds = sizeof(float4) * RING_F4 + sizeof(float) * WAVES * pool_floats(fam);
    lds += 16;                                                  // the item indices handed to the workgroup
    for (int i = 0; i < mp.n; ++i) {
        lds += sizeof(float) * (size_t)(mp.c[i].np[0] + mp.c[i].np[1] + mp.c[i].np[2]);
        if (!coop_shape(mp.shape[i]) || mp.c[i].stream_off < 0 || reg_family(mp.shape[i]) != fam)
            return fail(SAD_EINVAL, "launch_coop: chain %d has no stream image or belongs to another shape family", i);
    }
    return fam == 1 ? launch_coop_family<1>(mp, lds, st) : launch_coop_family<2>(mp, lds, st);
}

}  // namespace sad

#ifdef SAD_COOP_STAMPS
extern "C" __attribute__((visibility("default"))) int sad_debug_read_coop_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_cstamps), sizeof(unsigned long long) * 64 * 16);
}
extern "C" __attribute__((visibility("default"))) int sad_debug_read_coop_all(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_call), sizeof(unsigned long long) * 2048 * 4);
}
#endif
