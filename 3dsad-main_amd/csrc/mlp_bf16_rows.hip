// One plain-row layer  out[row, :] = act(W . x[row, :] + b)  in bfloat16 on the gfx950 matrix cores: the stage
// aggregations of the bf16 mode (SPEC.md §14; sa1.agg 128 -> 64 on 131 072 rows ... cluster.agg 1536 -> 512 on 8 192 rows
// per 32-scene step).  No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
//
// These layers are memory-bound (their input is the f32 pooled output of the stage: 50 - 67 MB per step, 2 - 13 GFLOP), so the
// kernel is built around reading every input row exactly once per channel block and keeping many bytes in flight:
//   * work item = 128 rows x up to 128 output channels, four waves, wave w owns row tile w (32 rows) and ALL channel tiles
//     of the item: its x operand never goes through LDS — lane (r, h) loads the 8 consecutive k of its row straight from
//     memory (two 16-byte loads of f32, or one of bf16), rounds once and has the MFMA B fragment;
//   * the weight fragments of a k-chunk (64 k x 128 channels = 16 KB) are shared by the four waves through two LDS stages
//     (each wave fetches a quarter, one barrier per chunk);
//   * the next chunk's rows are in flight while the current one is multiplied;
//   * items that share a row block run on the same XCD back to back (channel block fastest within an XCD's slots), so the
//     2nd .. 4th read of a row block is an L2 hit;
//   * D[cout, row] = W . X^T: a lane holds 4 consecutive channels of its row per accumulator quad: 8-byte (bf16) or 16-byte
//     (f32) stores.
// Same arithmetic as mlp_bf16.hip (bf16 products, binary32 accumulation in the matrix core, order unspecified): SPEC §14 tolerance.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
using sad::BfRowsJob;

constexpr int KC = 4;                 // k-steps (of 16) per chunk
#ifndef SAD_ROWS_DX
#define SAD_ROWS_DX 2      // row-queue depth of layers with more than two chunks (3: slower on every aggregation, 224 - 256 registers)
#endif

__device__ __forceinline__ bf16x8 cvt8(const float4 a, const float4 b) {
    bf16x8 v;
    v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
    v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
    return v;
}

// max of two vectors of eight NON-NEGATIVE bf16 (pooled rows: post-ReLU maxima): their order is the order of their bit patterns as
// unsigned 16-bit integers — four v_pk_max_u16
__device__ __forceinline__ float4 max_pos_bf16x8(const float4 a, const float4 b) {
    return __builtin_bit_cast(float4, __builtin_elementwise_max(__builtin_bit_cast(u16x8, a), __builtin_bit_cast(u16x8, b)));
}

#ifdef SAD_ROWS_STAMPS   // measurement build only (tools/probe/rows_stamps.py): s_memtime sums per phase over a wave's chunks
__device__ unsigned long long g_rowst[1024 * 8];
#define SAD_RSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SAD_RACC(i, t1, t0) do { if (blockIdx.x < 1024 / (int)(blockDim.x / 64) && lane == 0) g_rowst[(blockIdx.x * (blockDim.x / 64) + wave) * 8 + (i)] += (t1) - (t0); } while (0)
#else
#define SAD_RSTAMP(var)
#define SAD_RACC(i, t1, t0)
#endif

// NT = channel tiles (of 32) per item: 4, or 2 for layers with at most 64 output channels
// XCONT (with XBF16): the rows are SPLIT-POOLED (common.h BfRowsJob; mlp_bf16_reg.hip): row g of the chain behind a column range is the
// maximum of x[g] and of the chain's continuation rows of the one or two tiles after the one the group's packed rows begin in.
// The lane finds them once (two table reads per chain), loads the first beside every chunk of its row — row 0, all zero, for the
// seven groups in eight that have none: no branch — and takes the maximum when the chunk is consumed; a second continuation row
// (a group of more than 32 rows across three tiles) is rare and read behind a branch.
template <bool XBF16, int NT, int DX, bool XCONT = false>
__global__ __launch_bounds__(256) void bf16_rows_kernel(const BfRowsJob jb) {
    static_assert(!XCONT || XBF16, "split-pooled rows are bf16");
    constexpr int STAGE_F4 = KC * NT * 64;
    __shared__ __attribute__((aligned(16))) float4 lds[2 * STAGE_F4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int KS = jb.ks;                         // k-steps of the layer (K padded to 16)
    const int NC = (KS + KC - 1) / KC;
    const int ncb = jb.ncb, nrb = jb.nrb;
    // XCD-aware item order: workgroup g runs on XCD g % 8; its slot g / 8 walks (row block, channel block) with the channel
    // block fastest, row blocks dealt round-robin to the XCDs
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    const int rb = xcd + 8 * (slot / ncb), cb = slot % ncb;
    if (rb >= nrb) return;
    const int nt = jb.ct - cb * NT < NT ? jb.ct - cb * NT : NT;        // channel tiles of this item (1..4)
#ifdef SAD_ROWS_STAMPS
    if (blockIdx.x < 256 && lane == 0)
        for (int i = 0; i < 8; ++i) g_rowst[(blockIdx.x * 4 + wave) * 8 + i] = 0;
#endif
    SAD_RSTAMP(tk0);
#ifdef SAD_ROWS_STAMPS
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const long long row = (long long)rb * 128 + wave * 32 + r;
    const bool live = row < jb.rows;
    const long long rowc = live ? row : jb.rows - 1;
    const float4 *wimg = reinterpret_cast<const float4 *>(jb.w) + (size_t)(cb * NT) * KS * 64;      // [tile][k-step][lane]
    const unsigned ulane = (unsigned)lane;

    // this lane's x: 8 consecutive k at k = 16 s + 8 h of its row
    const char *xrow = reinterpret_cast<const char *>(jb.x) + (size_t)rowc * jb.ldx * (XBF16 ? 2 : 4);
    // split-pooled rows: byte offsets of this row's first / second continuation row inside the buffer of chain i (0: the zero row)
    unsigned co1[SAD_MAX_RADII] = {}, co2[SAD_MAX_RADII] = {};
    if constexpr (XCONT) {
#pragma unroll
        for (int i = 0; i < SAD_MAX_RADII; ++i) {
            if (i < jb.n_pool) {
                const int r0 = jb.pool_gstart[i][rowc], r1 = jb.pool_gstart[i][rowc + 1];
                const int t0 = r0 >> 5, n = ((r1 - 1) >> 5) - t0;
                co1[i] = n >= 1 ? (unsigned)(t0 + 1) * (unsigned)jb.pool_ld[i] * 2u : 0u;
                co2[i] = n >= 2 ? (unsigned)(t0 + 2) * (unsigned)jb.pool_ld[i] * 2u : 0u;
            }
        }
    }
    struct XRaw { float4 a[KC], b[KC]; };
    auto load_x = [&](int c) -> XRaw {
        XRaw v;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            int ks = c * KC + s;
            ks = ks < KS ? ks : KS - 1;
            const int k = 16 * ks + 8 * h;
            const bool ok = k < jb.kin;            // (K is a multiple of 8: a chunk of 8 is all inside or all padding)
            const int kk = ok ? k : 0;
            if constexpr (XBF16) {
                v.a[s] = *reinterpret_cast<const float4 *>(xrow + (size_t)kk * 2);
                v.b[s] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!ok) v.a[s] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (XCONT) {
                    // the chain behind this k-step (column ranges are multiples of 16: wave-uniform)
                    const int k16 = 16 * ks;
                    const int ci = (k16 >= jb.pool_col0[1]) + (k16 >= jb.pool_col0[2]) + (k16 >= jb.pool_col0[3]);
                    const char *cb = reinterpret_cast<const char *>(ci == 0 ? jb.pool_cont[0] : (ci == 1 ? jb.pool_cont[1] : (ci == 2 ? jb.pool_cont[2] : jb.pool_cont[3])));
                    const unsigned o1 = ci == 0 ? co1[0] : (ci == 1 ? co1[1] : (ci == 2 ? co1[2] : co1[3]));
                    const unsigned o2 = ci == 0 ? co2[0] : (ci == 1 ? co2[1] : (ci == 2 ? co2[2] : co2[3]));
                    const unsigned kc = ok ? (unsigned)(k - (ci == 0 ? 0 : (ci == 1 ? jb.pool_col0[1] : (ci == 2 ? jb.pool_col0[2] : jb.pool_col0[3])))) * 2u : 0u;
                    v.b[s] = *reinterpret_cast<const float4 *>(cb + (size_t)(ok ? o1 : 0u) + kc);
                    if (ok && o2 != 0u) v.b[s] = max_pos_bf16x8(v.b[s], *reinterpret_cast<const float4 *>(cb + (size_t)o2 + kc));
                }
            } else {
                v.a[s] = *reinterpret_cast<const float4 *>(xrow + (size_t)kk * 4);
                v.b[s] = *reinterpret_cast<const float4 *>(xrow + (size_t)kk * 4 + 16);
                if (!ok) { v.a[s] = make_float4(0.f, 0.f, 0.f, 0.f); v.b[s] = v.a[s]; }
            }
        }
        return v;
    };
    // this wave's quarter of the weight fragments of chunk c: fragment f = wave * NT + i -> (k-step f / NT of the chunk, tile f % NT)
    struct WRaw { float4 f[NT]; };
    auto load_w = [&](int c) -> WRaw {
        WRaw v;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = wave * NT + i;
            int ks = c * KC + (f / NT);
            ks = ks < KS ? ks : KS - 1;
            const int t = (f % NT) < nt ? (f % NT) : 0;
            v.f[i] = (wimg + ((size_t)t * KS + ks) * 64)[ulane];
        }
        return v;
    };
    auto store_w = [&](const WRaw &v, float4 *st) {
#pragma unroll
        for (int i = 0; i < NT; ++i) st[(wave * NT + i) * 64 + lane] = v.f[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float *bias = jb.bias + (cb * NT + (t < nt ? t : 0)) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + 8 * q + 4 * h);
            acc[t][4 * q] = bv.x; acc[t][4 * q + 1] = bv.y; acc[t][4 * q + 2] = bv.z; acc[t][4 * q + 3] = bv.w;
        }
    }
    if constexpr (DX == 0) {
        // layers of one or two chunks (sa1.agg: K = 128, 1 024 workgroups): the next chunk in flight during the current one, nothing
        // else — the queued loop below costs such a layer 26 -> 32 us (six unrolled chunk bodies, two weight sets)
        XRaw xn = load_x(0);
        {
            const WRaw w0 = load_w(0);
            store_w(w0, lds);
        }
        __syncthreads();
    #pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            const float4 *cur = lds + (c & 1) * STAGE_F4;
            bf16x8 x[KC];
    #pragma unroll
            for (int s = 0; s < KC; ++s)
                x[s] = XCONT ? __builtin_bit_cast(bf16x8, max_pos_bf16x8(xn.a[s], xn.b[s])) : (XBF16 ? __builtin_bit_cast(bf16x8, xn.a[s]) : cvt8(xn.a[s], xn.b[s]));
            const int cn = c + 1 < NC ? c + 1 : c;
            xn = load_x(cn);                            // the next chunk's rows and weights are in flight during the MFMAs
            const WRaw wn = load_w(cn);
            // A full chunk of a full item (every chunk but possibly the last, every item but those of a ragged last channel block) runs
            // WITHOUT per-product conditions: with them every MFMA sat behind its own branch, LDS read and lgkmcnt(0) — ~200 cycles per
            // 32-cycle product, 3 000 cycles per chunk (cluster.agg: 24 chunks, 50 us for 5 us of MFMAs).  The fragments of k-step s + 1
            // are read while the products of k-step s run.
            if (nt == NT && (c + 1) * KC <= KS) {
                float4 wf[2][NT];
    #pragma unroll
                for (int t = 0; t < NT; ++t) wf[0][t] = cur[t * 64 + lane];
    #pragma unroll
                for (int s = 0; s < KC; ++s) {
                    if (s + 1 < KC) {
    #pragma unroll
                        for (int t = 0; t < NT; ++t) wf[(s + 1) & 1][t] = cur[((s + 1) * NT + t) * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);      // the next k-step's LDS reads stay AHEAD of this k-step's MFMAs
    #pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s & 1][t]), x[s], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
    #pragma unroll
                for (int s = 0; s < KC; ++s) {
                    if (c * KC + s < KS) {              // (wave-uniform: the padded k-steps of the last chunk are skipped)
    #pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (t < nt)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[(s * NT + t) * 64 + lane]), x[s], acc[t], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < NC) store_w(wn, lds + ((c + 1) & 1) * STAGE_F4);
            __syncthreads();
        }
    } else {
        // Queues in registers: the rows of chunks c + 1 .. c + DX and the weight fragments of chunks c + 1, c + 2 are in flight while chunk c
        // multiplies.  One chunk ahead (the first version) left every chunk waiting for loads issued 512 MFMA cycles earlier: 2 700 cycles
        // per chunk whatever the rows cost (cluster.agg, 24 chunks: 49 us with the rows, 31 with the rows loaded once).  Static register
        // names need the chunk loop unrolled by lcm(DX, 2) = 6.
        static_assert(DX > 0 && 6 % DX == 0, "row queue depth must divide the unroll factor");
        XRaw xq[DX];
        WRaw wq[2];
    #pragma unroll
        for (int d = 0; d < DX; ++d) xq[d] = load_x(d < NC ? d : NC - 1);
        {
            const WRaw w0 = load_w(0);
            wq[1] = load_w(NC > 1 ? 1 : 0);
            store_w(w0, lds);
        }
        __syncthreads();
    #pragma unroll 1
        for (int c0 = 0; c0 < NC; c0 += 6) {
    #pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int c = c0 + i;
                if (c < NC) {                               // (workgroup-uniform)
                    const float4 *cur = lds + (i & 1) * STAGE_F4;
                    bf16x8 x[KC];
                    SAD_RSTAMP(ta);
    #pragma unroll
                    for (int s = 0; s < KC; ++s)
                        x[s] = XCONT ? __builtin_bit_cast(bf16x8, max_pos_bf16x8(xq[i % DX].a[s], xq[i % DX].b[s]))
                                     : (XBF16 ? __builtin_bit_cast(bf16x8, xq[i % DX].a[s]) : cvt8(xq[i % DX].a[s], xq[i % DX].b[s]));
#ifdef SAD_ROWS_STAMPS
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#endif
                    SAD_RSTAMP(tb);
                    const int cx = c + DX < NC ? c + DX : NC - 1, cw = c + 2 < NC ? c + 2 : NC - 1;
    #if defined(SAD_ROWS_ABL) && SAD_ROWS_ABL == 1       // measurement builds: 1 the rows are loaded once, 2 the weights are loaded once (wrong results)
                    if (c == 0) xq[i % DX] = load_x(cx);
    #else
                    xq[i % DX] = load_x(cx);
    #endif
    #if defined(SAD_ROWS_ABL) && SAD_ROWS_ABL == 2
                    wq[i & 1] = load_w(0);
    #else
                    wq[i & 1] = load_w(cw);
    #endif
                    // A full chunk of a full item (every chunk but possibly the last, every item but those of a ragged last channel block) runs
                    // WITHOUT per-product conditions: with them every MFMA sat behind its own branch, LDS read and lgkmcnt(0) — ~200 cycles per
                    // 32-cycle product.  The fragments of k-step s + 1 are read while the products of k-step s run.
                    if (nt == NT && (c + 1) * KC <= KS) {
                        float4 wf[2][NT];
    #pragma unroll
                        for (int t = 0; t < NT; ++t) wf[0][t] = cur[t * 64 + lane];
    #pragma unroll
                        for (int s = 0; s < KC; ++s) {
                            if (s + 1 < KC) {
    #pragma unroll
                                for (int t = 0; t < NT; ++t) wf[(s + 1) & 1][t] = cur[((s + 1) * NT + t) * 64 + lane];
                            }
                            __builtin_amdgcn_sched_barrier(0);      // the next k-step's LDS reads stay AHEAD of this k-step's MFMAs
    #pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s & 1][t]), x[s], acc[t], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
    #pragma unroll
                        for (int s = 0; s < KC; ++s) {
                            if (c * KC + s < KS) {              // (wave-uniform: the padded k-steps of the last chunk are skipped)
    #pragma unroll
                                for (int t = 0; t < NT; ++t)
                                    if (t < nt)
                                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[(s * NT + t) * 64 + lane]), x[s], acc[t], 0, 0, 0);
                            }
                        }
                    }
#ifdef SAD_ROWS_STAMPS
    #pragma unroll
                    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
#endif
                    SAD_RSTAMP(tc);
                    if (c + 1 < NC) store_w(wq[(i + 1) & 1], lds + ((i + 1) & 1) * STAGE_F4);      // (chunk c + 1: loaded during chunk c - 1)
#ifdef SAD_ROWS_STAMPS
                    __builtin_amdgcn_s_waitcnt(0);
#endif
                    SAD_RSTAMP(td);
                    __syncthreads();
                    SAD_RSTAMP(te);
                    SAD_RACC(0, tb, ta); SAD_RACC(1, tc, tb); SAD_RACC(2, td, tc); SAD_RACC(3, te, td); SAD_RACC(4, 1, 0);
                }
            }
        }
    }
    // ---- epilogue: lane = row, registers 4q .. 4q+3 = channels 32 t + 8 q + 4 h .. + 3 ----
#ifdef SAD_ROWS_STAMPS
    { SAD_RSTAMP(tk1); SAD_RACC(5, tk1, tk0); SAD_RACC(6, __builtin_amdgcn_s_memrealtime(), tr0); }
#endif
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= nt) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = (cb * NT + t) * 32 + 8 * q + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = acc[t][4 * q + e];
                v[e] = jb.relu ? __builtin_amdgcn_fmed3f(a, 0.f, __builtin_inff()) : a;
            }
            if (jb.out_bf16) {
                __bf16 *o = reinterpret_cast<__bf16 *>(jb.out) + (size_t)row * jb.ld_out + jb.col_off + co;
                if (co + 3 < jb.cout && jb.vec_out) {
                    bf16x4 pk;
                    pk[0] = (__bf16)v[0]; pk[1] = (__bf16)v[1]; pk[2] = (__bf16)v[2]; pk[3] = (__bf16)v[3];
                    *reinterpret_cast<bf16x4 *>(o) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < jb.cout) o[e] = (__bf16)v[e];
                }
            } else {
                float *o = reinterpret_cast<float *>(jb.out) + (size_t)row * jb.ld_out + jb.col_off + co;
                if (co + 3 < jb.cout && jb.vec_out) {
                    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < jb.cout) o[e] = v[e];
                }
            }
        }
    }
}


// ---- second form (round 5): a tiled GEMM fed by LDS-DMA ------------------------------------------------------------------------------
// What the phase stamps of the kernel above said on the aggregation layers (tools/probe/rows_stamps.py, cluster.agg 8 192 x 1 536 -> 512):
// ~2 200 cycles per 64-deep chunk against 512 of MFMAs — the fragment-shaped row loads (every instruction touches 32 lines for 32 bytes
// each), the register-staged weights (four loads + four ds_write per wave and chunk, waited for in the middle of the chunk) and a
// write -> barrier -> read turn-around of the LDS at every chunk.  This form moves both operands global -> LDS with global_load_lds_dwordx4
// (no registers, no ds_write; whole 128-byte lines per row) into a ring of three stages, two chunks ahead of the one being multiplied:
//   * workgroup = 8 waves = 128 rows x (2 NTW) channel tiles; wave (rw, cw) owns row tile rw and channel tiles cw NTW .. cw NTW + NTW - 1;
//   * a stage holds the chunk's weight fragments (lane-linear, as packed) and the 128 x 64 row tile as 128-byte rows whose 16-byte columns
//     are XOR-ed with the row's low three bits (the source address carries the swizzle, the LDS image of a DMA is lane-linear), so the
//     B fragment of a k-step — lane (r, h) reads column 2 s + h of row r — is spread over the banks;
//   * split-pooled rows (XCONT): a second tile per stage holds, for every row, the chain's FIRST continuation row (row 0 of the
//     continuation buffer, all zero, for rows that have none): B = max(x, cont) as unsigned 16-bit pairs.  A second continuation row
//     (a group of more than 32 rows across three tiles: dense scenes) is read from memory behind a branch;
//   * the DMA is issued through inline asm (the compiler keeps no count of it: no wait of its own), retired by a counted vmcnt one
//     chunk later and published by a raw s_barrier — a stage is read in the chunk AFTER the barrier that follows its wait, and refilled
//     after the barrier that follows its last read (cdna_hip_programming.md, "Read a staged buffer one phase after the wait that retires it").
// Same products and the same k order per accumulator as the first form: bit-identical results (tests/test_gpu_bf16.py).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;       // (M0 is the compiler's: saved and restored in the statement that uses it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)));      // (lds_dst is wave-uniform)
    // (no "memory" clobber: with one, the by-value argument block is kept in scratch and re-read through it; the statements are volatile,
    // so they keep their order among themselves and relative to the barriers, which is all the ring needs)
}
// byte offset of the k-th continuation row (k = 1, 2) of group `row` in its chain's buffer (row stride ld elements), 0 = none: the group's
// packed rows begin in tile gstart[row] >> 5 and its continuation rows are those of the following tiles it reaches
__device__ __forceinline__ unsigned cont_off(const int *gstart, int ld, long long row, int k) {
    const int r0 = gstart[row], r1 = gstart[row + 1];
    const int t0 = r0 >> 5, n = ((r1 - 1) >> 5) - t0;
    return n >= k ? (unsigned)(t0 + k) * (unsigned)ld * 2u : 0u;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N)); }

template <int NTW, bool XCONT>
__global__ __launch_bounds__(512) void bf16_rows2_kernel(const BfRowsJob jb) {
    constexpr int TW = 2 * NTW;                       // channel tiles per workgroup
    constexpr int WB = KC * TW * 1024;                // bytes of a stage: weight fragments
    constexpr int XB = 128 * 128;                     //                   row tile (128 rows x 64 k of bf16)
    constexpr int SB = WB + XB + (XCONT ? XB : 0);    //                   + continuation tile
    constexpr int PW = KC * TW / 8;                   // weight pieces (1 KB) per wave and chunk
#if defined(SAD_ROWS2_WHATIF) && SAD_ROWS2_WHATIF == 1
    constexpr int PV = PW + (XCONT ? 2 : 0);
#elif defined(SAD_ROWS2_WHATIF) && SAD_ROWS2_WHATIF == 2
    constexpr int PV = 2 + (XCONT ? 2 : 0);
#elif defined(SAD_ROWS2_WHATIF) && SAD_ROWS2_WHATIF == 3
    constexpr int PV = (XCONT ? 2 : 0);
#else
    constexpr int PV = PW + 2 + (XCONT ? 2 : 0);      // DMA instructions per wave and chunk
#endif
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 3, cw = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int KS = jb.ks, NC = KS / KC;               // (host: KS % KC == 0, kin == 16 KS)
    const int ncb = jb.ncb, nrb = jb.nrb;
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    const int rb = xcd + 8 * (slot / ncb), cb = slot % ncb;
    if (rb >= nrb) return;                            // (workgroup-uniform)
#ifdef SAD_ROWS_STAMPS
    if (blockIdx.x < 128 && lane == 0)
        for (int i = 0; i < 8; ++i) g_rowst[(blockIdx.x * 8 + wave) * 8 + i] = 0;
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    SAD_RSTAMP(tk0);
    const unsigned lds0 = (unsigned)(size_t)smem2;
    // ---- DMA sources of this lane ------------------------------------------------------------------------------------------------
    const char *wsrc[PW];
    unsigned wdst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = wave + 8 * i, ks = p / TW, t = p % TW;
        int tile = cb * TW + t;
        tile = tile < jb.ct ? tile : jb.ct - 1;        // (a ragged last channel block: its surplus tiles repeat the last one and store nothing)
        wsrc[i] = reinterpret_cast<const char *>(jb.w) + (((size_t)tile * KS + ks) * 64 + lane) * 16;
        wdst[i] = p * 1024;
    }
    const char *xsrc[2];
    unsigned xdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave + 8 * i;                    // piece q: rows 8 q .. 8 q + 7 of the tile, one 128-byte line each
        long long row = (long long)rb * 128 + 8 * q + (lane >> 3);
        row = row < jb.rows ? row : jb.rows - 1;
        xsrc[i] = reinterpret_cast<const char *>(jb.x) + (size_t)row * jb.ldx * 2 + (((lane & 7) ^ (lane >> 3)) * 16);
        xdst[i] = WB + q * 1024;
    }
    // split-pooled rows: byte offset of the first continuation row of this lane's two DMA rows in the buffer of chain i (0: the zero
    // row), and of the SECOND continuation row of the row this lane multiplies (rare; 0: none)
    const int drow0 = 8 * wave + (lane >> 3), drow1 = drow0 + 64;              // the tile rows of this lane's two DMA pieces
    auto issue_wx = [&](int c) __attribute__((always_inline)) {   // the DMA of chunk c into stage c % 3: weight fragments and row tile
        const unsigned st = lds0 + (unsigned)(c % 3) * SB;
#if !defined(SAD_ROWS2_WHATIF) || (SAD_ROWS2_WHATIF != 2 && SAD_ROWS2_WHATIF != 3)      // (measurement builds: 1 = no row tile, 2 = no weights, 3 = neither — wrong results, what the bytes cost)
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(wsrc[i] + (size_t)c * (KC * 1024), st + wdst[i]);
#endif
#if !defined(SAD_ROWS2_WHATIF) || (SAD_ROWS2_WHATIF != 1 && SAD_ROWS2_WHATIF != 3)
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(xsrc[i] + (size_t)c * 128, st + xdst[i]);
#endif
    };
    // the first two chunks' weights and rows are on their way before anything below waits for memory (the continuation tables, the bias)
    issue_wx(0);
    if (NC > 1) issue_wx(1);
    // split-pooled rows: per chain and tile row, the byte offset of the FIRST continuation row in the chain's buffer (0: the zero row) and of
    // the SECOND (0: none; rare), the chains' buffers and first columns — tables in LDS behind the ring, read by chain index per chunk
    // (as selects over per-lane registers the compiler built these tables itself, in scratch)
    unsigned *t_co1 = reinterpret_cast<unsigned *>(smem2 + 3 * SB);          // [4][128]
    unsigned *t_co2 = t_co1 + 4 * 128;                                         // [4][128]
    unsigned long long *t_pc = reinterpret_cast<unsigned long long *>(t_co2 + 4 * 128);      // [4]
    int *t_k0 = reinterpret_cast<int *>(t_pc + 4);                             // [4]: first column of the chain
    int pk1 = 0x7fffffff, pk2 = 0x7fffffff, pk3 = 0x7fffffff;                  // first column of chains 1..3 (beyond every column when absent)
    if constexpr (XCONT) {
        const int np = jb.n_pool;
        if (np > 1) pk1 = jb.pool_col0[1];
        if (np > 2) pk2 = jb.pool_col0[2];
        if (np > 3) pk3 = jb.pool_col0[3];
        if (tid < 128) {
            long long row = (long long)rb * 128 + tid;
            row = row < jb.rows ? row : jb.rows - 1;
#define SAD_CONT_OF(I) do { t_co1[(I) * 128 + tid] = cont_off(jb.pool_gstart[I], jb.pool_ld[I], row, 1); \
                            t_co2[(I) * 128 + tid] = cont_off(jb.pool_gstart[I], jb.pool_ld[I], row, 2); } while (0)
            SAD_CONT_OF(0);
            if (np > 1) SAD_CONT_OF(1);
            if (np > 2) SAD_CONT_OF(2);
            if (np > 3) SAD_CONT_OF(3);
#undef SAD_CONT_OF
        }
        if (tid == 0) {
            t_pc[0] = reinterpret_cast<unsigned long long>(jb.pool_cont[0]);
            t_pc[1] = reinterpret_cast<unsigned long long>(jb.pool_cont[1]);
            t_pc[2] = reinterpret_cast<unsigned long long>(jb.pool_cont[2]);
            t_pc[3] = reinterpret_cast<unsigned long long>(jb.pool_cont[3]);
            t_k0[0] = 0; t_k0[1] = jb.pool_col0[1]; t_k0[2] = jb.pool_col0[2]; t_k0[3] = jb.pool_col0[3];
        }
        __syncthreads();
    }
    auto issue_c = [&](int c) __attribute__((always_inline)) {    // ... and the continuation tile (needs the tables)
        if constexpr (XCONT) {
            const unsigned st = lds0 + (unsigned)(c % 3) * SB;
            // the chain behind this lane's eight columns (column ranges are multiples of 16: a 16-byte piece lies behind one chain)
            const int kl = 64 * c + 8 * ((lane & 7) ^ (lane >> 3));
            const int ci = (kl >= pk1) + (kl >= pk2) + (kl >= pk3);
            const char *cbp = reinterpret_cast<const char *>(t_pc[ci]);
            const unsigned oa = t_co1[ci * 128 + drow0], ob = t_co1[ci * 128 + drow1];
            const unsigned kc = (unsigned)(kl - t_k0[ci]) * 2u;
            glds16(cbp + (size_t)oa + kc, st + xdst[0] + XB);
            glds16(cbp + (size_t)ob + kc, st + xdst[1] + XB);
        }
    };
    // ---- accumulators: bias ------------------------------------------------------------------------------------------------------
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int tile = cb * TW + cw * NTW + j;
        tile = tile < jb.ct ? tile : jb.ct - 1;
        const float *bias = jb.bias + tile * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + 8 * q + 4 * h);
            acc[j][4 * q] = bv.x; acc[j][4 * q + 1] = bv.y; acc[j][4 * q + 2] = bv.z; acc[j][4 * q + 3] = bv.w;
        }
    }
    // the bias reads are consumed HERE (the compiler puts its wait for them in front of this statement): left to their first real use, the
    // first MFMA of the loop body, that wait — a vmcnt(0) — runs in every chunk and drains the DMA issued two chunks ahead
#pragma unroll
    for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(acc[j]));
    // ---- the ring ------------------------------------------------------------------------------------------------------------------
    // (in issue order the wait below leaves only chunk 1's continuation pieces — or, without them, all of chunk 1 — in flight)
    issue_c(0);
    if (NC > 1) { issue_c(1); wait_vm<XCONT ? 2 : PV>(); } else wait_vm<0>();
    __syncthreads();                                   // (s_waitcnt lgkmcnt(0) + s_barrier: the compiler knows of no vector-memory operation in flight)
    const int xrow = rw * 32 + r;                       // this lane's row of the tile
    const unsigned xoff = WB + xrow * 128, xsw = (unsigned)(xrow & 7);
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        SAD_RSTAMP(ta);
        // (the two waves of a SIMD — w and w + 4 — taking turns, one issuing its DMA while the other multiplies, was measured: no faster, the
        // largest layers 7 - 12 % slower)
        if (c + 2 < NC) { issue_wx(c + 2); issue_c(c + 2); }
        SAD_RSTAMP(tb);
        const unsigned char *st = smem2 + (c % 3) * SB;
        // the chunk's four B fragments first (row r of the tile, columns 2 s + h), then the weight fragments one k-step ahead of their MFMAs
        float4 bf[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) bf[s] = *reinterpret_cast<const float4 *>(st + xoff + (((unsigned)(2 * s + h) ^ xsw) * 16));
        if constexpr (XCONT) {
            float4 bc[KC];
#pragma unroll
            for (int s = 0; s < KC; ++s) bc[s] = *reinterpret_cast<const float4 *>(st + xoff + XB + (((unsigned)(2 * s + h) ^ xsw) * 16));
            // a SECOND continuation row (a group of more than 32 rows across three tiles: dense scenes) comes straight from memory, rarely
            unsigned o2s[KC];
            const char *p2[KC];
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int ks = 64 * c + 16 * s;                                  // (wave-uniform: a k-step lies behind one chain)
                const int ci = (ks >= pk1) + (ks >= pk2) + (ks >= pk3);
                o2s[s] = t_co2[ci * 128 + xrow];
                p2[s] = reinterpret_cast<const char *>(t_pc[ci]) + (size_t)o2s[s] + (unsigned)(ks + 8 * h - t_k0[ci]) * 2u;      // (offset 0: the zero row)
            }
            if ((o2s[0] | o2s[1] | o2s[2] | o2s[3]) != 0u) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 t2[KC];
                // (loads and their wait in ONE asm statement: an ordinary load here makes the compiler wait for it — vmcnt(0), draining the
                // DMA two chunks ahead as well — at the join behind the branch, on the common path)
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                             "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(t2[0]), "=&v"(t2[1]), "=&v"(t2[2]), "=&v"(t2[3]) : "v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(p2[3]));
#pragma unroll
                for (int s = 0; s < KC; ++s) bc[s] = max_pos_bf16x8(bc[s], __builtin_bit_cast(float4, t2[s]));
            }
#pragma unroll
            for (int s = 0; s < KC; ++s) bf[s] = max_pos_bf16x8(bf[s], bc[s]);
        }
#ifdef SAD_ROWS_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)");
#endif
        SAD_RSTAMP(tc);
        float4 wf[2][NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) wf[0][j] = *reinterpret_cast<const float4 *>(st + ((cw * NTW + j) * 64 + lane) * 16);
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) wf[(s + 1) & 1][j] = *reinterpret_cast<const float4 *>(st + (((s + 1) * TW + cw * NTW + j) * 64 + lane) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);             // the next k-step's fragment reads stay AHEAD of this k-step's MFMAs
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s & 1][j]), __builtin_bit_cast(bf16x8, bf[s]), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SAD_ROWS_STAMPS
#pragma unroll
        for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(acc[j]));
#endif
        SAD_RSTAMP(td);
        if (c + 1 < NC) {
            // retire chunk c + 1's DMA (chunk c + 2's stays in flight), retire this chunk's LDS reads, publish
            if (c + 2 < NC) wait_vm<PV>(); else wait_vm<0>();
            SAD_RSTAMP(te);
            __syncthreads();
            SAD_RSTAMP(tf);
            SAD_RACC(0, tb, ta); SAD_RACC(1, tc, tb); SAD_RACC(2, td, tc); SAD_RACC(3, te, td); SAD_RACC(7, tf, te); SAD_RACC(4, 1, 0);
        }
    }
#ifdef SAD_ROWS_STAMPS
    { SAD_RSTAMP(tk1); SAD_RACC(5, tk1, tk0); SAD_RACC(6, __builtin_amdgcn_s_memrealtime(), tr0); }
#endif
    // ---- epilogue: lane = row, registers 4q .. 4q+3 = channels 32 t + 8 q + 4 h .. + 3 (as in the first form) ----
    const long long row = (long long)rb * 128 + rw * 32 + r;
    if (row >= jb.rows) return;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int tile = cb * TW + cw * NTW + j;
        if (tile >= jb.ct) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = tile * 32 + 8 * q + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = acc[j][4 * q + e];
                v[e] = jb.relu ? __builtin_amdgcn_fmed3f(a, 0.f, __builtin_inff()) : a;
            }
            if (jb.out_bf16) {
                __bf16 *o = reinterpret_cast<__bf16 *>(jb.out) + (size_t)row * jb.ld_out + jb.col_off + co;
                if (co + 3 < jb.cout && jb.vec_out) {
                    bf16x4 pk;
                    pk[0] = (__bf16)v[0]; pk[1] = (__bf16)v[1]; pk[2] = (__bf16)v[2]; pk[3] = (__bf16)v[3];
                    *reinterpret_cast<bf16x4 *>(o) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < jb.cout) o[e] = (__bf16)v[e];
                }
            } else {
                float *o = reinterpret_cast<float *>(jb.out) + (size_t)row * jb.ld_out + jb.col_off + co;
                if (co + 3 < jb.cout && jb.vec_out) {
                    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < jb.cout) o[e] = v[e];
                }
            }
        }
    }
}

}  // namespace

namespace sad {

int launch_bf16_rows(const BfRowsJob &job, hipStream_t st) {
    BfRowsJob jb = job;
    jb.nrb = (int)((jb.rows + 127) / 128);
    // (two tiles per item for layers with few rows — twice the workgroups — measured slower: cluster.agg 70 vs 50 us, the
    // weights and the rows are then read twice as often)
    const int nt = jb.ct > 2 ? 4 : 2;
    jb.ncb = (jb.ct + nt - 1) / nt;
    // second form (tiled GEMM fed by LDS-DMA): bf16 rows whose K is a whole number of 64-deep chunks, 16-byte aligned row starts
    if (get_option(OPT_MLP_ROWS_FORM) != 1 && jb.x_bf16 && jb.kin == jb.ks * 16 && jb.ks % KC == 0 && (jb.ldx * 2) % 16 == 0 &&
        (reinterpret_cast<uintptr_t>(jb.x) & 15) == 0) {
        {   // (split-pooled rows: column ranges are multiples of 16 — prepare_bf16 — so every 16-byte piece of a row lies behind one chain)
            const int ntw = jb.ct > 2 ? 2 : 1;
            jb.ncb = (jb.ct + 2 * ntw - 1) / (2 * ntw);
            const long long grid2 = 8LL * ((jb.nrb + 7) / 8) * jb.ncb;
            if (grid2 >= (1LL << 31)) return fail(SAD_EINVAL, "sad_mlp_chain_bf16: too many rows");
            const size_t lds = 3 * (size_t)(KC * 2 * ntw * 1024 + 128 * 128 * (jb.n_pool ? 2 : 1)) + (jb.n_pool ? 2 * 4 * 128 * 4 + 64 : 0);      // (+ the continuation tables)
#define SAD_ROWS2_LAUNCH(NTWV, XC) do { \
                static std::atomic<uint64_t> attr_done{0}; \
                lds_attr_once(attr_done, reinterpret_cast<const void *>(&bf16_rows2_kernel<NTWV, XC>), 160 * 1024); \
                hipLaunchKernelGGL((bf16_rows2_kernel<NTWV, XC>), dim3((unsigned)grid2), dim3(512), lds, st, jb); } while (0)
            if (ntw == 2) { if (jb.n_pool) SAD_ROWS2_LAUNCH(2, true); else SAD_ROWS2_LAUNCH(2, false); }
            else { if (jb.n_pool) SAD_ROWS2_LAUNCH(1, true); else SAD_ROWS2_LAUNCH(1, false); }
#undef SAD_ROWS2_LAUNCH
            return check_launch("sad_mlp_chain_bf16 (plain-row layer, second form)");
        }
    }
    const long long grid = 8LL * ((jb.nrb + 7) / 8) * jb.ncb;
    if (grid >= (1LL << 31)) return fail(SAD_EINVAL, "sad_mlp_chain_bf16: too many rows");
    // Which chunk loop.  The queued loop (rows and weights two chunks ahead, ~60 more registers) pays where nothing else hides the load
    // latency: more than two chunks AND about one workgroup per CU (32 KITTI-shaped scenes: cluster.agg / sa3.agg / sa2.agg launch 256
    // workgroups: 49 -> 42, 29 -> 27, 19.6 -> 18.9 us).  With many workgroups per CU the resident ones hide it and the registers cost
    // occupancy instead: sa1.agg (1 024 workgroups, two chunks) 26 -> 32 us, and at 32 nuScenes-shaped scenes (1 024 workgroups per layer)
    // cluster.agg 160 -> 180, sa3.agg 86 -> 100, sa2.agg 64 -> 69 us.
    const int cus = sad::device_cus();
#ifdef SAD_ROWS_DEEP_ALWAYS      // measurement builds: the queued loop for every layer of more than two chunks, whatever the grid
    const bool deep = (jb.ks + KC - 1) / KC > 2 && cus > 0;
#else
    const bool deep = (jb.ks + KC - 1) / KC > 2 && grid <= 2LL * cus;
#endif
#define SAD_ROWS_LAUNCH(XB, NTV, DXV) hipLaunchKernelGGL((bf16_rows_kernel<XB, NTV, DXV>), dim3((unsigned)grid), dim3(256), 0, st, jb)
    if (jb.n_pool) {
        if (!jb.x_bf16) return fail(SAD_EINVAL, "sad_mlp_chain_bf16: split-pooled rows are bf16");
#define SAD_ROWS_LAUNCH_C(NTV, DXV) hipLaunchKernelGGL((bf16_rows_kernel<true, NTV, DXV, true>), dim3((unsigned)grid), dim3(256), 0, st, jb)
        if (nt == 4) { if (deep) SAD_ROWS_LAUNCH_C(4, SAD_ROWS_DX); else SAD_ROWS_LAUNCH_C(4, 0); }
        else { if (deep) SAD_ROWS_LAUNCH_C(2, SAD_ROWS_DX); else SAD_ROWS_LAUNCH_C(2, 0); }
#undef SAD_ROWS_LAUNCH_C
    } else
    if (nt == 4) {
        if (jb.x_bf16) { if (deep) SAD_ROWS_LAUNCH(true, 4, SAD_ROWS_DX); else SAD_ROWS_LAUNCH(true, 4, 0); }
        else { if (deep) SAD_ROWS_LAUNCH(false, 4, SAD_ROWS_DX); else SAD_ROWS_LAUNCH(false, 4, 0); }
    } else {
        if (jb.x_bf16) { if (deep) SAD_ROWS_LAUNCH(true, 2, SAD_ROWS_DX); else SAD_ROWS_LAUNCH(true, 2, 0); }
        else { if (deep) SAD_ROWS_LAUNCH(false, 2, SAD_ROWS_DX); else SAD_ROWS_LAUNCH(false, 2, 0); }
    }
#undef SAD_ROWS_LAUNCH
    return check_launch("sad_mlp_chain_bf16 (plain-row layer)");
}

}  // namespace sad

#ifdef SAD_ROWS_STAMPS
extern "C" __attribute__((visibility("default"))) int sad_debug_read_rows_stamps(unsigned long long *dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rowst), sizeof(unsigned long long) * 1024 * 8);
}
#endif
