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
    #pragma unroll
                    for (int s = 0; s < KC; ++s)
                        x[s] = XCONT ? __builtin_bit_cast(bf16x8, max_pos_bf16x8(xq[i % DX].a[s], xq[i % DX].b[s]))
                                     : (XBF16 ? __builtin_bit_cast(bf16x8, xq[i % DX].a[s]) : cvt8(xq[i % DX].a[s], xq[i % DX].b[s]));
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
                    if (c + 1 < NC) store_w(wq[(i + 1) & 1], lds + ((i + 1) & 1) * STAGE_F4);      // (chunk c + 1: loaded during chunk c - 1)
                    __syncthreads();
                }
            }
        }
    }
    // ---- epilogue: lane = row, registers 4q .. 4q+3 = channels 32 t + 8 q + 4 h .. + 3 ----
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

}  // namespace

namespace sad {

int launch_bf16_rows(const BfRowsJob &job, hipStream_t st) {
    BfRowsJob jb = job;
    jb.nrb = (int)((jb.rows + 127) / 128);
    // (two tiles per item for layers with few rows — twice the workgroups — measured slower: cluster.agg 70 vs 50 us, the
    // weights and the rows are then read twice as often)
    const int nt = jb.ct > 2 ? 4 : 2;
    jb.ncb = (jb.ct + nt - 1) / nt;
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
