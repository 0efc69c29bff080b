// Grouped / plain point-feature MLP chain in bfloat16 on the gfx950 matrix cores (SPEC.md §14;
// BASELINE.json configs[4] "bf16, MFMA grouped-MLP path").  No reference source exists
// (/root/reference/README.md:1-2).
//
// One workgroup (4 waves) owns a tile of R rows (R = 128 / 64 / 32, the largest that fits LDS) and
// carries it through ALL layers of the chain: activations never leave LDS between layers.
//   * layer-0 input [feat(C) ‖ rel_xyz(3) ‖ 0-pad] is gathered straight into LDS as bf16 rows
//     (16-byte chunks of the point-major bf16 feature array; relative coordinates are formed in
//     binary32 and rounded once).  The weight columns are permuted at pack time to this order.
//   * a layer is a set of 32x32 output tiles, dealt round-robin to the waves with the row block
//     fastest, so the four waves work on the same weight tile at the same time (L1/L2 hits);
//     v_mfma_f32_32x32x16_bf16: the activation fragment is one ds_read_b128 per lane, the weight
//     fragment one global_load_dwordx4 per lane from a pre-packed fragment-order image.
//   * hidden layers run as  D[cout, row] = W·Xᵀ  — a lane then holds 4 consecutive output channels
//     of ONE row per accumulator quad, i.e. one 8-byte bf16 store into the next layer's row-major
//     LDS image;  the last grouped layer runs as  D[row, cout] = X·Wᵀ  — a lane holds 16 rows of
//     ONE channel, so the max-pool over a group is 15 v_max + one cross-half exchange, merged into
//     the zero-initialised output with an integer atomic max (values are >= +0 after the ReLU).
// Accumulation is binary32 inside the matrix core; SPEC §14 leaves its order free, so parity is a
// stated tolerance, not bit equality.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BF_T = 256;   // threads per workgroup (4 waves)

struct BfParams {
    const float *xyz, *new_xyz;
    const int32_t *idx;        // NULL = plain mode
    const void *feat;
    int feat_bf16;             // element type of feat: 1 = bf16, 0 = f32
    int ld_feat;               // row stride of feat in elements
    int N, M, S, C;
    long long rows;            // B*M*S (grouped) or B*M (plain)
    int L;
    int kp[SAD_MAX_LAYERS + 1];   // padded input width of layer l (kp[0] % 16 == 0, others % 32 == 0)
    int cout[SAD_MAX_LAYERS];     // true output width of layer l
    const bf16x8 *w[SAD_MAX_LAYERS];   // fragment image: [cout tile][k block][lane] x 8 bf16
    const float *bias[SAD_MAX_LAYERS]; // padded to a multiple of 32 with zeros
    int relu_mask;
    void *out;
    int out_bf16;              // plain mode only
    int ld_out, col_off;
    int R;                     // rows per tile
    int bufA_elems;            // bf16 elements of LDS buffer 0 (buffer 1 follows)
    int tiles;
    // packed mode (cnt given): only the leading cnt[g] rows of every group are computed (a row that
    // repeats the group's first neighbour cannot change the max).  rowtab from sad::launch_rowscan:
    // hdr[4] = {total rows, tiles, -, -}, row_start[ngroups + 1], tile_first[tiles]
    const int *rowtab;
    const int *row_src, *row_gid;   // row map of the packed order (sad::launch_rowscan with idx): source point, group
    int ngroups;
    int meta_off;              // byte offset of the per-tile row maps (s_pt[R], s_grp[R]) in LDS
    int noxcd;                 // A/B switch (option mlp_noxcd): plain tile stride
};

__device__ __forceinline__ void atomic_max_pos(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned *>(addr), __builtin_bit_cast(unsigned, v));
}

__device__ __forceinline__ __bf16 load_feat(const void *feat, int is_bf16, size_t i) {
    return is_bf16 ? reinterpret_cast<const __bf16 *>(feat)[i] : (__bf16) reinterpret_cast<const float *>(feat)[i];
}

// All output tiles of layer l for one row tile, in work units of RW row blocks x CW channel tiles
// dealt round-robin to the waves (an operand fragment then feeds RW or CW MFMAs).  An odd last
// block / tile is clamped onto its neighbour and its result dropped.  POOL: last layer of a grouped
// chain (D[row, cout] = X·Wᵀ, max over the group); else D[cout, row] = W·Xᵀ.
template <int RW, int CW, bool POOL>
__device__ __forceinline__ void run_units(const BfParams &p, int l, const __bf16 *X, __bf16 *Y, long long row0,
                                          int wave, int lane, int NRB, const int *s_grp) {
    const int ldx = p.kp[l] + 8, ldy = p.kp[l + 1] + 8;
    const int KB = p.kp[l] >> 4;
    const int CT = (p.cout[l] + 31) >> 5;
    const bool last = l == p.L - 1;
    const bool relu = (p.relu_mask >> l) & 1;
    const int NRP = (NRB + RW - 1) / RW, NCP = (CT + CW - 1) / CW;
    for (int tt = wave; tt < NRP * NCP; tt += BF_T / 64) {
        const int cp = tt / NRP, rp = tt - cp * NRP;
        int rb[RW], ct[CW];
        const bf16x8 *wp[CW];
        const __bf16 *xp[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            rb[r] = rp * RW + r < NRB ? rp * RW + r : rp * RW;
            xp[r] = X + (rb[r] * 32 + (lane & 31)) * ldx + (lane >> 5) * 8;
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            ct[c] = cp * CW + c < CT ? cp * CW + c : cp * CW;
            wp[c] = p.w[l] + (size_t)ct[c] * KB * 64 + lane;
        }
        f32x16 acc[RW][CW];
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float bv = POOL ? p.bias[l][ct[c] * 32 + (lane & 31)]
                                      : p.bias[l][ct[c] * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)];
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r][c][i] = bv;
            }
        // k loop: weight fragments run WD k-blocks ahead in a register ring (L2 latency), activation
        // fragments one k-block ahead (LDS latency)
        constexpr int WD = 4;
        bf16x8 wr[WD][CW], x[RW];
#pragma unroll
        for (int d = 0; d < WD; ++d)
#pragma unroll
            for (int c = 0; c < CW; ++c) wr[d][c] = wp[c][(d < KB ? d : KB - 1) * 64];
#pragma unroll
        for (int r = 0; r < RW; ++r) x[r] = *reinterpret_cast<const bf16x8 *>(xp[r]);
        for (int kb0 = 0; kb0 < KB; kb0 += WD) {
#pragma unroll
            for (int d = 0; d < WD; ++d) {
                const int kb = kb0 + d;
                if (kb < KB) {
                    const int kn = kb + 1 < KB ? kb + 1 : kb;
                    bf16x8 nx[RW];
#pragma unroll
                    for (int r = 0; r < RW; ++r) nx[r] = *reinterpret_cast<const bf16x8 *>(xp[r] + kn * 16);
#pragma unroll
                    for (int c = 0; c < CW; ++c)
#pragma unroll
                        for (int r = 0; r < RW; ++r)
                            acc[r][c] = POOL ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[r], wr[d][c], acc[r][c], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[d][c], x[r], acc[r][c], 0, 0, 0);
                    const int kw = kb + WD < KB ? kb + WD : KB - 1;
#pragma unroll
                    for (int c = 0; c < CW; ++c) wr[d][c] = wp[c][kw * 64];
#pragma unroll
                    for (int r = 0; r < RW; ++r) x[r] = nx[r];
                }
            }
        }
        if constexpr (POOL) {
            // acc[r][c][i] = y[row rb*32 + 8*(i/4) + 4*(lane/32) + i%4][cout ct*32 + lane%32]
            const int S = p.S;
            if (p.rowtab) {
                // packed rows: a group is a run of consecutive tile rows with arbitrary boundaries.
                // Each lane walks its 4 runs of 4 consecutive rows; a run inside one group is one atomic.
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (c > 0 && cp * CW + c >= CT) continue;
                    const int co = ct[c] * 32 + (lane & 31);
                    if (co >= p.cout[l]) continue;
                    float *orow = reinterpret_cast<float *>(p.out) + p.col_off + co;
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        if (r > 0 && rp * RW + r >= NRB) continue;
                        const f32x16 &av = acc[r][c];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int4 gq = *reinterpret_cast<const int4 *>(s_grp + rb[r] * 32 + 8 * q + 4 * (lane >> 5));
                            const float v0 = av[4 * q], v1 = av[4 * q + 1], v2 = av[4 * q + 2], v3 = av[4 * q + 3];
                            if (gq.x == gq.w) {           // (covers the all-invalid run: -1 == -1)
                                if (gq.x >= 0) {
                                    float m = v0 > v1 ? v0 : v1;
                                    const float m2 = v2 > v3 ? v2 : v3;
                                    m = m > m2 ? m : m2;
                                    atomic_max_pos(orow + (size_t)gq.x * p.ld_out, m > 0.f ? m : 0.f);
                                }
                            } else {
                                if (gq.x >= 0) atomic_max_pos(orow + (size_t)gq.x * p.ld_out, v0 > 0.f ? v0 : 0.f);
                                if (gq.y >= 0) atomic_max_pos(orow + (size_t)gq.y * p.ld_out, v1 > 0.f ? v1 : 0.f);
                                if (gq.z >= 0) atomic_max_pos(orow + (size_t)gq.z * p.ld_out, v2 > 0.f ? v2 : 0.f);
                                if (gq.w >= 0) atomic_max_pos(orow + (size_t)gq.w * p.ld_out, v3 > 0.f ? v3 : 0.f);
                            }
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if (c > 0 && cp * CW + c >= CT) continue;
                const int co = ct[c] * 32 + (lane & 31);
                const bool cok = co < p.cout[l];
                float *orow = reinterpret_cast<float *>(p.out) + p.col_off + co;
                if ((S & 31) == 0) {
                    float m[RW];
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        float v = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) v = acc[r][c][i] > v ? acc[r][c][i] : v;
                        const float o = __shfl_xor(v, 32, 64);
                        m[r] = o > v ? o : v;
                    }
                    if (lane < 32 && cok) {
                        const long long rbase = row0 + rb[0] * 32;
                        if (rbase < p.rows) {
                            if (S == 32) {          // a block is a whole group: plain stores
#pragma unroll
                                for (int r = 0; r < RW; ++r)
                                    if ((r == 0 || rp * RW + r < NRB) && rbase + r * 32 < p.rows)
                                        orow[(size_t)((rbase + r * 32) / 32) * p.ld_out] = m[r];
                            } else if (S == 64 && RW == 2 && rp * RW + 1 < NRB) {   // the unit is a whole group
                                orow[(size_t)(rbase / 64) * p.ld_out] = m[0] > m[RW - 1] ? m[0] : m[RW - 1];
                            } else {
#pragma unroll
                                for (int r = 0; r < RW; ++r)
                                    if ((r == 0 || rp * RW + r < NRB) && rbase + r * 32 < p.rows)
                                        atomic_max_pos(orow + (size_t)((rbase + r * 32) / S) * p.ld_out, m[r]);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        if (r > 0 && rp * RW + r >= NRB) continue;
                        const long long rbase = row0 + rb[r] * 32;
                        if (rbase >= p.rows) continue;
                        const f32x16 &av = acc[r][c];
                        if (S == 16) {
                            float m0 = 0.f, m1 = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) { m0 = av[i] > m0 ? av[i] : m0; m1 = av[8 + i] > m1 ? av[8 + i] : m1; }
                            const float o0 = __shfl_xor(m0, 32, 64), o1 = __shfl_xor(m1, 32, 64);
                            m0 = o0 > m0 ? o0 : m0;
                            m1 = o1 > m1 ? o1 : m1;
                            if (lane < 32 && cok) {
                                const size_t g0 = (size_t)(rbase / 16);
                                orow[g0 * p.ld_out] = m0;
                                if (rbase + 16 < p.rows) orow[(g0 + 1) * p.ld_out] = m1;
                            }
                        } else if (cok) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const long long rr = rbase + 8 * q + 4 * (lane >> 5);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (rr + i < p.rows) {
                                        const float v = av[4 * q + i] > 0.f ? av[4 * q + i] : 0.f;
                                        atomic_max_pos(orow + (size_t)((rr + i) / S) * p.ld_out, v);
                                    }
                            }
                        }
                    }
                }
            }
        } else {
            // acc[r][c][i] = y[row rb*32 + lane%32][cout ct*32 + 8*(i/4) + 4*(lane/32) + i%4]
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (r > 0 && rp * RW + r >= NRB) continue;
                const int row = rb[r] * 32 + (lane & 31);
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (c > 0 && cp * CW + c >= CT) continue;
                    const f32x16 &av = acc[r][c];
                    if (!last) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bf16x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float t = av[4 * q + i];
                                v[i] = (__bf16)(relu && !(t > 0.f) ? 0.f : t);
                            }
                            *reinterpret_cast<bf16x4 *>(Y + row * ldy + ct[c] * 32 + 8 * q + 4 * (lane >> 5)) = v;
                        }
                    } else if (row0 + row < p.rows) {
                        const size_t o = (size_t)(row0 + row) * p.ld_out + p.col_off;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int co = ct[c] * 32 + 8 * q + 4 * (lane >> 5) + i;
                                float t = av[4 * q + i];
                                t = relu && !(t > 0.f) ? 0.f : t;
                                if (co < p.cout[l]) {
                                    if (p.out_bf16) reinterpret_cast<__bf16 *>(p.out)[o + co] = (__bf16)t;
                                    else reinterpret_cast<float *>(p.out)[o + co] = t;
                                }
                            }
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void mlp_bf16_body(const BfParams &p, const int block, const int nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *const buf0 = reinterpret_cast<__bf16 *>(smem);
    __bf16 *const buf1 = buf0 + p.bufA_elems;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = p.R, NRB = R >> 5;
    const bool grouped = p.idx != nullptr;
    const int C = p.C;
    const int ld0 = p.kp[0] + 8;
    const bool vec = p.feat && (C & 7) == 0 && (p.ld_feat & 7) == 0;   // 16-byte (bf16) / 32-byte (f32) chunks

    int *const s_pt = reinterpret_cast<int *>(smem + p.meta_off);     // source row of tile row r: point b*N + j (grouped) or
                                                                      // plain row index; -1 = none
    int *const s_grp = s_pt + R;                                      // its group b*M + m (plain: the row), -1 = none
    const bool packed = p.rowtab != nullptr;
    const int ntiles = packed ? p.rowtab[1] : p.tiles;

    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin, so XCD x (workgroups with
    // L % 8 == x) strides through the x-th eighth of the tiles — its L2 serves one contiguous range of
    // scenes (points, features, indices) while concurrently running workgroups still work on
    // neighbouring tiles.  Needs a grid that is a multiple of 8 (otherwise the plain stride).
    const bool xcd = (nblocks & 7) == 0 && !p.noxcd;
    const int t8 = xcd ? (ntiles + 7) >> 3 : ntiles;                 // tiles per XCD range
    const int t_lo = xcd ? (block & 7) * t8 : 0;
    const int t_hi = t_lo + t8 < ntiles ? t_lo + t8 : ntiles;
    const int t_step = xcd ? nblocks >> 3 : nblocks;
    for (int tile = t_lo + (xcd ? block >> 3 : block); tile < t_hi; tile += t_step) {
        const long long row0 = (long long)tile * R;
        // ---- which source row does tile row r stand for? --------------------------------------
        if (packed) {        // two coalesced loads from the row map written by the scan kernels
            const int total = p.rowtab[0];
            for (int r = tid; r < R; r += BF_T) {
                const long long q = row0 + r;
                const bool ok = q < total;
                s_pt[r] = ok ? p.row_src[q] : -1;
                s_grp[r] = ok ? (p.row_gid[q] & 0x3fffffff) : -1;
            }
        } else {
            for (int r = tid; r < R; r += BF_T) {
                const long long row = row0 + r;
                const bool ok = row < p.rows;
                int pt = -1, g = -1;
                if (ok && grouped) {
                    g = (int)(row / p.S);
                    pt = (int)((long long)(g / p.M) * p.N + p.idx[row]);
                } else if (ok) {
                    pt = g = (int)row;
                }
                s_pt[r] = pt;
                s_grp[r] = g;
            }
        }
        __syncthreads();
        // ---- stage the layer-0 input rows ----------------------------------------------------
        if (vec) {
            const int C8 = C >> 3;
            for (int e = tid; e < R * C8; e += BF_T) {
                const int r = e / C8, c8 = e - r * C8;
                const int pt = s_pt[r];
                bf16x8 v = {};
                if (pt >= 0) {
                    const size_t src = (size_t)pt * p.ld_feat + c8 * 8;
                    if (p.feat_bf16) {
                        v = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(p.feat) + src);
                    } else {
                        const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.feat) + src);
                        const float4 b4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.feat) + src + 4);
                        v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
                        v[4] = (__bf16)b4.x; v[5] = (__bf16)b4.y; v[6] = (__bf16)b4.z; v[7] = (__bf16)b4.w;
                    }
                }
                *reinterpret_cast<bf16x8 *>(buf0 + r * ld0 + c8 * 8) = v;
            }
        }
        for (int r = tid; r < R; r += BF_T) {
            const int pt = s_pt[r];
            const bool ok = pt >= 0;
            __bf16 *x = buf0 + r * ld0;
            const size_t src = ok ? (size_t)pt * p.ld_feat : 0;
            float d[3] = {0.f, 0.f, 0.f};
            if (ok && grouped) {
                const float *q = p.xyz + (size_t)pt * 3;
                const float *cen = p.new_xyz + (size_t)s_grp[r] * 3;
                d[0] = q[0] - cen[0];
                d[1] = q[1] - cen[1];
                d[2] = q[2] - cen[2];
            }
            // everything the vector gather did not write: (features,) relative coordinates, zero pad —
            // assembled in registers, one 16-byte store per 8 channels
            const int cx = grouped ? C : p.kp[0];          // first channel of the coordinates (none in plain mode)
            for (int c8 = vec ? C >> 3 : 0; c8 < p.kp[0] >> 3; ++c8) {
                bf16x8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = c8 * 8 + i;
                    float t = 0.f;
                    if (c < C) t = ok ? (float)load_feat(p.feat, p.feat_bf16, src + c) : 0.f;
                    else if (c >= cx && c < cx + 3) t = d[c - cx];
                    v[i] = (__bf16)t;
                }
                *reinterpret_cast<bf16x8 *>(x + c8 * 8) = v;
            }
        }
        __syncthreads();

        // ---- layers ---------------------------------------------------------------------------
        for (int l = 0; l < p.L; ++l) {
            const __bf16 *X = (l & 1) ? buf1 : buf0;
            __bf16 *Y = (l & 1) ? buf0 : buf1;
            const int CT = (p.cout[l] + 31) >> 5;
            const bool pool = l == p.L - 1 && grouped;
            const int T = NRB * CT;
            const int RWs = (NRB >= 2 && T >= 8) ? 2 : 1;
            const int CWs = (CT >= 2 && (T >= 16 || (T >= 8 && NRB < 2))) ? 2 : 1;
            if (pool) {
                if (RWs == 2 && CWs == 2) run_units<2, 2, true>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else if (RWs == 2) run_units<2, 1, true>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else if (CWs == 2) run_units<1, 2, true>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else run_units<1, 1, true>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
            } else {
                if (RWs == 2 && CWs == 2) run_units<2, 2, false>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else if (RWs == 2) run_units<2, 1, false>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else if (CWs == 2) run_units<1, 2, false>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
                else run_units<1, 1, false>(p, l, X, Y, row0, wave, lane, NRB, s_grp);
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(BF_T) void mlp_bf16_kernel(const BfParams p) {
    mlp_bf16_body(p, blockIdx.x, gridDim.x);
}

// Several independent chains (the branches of one multi-radius stage) in one dispatch, heaviest
// first: the light chains fill the tail of the heavy one and the launch gaps disappear.
constexpr int BF_MULTI_MAX = 4;
struct BfMultiParams {
    BfParams p[BF_MULTI_MAX];
    int first[BF_MULTI_MAX + 1];   // first block of chain i; first[n] = grid size
    int n;
};

__global__ __launch_bounds__(BF_T) void mlp_bf16_multi_kernel(const BfMultiParams mp) {
    int c = 0;
    while (c + 1 < mp.n && (int)blockIdx.x >= mp.first[c + 1]) ++c;
    c = __builtin_amdgcn_readfirstlane(c);
    mlp_bf16_body(mp.p[c], blockIdx.x - mp.first[c], mp.first[c + 1] - mp.first[c]);
}

// W[l] f32 [cout][cin] row-major -> fragment image + padded bias.  Internal input order of layer 0
// in grouped mode is [feat(C) ‖ xyz(3)]: internal k < C reads W column 3 + k, k = C..C+2 column k - C.
__global__ void pack_bf16_kernel(const float *__restrict__ W, const float *__restrict__ bias, int cin, int cout,
                                 int kp, int xyz_first, __bf16 *__restrict__ wout, float *__restrict__ bout) {
    const int KB = kp >> 4, CT = (cout + 31) >> 5;
    const size_t total = (size_t)CT * KB * 64 * 8;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63;
        const size_t f = e >> 9;
        const int kb = f % KB, ct = f / KB;
        const int co = ct * 32 + (lane & 31);
        const int k = kb * 16 + (lane >> 5) * 8 + j;
        float v = 0.f;
        if (co < cout && k < cin) {
            int col = k;
            if (xyz_first) col = k < cin - 3 ? k + 3 : k - (cin - 3);
            v = W[(size_t)co * cin + col];
        }
        wout[e] = (__bf16)v;
    }
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < CT * 32; o += gridDim.x * blockDim.x)
        bout[o] = o < cout ? bias[o] : 0.f;
}

inline int kpad(int l, int width) { return l == 0 ? (width + 15) & ~15 : (width + 31) & ~31; }
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

// bytes of the per-layer images (fragment image + padded bias of every layer); the stream image of the register-resident
// chain kernel (mlp_bf16_reg.hip) follows them for grouped 3-layer chains with a compiled shape
static size_t layer_images_bytes(int L, const int *dims) {
    size_t n = 0;
    for (int l = 0; l < L; ++l) {
        const int CT = (dims[l + 1] + 31) / 32;
        n += align16((size_t)CT * 32 * kpad(l, dims[l]) * 2) + align16((size_t)CT * 32 * 4);
    }
    return n;
}

SAD_API size_t sad_mlp_packed_bytes_bf16(int L, const int *dims, int first_has_xyz) {
    if (L < 1 || L > SAD_MAX_LAYERS || !dims) return 0;
    size_t n = layer_images_bytes(L, dims);
    if (first_has_xyz) n += (size_t)sad::bfreg_stream_frags(sad::bfreg_shape_id(L, dims)) * 1024;
    return n;
}

SAD_API int sad_mlp_preferred_geometry_bf16(int L, const int *dims) {
    return sad::bfreg_shape_id(L, dims) >= 0 ? 2 : 0;
}

SAD_API int sad_mlp_pack_bf16(int L, const int *dims, int first_has_xyz, const float *const *W,
                              const float *const *bias, void *packed, sad_stream_t stream) {
    SAD_REQUIRE(L >= 1 && L <= SAD_MAX_LAYERS && dims && W && bias && packed, "sad_mlp_pack_bf16: bad argument");
    SAD_REQUIRE((uintptr_t)packed % 16 == 0, "sad_mlp_pack_bf16: packed must be 16-byte aligned");
    unsigned char *q = (unsigned char *)packed;
    for (int l = 0; l < L; ++l) {
        SAD_REQUIRE(dims[l] >= 1 && dims[l + 1] >= 1 && W[l] && bias[l], "sad_mlp_pack_bf16: bad layer %d", l);
        SAD_REQUIRE(!(l == 0 && first_has_xyz) || dims[0] >= 3, "sad_mlp_pack_bf16: first layer needs >= 3 inputs");
        const int CT = (dims[l + 1] + 31) / 32, kp = kpad(l, dims[l]);
        __bf16 *wout = (__bf16 *)q;
        q += align16((size_t)CT * 32 * kp * 2);
        float *bout = (float *)q;
        q += align16((size_t)CT * 32 * 4);
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W[l], bias[l], dims[l],
                           dims[l + 1], kp, (l == 0 && first_has_xyz) ? 1 : 0, wout, bout);
    }
    if (first_has_xyz) {
        const int shape = sad::bfreg_shape_id(L, dims);
        if (shape >= 0)
            if (int e = sad::bfreg_pack(shape, dims, 1, W, (unsigned char *)packed + layer_images_bytes(L, dims), (hipStream_t)stream)) return e;
    }
    return sad::check_launch("sad_mlp_pack_bf16");
}

namespace {
struct BfPrepared {
    BfParams p;
    size_t lds;
    int grid;
    bool rows;              // plain single layer on the row-streaming kernel (mlp_bf16_rows.hip); `rj` is filled, p is not
    sad::BfRowsJob rj;
    bool reg;               // geometry 2: register-resident chain (mlp_bf16_reg.hip); `rc` is filled, p is not
    sad::BfRegChain rc;
    int reg_shape;
    long long reg_tiles;
    sad::ScanJob scan;      // row-packing scan the chain needs before its kernel ...
    bool prescanned;        // ... unless the caller ran sad_mlp_rowscan on the workspace
};
}  // namespace

// Validation, tile size, row-packing scan of one chain; fills `q` for the launch.
static int prepare_bf16(const sad_mlp_bf16_args *a, sad_stream_t stream, BfPrepared &prep) {
    SAD_REQUIRE(a, "sad_mlp_chain_bf16: NULL args");
    SAD_REQUIRE(a->struct_size == sizeof(sad_mlp_bf16_args), "sad_mlp_chain_bf16: struct_size=%zu, this library's sad_mlp_bf16_args has %zu bytes "
                "(caller built against another sad_amd.h; ABI version %d)", a->struct_size, sizeof(sad_mlp_bf16_args), SAD_ABI_VERSION);
    SAD_REQUIRE(a->L >= 1 && a->L <= SAD_MAX_LAYERS && a->packed && a->out, "sad_mlp_chain_bf16: bad argument");
    SAD_REQUIRE((uintptr_t)a->packed % 16 == 0, "sad_mlp_chain_bf16: packed must be 16-byte aligned");
    SAD_REQUIRE(a->B >= 1 && a->M >= 1 && a->C >= 0, "sad_mlp_chain_bf16: bad sizes");
    const bool grouped = a->idx != nullptr;
    BfParams p{};
    if (grouped) {
        SAD_REQUIRE(a->xyz && a->new_xyz && a->N >= 1 && a->S >= 1, "sad_mlp_chain_bf16: grouped mode needs xyz, new_xyz, N, S");
        SAD_REQUIRE(a->dims[0] == a->C + 3, "sad_mlp_chain_bf16: dims[0]=%d != C+3=%d", a->dims[0], a->C + 3);
        SAD_REQUIRE((a->relu_mask & ((1 << a->L) - 1)) == (1 << a->L) - 1, "sad_mlp_chain_bf16: every grouped layer needs a ReLU");
        SAD_REQUIRE(!a->out_bf16 || (a->geometry == 2 && a->cont), "sad_mlp_chain_bf16: grouped output is f32 (atomic max merge), or bf16 with "
                    "continuation rows (out_bf16 = 1, cont != NULL: split pooling) on the register-resident chain (geometry 2)");
        SAD_REQUIRE(!a->n_pool, "sad_mlp_chain_bf16: n_pool describes the INPUT rows of a plain layer");
        p.rows = (long long)a->B * a->M * a->S;
    } else {
        SAD_REQUIRE(a->dims[0] == a->C && a->C >= 1 && a->S == 1, "sad_mlp_chain_bf16: plain mode needs dims[0] == C, S == 1");
        p.rows = (long long)a->B * a->M;
    }
    SAD_REQUIRE(a->C == 0 || a->feat, "sad_mlp_chain_bf16: NULL feat");
    SAD_REQUIRE(p.rows < (1LL << 31), "sad_mlp_chain_bf16: %lld rows (the row maps are 32-bit)", p.rows);
    if (a->feat && (a->C & 7) == 0 && (a->ld_feat & 7) == 0)
        SAD_REQUIRE((uintptr_t)a->feat % 16 == 0, "sad_mlp_chain_bf16: feat must be 16-byte aligned");
    p.xyz = a->xyz; p.new_xyz = a->new_xyz; p.idx = a->idx;
    p.feat = a->feat; p.feat_bf16 = a->feat_bf16; p.ld_feat = a->ld_feat;
    p.N = a->N; p.M = a->M; p.S = a->S; p.C = a->C;
    p.L = a->L; p.relu_mask = a->relu_mask;
    p.out = a->out; p.out_bf16 = a->out_bf16; p.ld_out = a->ld_out; p.col_off = a->col_off;
    prep.reg = false;
    prep.rows = false;
    prep.prescanned = a->prescanned != 0;
    if (!grouped && a->n_pool) {
        // the input rows are split-pooled outputs (bf16 rows + continuation rows of n_pool chains side by side): one layer, row-streaming kernel only
        SAD_REQUIRE(a->n_pool >= 1 && a->n_pool <= SAD_MAX_RADII, "sad_mlp_chain_bf16: n_pool must be 0..%d", SAD_MAX_RADII);
        if (a->L != 1 || (a->geometry != 0 && a->geometry != 3) || !a->feat_bf16)
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: split-pooled input rows (n_pool > 0) are read by the row-streaming layer only "
                                               "(one plain layer, geometry 0 or 3, bf16 rows)");
        int cols = 0;
        for (int i = 0; i < a->n_pool; ++i) {
            SAD_REQUIRE(a->pool_ws[i] && a->pool_cont[i] && (uintptr_t)a->pool_ws[i] % 16 == 0 && (uintptr_t)a->pool_cont[i] % 16 == 0,
                        "sad_mlp_chain_bf16: pool_ws[%d] / pool_cont[%d]: NULL or not 16-byte aligned", i, i);
            SAD_REQUIRE(a->pool_S[i] >= 1 && a->pool_S[i] <= 64 && a->pool_cols[i] >= 16 && a->pool_cols[i] % 16 == 0,
                        "sad_mlp_chain_bf16: pool_S[%d] must be 1..64 and pool_cols[%d] a multiple of 16", i, i);
            cols += a->pool_cols[i];
        }
        SAD_REQUIRE(cols == a->C, "sad_mlp_chain_bf16: the pool_cols add up to %d, the layer reads C = %d channels", cols, a->C);
        for (int i = 0; i < a->n_pool; ++i)     // (the layer addresses continuation rows with 32-bit byte offsets)
            SAD_REQUIRE(sad_mlp_cont_bytes(a->B, a->M, a->pool_S[i], a->pool_cols[i]) < (1ull << 32),
                        "sad_mlp_chain_bf16: continuation buffer %d of %zu bytes (limit 4 GB)", i, sad_mlp_cont_bytes(a->B, a->M, a->pool_S[i], a->pool_cols[i]));
    }
    if (!grouped && a->L == 1 && (a->geometry == 0 || a->geometry == 3)) {
        // ---- one plain layer: the row-streaming kernel (every input row read once per 128 output channels) ----
        const size_t esz = a->feat_bf16 ? 2 : 4;
        const bool ok = (a->C & 7) == 0 && ((size_t)a->ld_feat * esz) % 16 == 0 && (uintptr_t)a->feat % 16 == 0;
        if (ok) {
            sad::BfRowsJob &j = prep.rj;
            j = sad::BfRowsJob{};
            j.x = a->feat; j.x_bf16 = a->feat_bf16; j.ldx = a->ld_feat; j.kin = a->C; j.rows = p.rows;
            const int kp = kpad(0, a->dims[0]), CT = (a->dims[1] + 31) / 32;
            j.w = a->packed;
            j.bias = (const float *)((const unsigned char *)a->packed + align16((size_t)CT * 32 * kp * 2));
            j.ks = kp / 16; j.ct = CT; j.cout = a->dims[1];
            j.relu = a->relu_mask & 1;
            j.out = a->out; j.out_bf16 = a->out_bf16; j.ld_out = a->ld_out; j.col_off = a->col_off;
            const size_t osz = a->out_bf16 ? 2 : 4;
            j.vec_out = ((size_t)a->ld_out * osz) % (4 * osz) == 0 && ((size_t)a->col_off * osz) % (4 * osz) == 0 &&
                        (uintptr_t)a->out % (4 * osz) == 0;
            j.n_pool = a->n_pool;
            for (int i = 0, c0 = 0; i < a->n_pool; ++i) {
                j.pool_gstart[i] = (const int *)a->pool_ws[i] + sad::scan_gstart_off(p.rows, a->pool_S[i]);
                j.pool_cont[i] = a->pool_cont[i];
                j.pool_ld[i] = a->pool_cols[i];
                j.pool_col0[i] = c0;
                c0 += a->pool_cols[i];
                for (int k = i + 1; k <= SAD_MAX_RADII; ++k) j.pool_col0[k] = c0;
            }
            prep.rows = true;
            return SAD_OK;
        }
        if (a->n_pool) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: split-pooled input rows need C %% 8 == 0 and 16-byte aligned rows");
        if (a->geometry == 3) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: geometry 3 (row-streaming layer) needs C %% 8 == 0 and 16-byte aligned rows");
    }
    if (a->geometry == 2) {
        // ---- register-resident chain: one wave per 32-row tile, activations in registers, weights through an LDS ring ----
        const int shape = grouped ? sad::bfreg_shape_id(a->L, a->dims) : -1;
        const bool vec = a->feat_bf16 && (a->C & 7) == 0 && (a->ld_feat & 7) == 0 && a->C >= 8;
        if (shape < 0 || !a->cnt || !a->workspace || (a->dims[0] > 16 && !vec))
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: geometry 2 (register-resident chain) needs a compiled grouped 3-layer shape, "
                                               "cnt + workspace and, for more than 13 feature channels, 16-byte bf16 feature rows");
        SAD_REQUIRE((uintptr_t)a->workspace % 16 == 0, "sad_mlp_chain_bf16: workspace must be 16-byte aligned");
        SAD_REQUIRE(!vec || (uintptr_t)a->feat % 16 == 0, "sad_mlp_chain_bf16: feat must be 16-byte aligned");
        SAD_REQUIRE((long long)a->B * a->M < (long long)sad::CONT_BIT, "sad_mlp_chain_bf16: too many groups");
        SAD_REQUIRE((long long)a->B * a->N < (1LL << 31), "sad_mlp_chain_bf16: B*N too large");
        const int ngroups = a->B * a->M;
        int *tab = (int *)a->workspace;
        prep.scan = sad::make_scan_job(a->cnt, ngroups, a->S, 32, tab, 0, a->idx, a->N, a->M);
        if (a->out_bf16) {
            // split pooling: bf16 rows + continuation rows, plain stores only (nothing to zero; the table must come from a split scan:
            // this dispatch's own, or sad_mlp_rowscan_split)
            SAD_REQUIRE(a->dims[3] % 8 == 0 && a->ld_out % 8 == 0 && a->col_off % 8 == 0 && (uintptr_t)a->out % 16 == 0 && (uintptr_t)a->cont % 16 == 0,
                        "sad_mlp_chain_bf16: split pooling needs cout, ld_out and col_off multiples of 8 and 16-byte aligned out / cont");
            SAD_REQUIRE(sad_mlp_cont_bytes(a->B, a->M, a->S, a->dims[3]) < (1ull << 32), "sad_mlp_chain_bf16: continuation buffer of %zu bytes (limit 4 GB)",
                        sad_mlp_cont_bytes(a->B, a->M, a->S, a->dims[3]));
            prep.scan.split = 1;
            prep.scan.cont0 = a->cont;
            prep.scan.cont_cols = a->dims[3];
        } else {
            prep.scan.zout = (float *)a->out + a->col_off;     // (a scan launched by the dispatch itself zero-fills the groups that need it)
            prep.scan.zld = a->ld_out;
            prep.scan.zcols = a->dims[a->L];
        }
        sad::BfRegChain &rc = prep.rc;
        rc = sad::BfRegChain{};
        rc.xyz = a->xyz; rc.new_xyz = a->new_xyz; rc.feat = a->feat; rc.feat_bf16 = a->feat_bf16; rc.ld_feat = a->ld_feat; rc.C = a->C;
        const unsigned char *q = (const unsigned char *)a->packed;
        for (int l = 0; l < 3; ++l) {
            const int CT = (a->dims[l + 1] + 31) / 32;
            q += align16((size_t)CT * 32 * kpad(l, a->dims[l]) * 2);
            rc.bias[l] = (const float *)q;
            rc.np[l] = CT * 32;
            q += align16((size_t)CT * 32 * 4);
        }
        rc.stream = q;
        rc.out = (float *)a->out; rc.ld_out = a->ld_out; rc.col_off = a->col_off; rc.cout_last = a->dims[3];
        rc.rowtab = tab; rc.row_src = prep.scan.row_src; rc.row_gid = prep.scan.row_gid;
        rc.out_bf16 = a->out_bf16 ? 1 : 0; rc.cont = a->cont; rc.ld_cont = a->dims[3];
        prep.reg = true;
        prep.reg_shape = shape;
        prep.reg_tiles = ((long long)ngroups * a->S + 31) / 32;
        return SAD_OK;
    }
    const bool packed = grouped && a->cnt && a->workspace;
    if (packed) {
        SAD_REQUIRE((uintptr_t)a->workspace % 16 == 0, "sad_mlp_chain_bf16: workspace must be 16-byte aligned");
        SAD_REQUIRE((long long)a->B * a->M < (1LL << 30) && p.rows < (1LL << 31), "sad_mlp_chain_bf16: too many groups");
        p.ngroups = a->B * a->M;
    }
    const unsigned char *q = (const unsigned char *)a->packed;
    int ldA = 0, ldB = 0;
    for (int l = 0; l < a->L; ++l) {
        p.kp[l] = kpad(l, a->dims[l]);
        p.cout[l] = a->dims[l + 1];
        const int CT = (a->dims[l + 1] + 31) / 32;
        p.w[l] = (const bf16x8 *)q;
        q += align16((size_t)CT * 32 * p.kp[l] * 2);
        p.bias[l] = (const float *)q;
        q += align16((size_t)CT * 32 * 4);
        int &ld = (l & 1) ? ldB : ldA;
        ld = ld > p.kp[l] + 8 ? ld : p.kp[l] + 8;
    }
    p.kp[a->L] = kpad(a->L, a->dims[a->L]);
    const size_t budget = 150 * 1024;
    int R = 128;
    auto lds_of = [&](int r) { return (((size_t)r * 2 * (ldA + ldB) + 15) & ~(size_t)15) + (size_t)(2 * r + 4) * sizeof(int); };
    if (a->geometry) {     // forced rows per tile (autotuners): 32 / 64 / 128 / 256 (2 = the register-resident chain, 3 = the row-streaming layer, above)
        SAD_REQUIRE(a->geometry == 32 || a->geometry == 64 || a->geometry == 128 || a->geometry == 256,
                    "sad_mlp_chain_bf16: geometry (rows per tile) must be 32, 64, 128 or 256");
        R = a->geometry;
        if (lds_of(R) > 160 * 1024) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: %d rows per tile do not fit LDS", R);
    }
    while (!a->geometry && R > 32 && lds_of(R) > budget) R >>= 1;
    if (lds_of(R) > 160 * 1024) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_bf16: layer widths need %zu bytes of LDS", lds_of(R));
    while (!a->geometry && R > 32 && p.rows <= R / 2) R >>= 1;
    p.R = R;
    p.bufA_elems = R * ldA;
    p.meta_off = (int)(((size_t)R * 2 * (ldA + ldB) + 15) & ~(size_t)15);
    const long long tiles = (p.rows + R - 1) / R;       // packed mode: an upper bound, the kernel reads the real count
    if (packed) {
        p.rowtab = (const int *)a->workspace;
        if (int e = sad::launch_rowscan(a->cnt, p.ngroups, a->S, R, (int *)a->workspace, (hipStream_t)stream, 0,
                                        a->idx, a->N, a->M)) return e;
        p.row_src = p.rowtab + 4 + ((long long)p.ngroups + 1) + ((long long)p.ngroups * a->S / 32 + 2) + (p.ngroups / 1024 + 2);
        p.row_gid = p.row_src + (long long)p.ngroups * a->S;
    }
    SAD_REQUIRE(tiles < (1LL << 31), "sad_mlp_chain_bf16: too many rows");
    p.tiles = (int)tiles;
    const size_t lds_now = lds_of(R);
    int per_cu = (int)((160 * 1024) / (lds_now > 0 ? lds_now : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    p.noxcd = sad::get_option(sad::OPT_MLP_NOXCD);
    prep.p = p;
    prep.lds = lds_now;
    prep.grid = (int)(tiles < 256LL * per_cu ? tiles : 256LL * per_cu);
    return SAD_OK;
}

static void bf16_attrs() {
    static std::atomic<uint64_t> attr_done1{0}, attr_done2{0};
    sad::lds_attr_once(attr_done1, reinterpret_cast<const void *>(&mlp_bf16_kernel), 160 * 1024);
    sad::lds_attr_once(attr_done2, reinterpret_cast<const void *>(&mlp_bf16_multi_kernel), 160 * 1024);
}

// register-resident chains of one shape family (<= REG_MAX_CHAINS): pending scans in one pair of launches, then one dispatch
static int launch_bfreg_chains(const BfPrepared *const *qs, int n, hipStream_t st) {
    sad::ScanJob jobs[sad::REG_MAX_CHAINS];
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!qs[i]->prescanned) jobs[m++] = qs[i]->scan;
    if (m)
        if (int e = sad::launch_rowscan_multi(jobs, m, st)) return e;
    sad::BfRegMulti mp{};
    mp.n = n;
    for (int i = 0; i < n; ++i) {
        mp.c[i] = qs[i]->rc;
        mp.shape[i] = qs[i]->reg_shape;
        mp.max_tiles += qs[i]->reg_tiles;
    }
    return sad::launch_bfreg(mp, st);
}

SAD_API int sad_mlp_chain_bf16(const sad_mlp_bf16_args *a, sad_stream_t stream) {
    BfPrepared q;
    if (int e = prepare_bf16(a, stream, q)) return e;
    if (q.rows) return sad::launch_bf16_rows(q.rj, (hipStream_t)stream);
    if (q.reg) {
        const BfPrepared *one = &q;
        return launch_bfreg_chains(&one, 1, (hipStream_t)stream);
    }
    bf16_attrs();
    hipLaunchKernelGGL(mlp_bf16_kernel, dim3(q.grid), dim3(BF_T), q.lds, (hipStream_t)stream, q.p);
    return sad::check_launch("sad_mlp_chain_bf16");
}

SAD_API int sad_mlp_chain_multi_bf16(const sad_mlp_bf16_args *const *args, int n, sad_stream_t stream) {
    SAD_REQUIRE(args && n >= 1, "sad_mlp_chain_multi_bf16: need at least one chain");
    if (n > BF_MULTI_MAX) {
        for (int i = 0; i < n; i += BF_MULTI_MAX)
            if (int e = sad_mlp_chain_multi_bf16(args + i, n - i < BF_MULTI_MAX ? n - i : BF_MULTI_MAX, stream)) return e;
        return SAD_OK;
    }
    BfPrepared q[BF_MULTI_MAX];
    for (int i = 0; i < n; ++i)
        if (int e = prepare_bf16(args[i], stream, q[i])) return e;
    {   // register-resident chains: one dispatch per shape family, heaviest chain first
        bool any_reg = false;
        for (int i = 0; i < n; ++i) any_reg = any_reg || q[i].reg || q[i].rows;
        if (any_reg) {
            bool done[BF_MULTI_MAX] = {};
            for (int i = 0; i < n; ++i) {
                if (done[i]) continue;
                if (q[i].rows) {
                    if (int e = sad::launch_bf16_rows(q[i].rj, (hipStream_t)stream)) return e;
                    done[i] = true;
                    continue;
                }
                if (!q[i].reg) {
                    bf16_attrs();
                    hipLaunchKernelGGL(mlp_bf16_kernel, dim3(q[i].grid), dim3(BF_T), q[i].lds, (hipStream_t)stream, q[i].p);
                    if (int e = sad::check_launch("sad_mlp_chain_bf16")) return e;
                    done[i] = true;
                    continue;
                }
                const BfPrepared *ord[sad::REG_MAX_CHAINS];
                int m = 0;
                for (int k = i; k < n && m < sad::REG_MAX_CHAINS; ++k)
                    if (!done[k] && q[k].reg && sad::bfreg_family(q[k].reg_shape) == sad::bfreg_family(q[i].reg_shape)) { ord[m++] = &q[k]; done[k] = true; }
                auto heavy = [](const BfPrepared *s) { return (double)sad::bfreg_stream_frags(s->reg_shape) * (double)s->reg_tiles; };
                for (int x = 0; x < m; ++x)
                    for (int y = x + 1; y < m; ++y)
                        if (heavy(ord[y]) > heavy(ord[x])) { const BfPrepared *t = ord[x]; ord[x] = ord[y]; ord[y] = t; }
                if (int e = launch_bfreg_chains(ord, m, (hipStream_t)stream)) return e;
            }
            return SAD_OK;
        }
    }
    bf16_attrs();
    if (n == 1) {
        hipLaunchKernelGGL(mlp_bf16_kernel, dim3(q[0].grid), dim3(BF_T), q[0].lds, (hipStream_t)stream, q[0].p);
        return sad::check_launch("sad_mlp_chain_bf16");
    }
    int order[BF_MULTI_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    auto weight = [&](int i) {
        double m = 0;
        for (int l = 0; l < q[i].p.L; ++l) m += (double)q[i].p.kp[l] * q[i].p.cout[l];
        return m * (double)q[i].p.rows;
    };
    for (int i = 0; i < n; ++i)
        for (int k = i + 1; k < n; ++k)
            if (weight(order[k]) > weight(order[i])) { const int t = order[i]; order[i] = order[k]; order[k] = t; }
    BfMultiParams mp{};
    mp.n = n;
    size_t lds = 0;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        mp.p[i] = q[order[i]].p;
        mp.first[i] = (int)total;
        total += q[order[i]].grid;
        lds = q[order[i]].lds > lds ? q[order[i]].lds : lds;
    }
    mp.first[n] = (int)total;
    hipLaunchKernelGGL(mlp_bf16_multi_kernel, dim3((unsigned)total), dim3(BF_T), lds, (hipStream_t)stream, mp);
    return sad::check_launch("sad_mlp_chain_multi_bf16");
}
