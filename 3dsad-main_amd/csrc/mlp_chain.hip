// Fused group -> shared-MLP (1x1 conv + bias + ReLU chain) -> max-pool for gfx950 (SPEC.md §6).
// No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
//
// Formulation.  Every layer is computed TRANSPOSED on v_mfma_f32_32x32x2_f32:
//     Y^T[oc, row] = W[oc, k] * X^T[k, row]        A = weights, B = activations, D = Y^T tile
// so the MFMA output has the sample row on the lane (col = lane&31) and the output channels in the
// 16 accumulator registers — exactly the [k][row] image the next layer reads as its B operand.
// Activations of a tile of R = 32*RW*WM rows therefore stay in LDS as a [channel][row] image for the
// whole chain (never in HBM); weights are pre-packed in A-fragment order (one coalesced 16-B load
// per lane feeds four MFMAs) and stream from L2 straight into VGPRs through a 4-deep register ring.
// The accumulator starts at the bias and the k-loop ascends, so each output is the fmaf chain of
// SPEC.md §6 bit for bit (the MFMA is an exact k-ordered f32 fma chain on gfx950).
//
// Work split inside a workgroup of W waves: WN waves along the 32-channel output tiles, WM = W/WN
// along the row tiles (RW row tiles of 32 rows per wave).  Narrow layers (SA1: one output tile) use
// WN = 1 so every wave owns its own rows; wide layers use WN = W so all waves share one row tile
// and the LDS image stays small.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SAD_MLP_BDEPTH
#define SAD_MLP_BDEPTH 2
#endif

constexpr int MAXL = SAD_MAX_LAYERS;

struct MlpParams {
    const float *xyz;
    const float *new_xyz;
    const int32_t *idx;
    const int32_t *cnt;    // optional per-group row counts
    const float *feat;
    const float *packed;
    float *out;
    long long total_rows;  // B*M*Sp
    int ld_feat, N, M, S, C;
    int sp_shift;          // Sp = 1 << sp_shift rows per pooling group (Sp >= S)
    int grouped;           // idx != NULL
    int L;
    int kp[MAXL];          // padded input channels of layer l (multiple of 8)
    int np[MAXL];          // padded output channels of layer l (multiple of 32)
    int cout_last;
    long long off[MAXL];   // float offset of layer l inside `packed`
    int relu_mask;
    int ld_out, col_off;
    int wn_shift;          // WN = 1 << wn_shift
    int xcd_nb;            // > 0 (static packing / plain rows): number of work chunks; workgroup L takes chunk
                           //   (L % 8) * ceil(nb / 8) + L / 8, so the chunks an XCD works on are one contiguous
                           //   range of scenes and its L2 holds only their points / features / indices
    int flex;              // 1: (output tile, row tile) items of a layer are dealt round-robin to ALL waves
                           //    (RW == 1): no wave idles in a layer with fewer than WN output tiles
    int kc;                // layer-0 k-chunk (multiple of 8); == kp[0] when the whole input fits
    int bufA_rows, bufB_rows;
    int cpr, cshift;       // float4 chunks per feature row (0 = scalar path), log2 of lanes per row
    int vec_out;           // 16-B output stores allowed
    int bias_total;        // sum of np[l]: biases are copied to LDS once per workgroup
    int G;                 // grouped mode: (b,m) groups per workgroup
    int nodedup;           // tuning/A-B switch: compute the padded duplicate rows too
    int s_off_entries;     // capacity of s_off (the work-counter broadcast slot follows it)
    int *rowtab;           // global row packing (see rowscan_kernel): hdr[4], row_start[ngroups+1], pass_first[]
    const int *row_src, *row_gid;   // row map of the packed order (see RowMap)
    long long total_groups; // B*M
};

// LDS activation image of a tile of R rows.  Channels are grouped in k-blocks of 8; inside a block
// the even channels (plane 0) and the odd channels (plane 1) are separate [row][4] arrays, so lane
// (j,h) of a wave fetches the four B operands of a k-step (k = 8t + 2e + h, e = 0..3) for its row
// with ONE conflict-free ds_read_b128:   float index = (((c>>3)*2 + (c&1))*R + row)*4 + ((c&7)>>1)
// Each plane is PS = 4R + 8 floats long: the 8-float pad staggers the planes over the LDS banks so
// the staging stores of one row (16 lanes, 16 different planes) do not all hit one bank.
__device__ __forceinline__ int act_idx(int c, int r, int PS) {
    return (((c >> 3) << 1) + (c & 1)) * PS + r * 4 + ((c & 7) >> 1);
}

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
    return o > v ? o : v;
}

// max over aligned groups of 2^steps lanes (steps <= 5), every lane of a group gets the result
__device__ __forceinline__ float group_max(float v, int steps) {
    if (steps > 0) v = dpp_max<0xB1>(v);   // quad_perm [1,0,3,2]
    if (steps > 1) v = dpp_max<0x4E>(v);   // quad_perm [2,3,0,1]
    if (steps > 2) v = dpp_max<0x141>(v);  // row_half_mirror
    if (steps > 3) v = dpp_max<0x140>(v);  // row_mirror
    if (steps > 4) {
        const float o = __shfl_xor(v, 16, 64);
        v = o > v ? o : v;
    }
    return v;
}

// The k-loop of one output tile: acc[rt] += W[oc tile, k0 : k0 + 8*n4] * X[k, rows of tile rt].
// af: this lane's A fragments (one float4 = 4 k-pairs per k-step, 64 float4 apart per k-step),
// bp: this lane's B operands (one float4 per k-step, kbs float4 apart per k-step, 32 per row tile).
// Software pipeline written out by hand (hipcc otherwise sinks each load to its use): a ring of
// four A registers refilled right after use (L2 latency covered by three k-steps of MFMAs) and two
// B registers (LDS latency covered by one k-step).
#ifndef SAD_MLP_ADEPTH
#define SAD_MLP_ADEPTH 4   // k-steps of weight fragments in flight per accumulator chain
#endif

template <int RW>
struct BFrag { float4 v[RW]; };

template <int RW>
__device__ __forceinline__ void mma4(f32x16 (&acc)[RW], const float4 a, const BFrag<RW> &b) {
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.v[rt].x, acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.v[rt].y, acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.v[rt].z, acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.v[rt].w, acc[rt], 0, 0, 0);
}

template <int RW>
__device__ __forceinline__ BFrag<RW> ldb(const float4 *__restrict__ bp, int t, int n4, int kbs) {
    BFrag<RW> b;
    const int tc = t < n4 ? t : n4 - 1;
#pragma unroll
    for (int rt = 0; rt < RW; ++rt) b.v[rt] = bp[tc * kbs + rt * 32];
    return b;
}

__device__ __forceinline__ float4 lda(const float4 *__restrict__ af, int t, int n4) {
    return af[(t < n4 ? t : n4 - 1) * 64];
}

template <int RW, int D>
__device__ __forceinline__ void mma_ktile(f32x16 (&acc)[RW], const float4 *__restrict__ af, int n4,
                                          const float4 *__restrict__ bp, int kbs) {
    // ring of D A registers (refilled right behind the MFMAs that consumed them: L2 latency is
    // covered by D-1 k-steps of MFMAs) and two B registers (LDS latency: one k-step)
    float4 a[D];
#pragma unroll
    for (int u = 0; u < D; ++u) a[u] = lda(af, u, n4);
    constexpr int BD = SAD_MLP_BDEPTH <= 2 || RW > 2 ? 2 : 4;   // B ring depth (LDS prefetch distance)
    BFrag<RW> b[BD];
#pragma unroll
    for (int u = 0; u < BD; ++u) b[u] = ldb<RW>(bp, u, n4, kbs);
    int t = 0;
    if (n4 >= D) {
        // Opaque touch: keeps InstCombine from folding the loop PHIs of loads into "load at use".
#pragma unroll
        for (int u = 0; u < D; ++u) asm volatile("" : "+v"(a[u].x));
#pragma unroll
        for (int u = 0; u < BD; ++u)
#pragma unroll
            for (int rt = 0; rt < RW; ++rt) asm volatile("" : "+v"(b[u].v[rt].x));
        for (; t + D <= n4; t += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                mma4<RW>(acc, a[u], b[u % BD]);
                a[u] = lda(af, t + D + u, n4);
                b[u % BD] = ldb<RW>(bp, t + u + BD, n4, kbs);
                // the refill must stay right behind the MFMAs that consumed the register (the
                // machine scheduler otherwise clusters all refills at the loop end)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const int rem = n4 - t;   // 0..D-1 k-steps left; a[0..rem) and b[0..BD) already hold them
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
        if (u < rem) {
            mma4<RW>(acc, a[u], b[u % BD]);
            if (u + BD < rem) b[u % BD] = ldb<RW>(bp, t + u + BD, n4, kbs);
        }
    }
}

// Two output tiles per wave (RW = 1): two independent accumulator chains share every B fragment and
// alternate on the matrix pipe, so a wave keeps issuing MFMAs while one chain's operands are late —
// worth it where LDS size leaves one 8-wave workgroup (2 waves per SIMD) on a CU.
__device__ __forceinline__ void mma4x2(f32x16 &acc0, f32x16 &acc1, const float4 a0, const float4 a1, const float4 b) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
}

template <int D>
__device__ __forceinline__ void mma_ktile2(f32x16 &acc0, f32x16 &acc1, const float4 *__restrict__ af0,
                                           const float4 *__restrict__ af1, int n4,
                                           const float4 *__restrict__ bp, int kbs) {
    float4 a0[D], a1[D], b[2];
#pragma unroll
    for (int u = 0; u < D; ++u) { a0[u] = lda(af0, u, n4); a1[u] = lda(af1, u, n4); }
#pragma unroll
    for (int u = 0; u < 2; ++u) b[u] = bp[(u < n4 ? u : n4 - 1) * kbs];
    int t = 0;
    if (n4 >= D) {
#pragma unroll
        for (int u = 0; u < D; ++u) { asm volatile("" : "+v"(a0[u].x)); asm volatile("" : "+v"(a1[u].x)); }
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(b[u].x));
        for (; t + D <= n4; t += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                mma4x2(acc0, acc1, a0[u], a1[u], b[u % 2]);
                a0[u] = lda(af0, t + D + u, n4);
                a1[u] = lda(af1, t + D + u, n4);
                const int tb = t + u + 2;
                b[u % 2] = bp[(tb < n4 ? tb : n4 - 1) * kbs];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const int rem = n4 - t;
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
        if (u < rem) {
            mma4x2(acc0, acc1, a0[u], a1[u], b[u % 2]);
            if (u + 2 < rem) b[u % 2] = bp[(t + u + 2) * kbs];
        }
    }
}

// ---- global row packing ------------------------------------------------------------------------
// With per-group counts from the ball query, the surviving rows of ALL groups of a launch are numbered
// consecutively (prefix sum of the counts), so every pass / tile of R rows is full.  The table (ints):
// hdr[0] = total rows, hdr[1] = passes of R rows, hdr[2] = 0 (work counter), then (unused, kept for the
// layout) row_start / pass_first areas, block sums, and the ROW MAP: for every packed row its source
// point b*N + idx[g*S + s] and its group g (bit 30 set when the group lies inside one 32-row tile of
// the packed order) — the MLP kernels find the rows of a tile with two coalesced loads.
// Two launches serve up to three chains at once (the branches of a stage): block sums, then every
// block adds the sums of the blocks before it to its own scan and writes its part of the row map
// COOPERATIVELY by destination row (coalesced; each row finds its group by a binary search of the
// block's offsets in LDS) — a thread walking its own group's rows wrote 4 bytes per lane per step at
// scattered addresses and took 22 us for 16 384 groups; this takes ~4.
constexpr int SCAN_T = 1024;
constexpr int WHOLE_BIT = 1 << 30;

__device__ __forceinline__ int scan_job_of(const sad::ScanMulti &sm, int block, int &local) {
    int ji = 0;
    while (ji + 1 < sm.n && block >= sm.j[ji + 1].blk0) ++ji;
    local = block - sm.j[ji].blk0;
    return ji;
}

__device__ __forceinline__ int clamp_cnt(const sad::ScanJob &jb, int g) {
    int c = jb.cnt[g];
    c = c < 1 ? 1 : (c > jb.S ? jb.S : c);
    return jb.nodedup ? jb.S : c;
}

__global__ __launch_bounds__(SCAN_T) void rowscan_sums_kernel(const sad::ScanMulti sm) {
    __shared__ int wsum[16];
    int lb;
    const sad::ScanJob &jb = sm.j[scan_job_of(sm, blockIdx.x, lb)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lb * SCAN_T + tid;
    int c = g < jb.ngroups ? clamp_cnt(jb, g) : 0;
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        jb.blk_sum[lb] = t;
    }
}

__global__ __launch_bounds__(SCAN_T) void rowscan_write_kernel(const sad::ScanMulti sm) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    __shared__ int s_start[SCAN_T + 1];          // row offsets of this block's groups, relative to s_base
    int lb;
    const sad::ScanJob &jb = sm.j[scan_job_of(sm, blockIdx.x, lb)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int part = 0;
    for (int b = tid; b < lb; b += SCAN_T) part += jb.blk_sum[b];
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        s_base = t;
    }
    __syncthreads();
    const int base = s_base;
    const int g = lb * SCAN_T + tid;
    const int c = g < jb.ngroups ? clamp_cnt(jb, g) : 0;
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - c;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    s_start[tid] = run;
    if (tid == SCAN_T - 1) s_start[SCAN_T] = run + c;
    __syncthreads();
    const int blk_rows = s_start[SCAN_T];
    if (g == jb.ngroups - 1) {
        const int total = base + run + c;
        jb.tab[0] = total;
        jb.tab[1] = (total + jb.R - 1) / jb.R;
        jb.tab[2] = 0;                           // item queues of the kernels that deal items dynamically (common.h, ItemQueue;
        jb.tab[3] = 0;                           // they re-arm them when they finish): single counter, finished workgroups,
        if (jb.ngroups + 1 >= sad::ITEMQ_INTS) { // and one counter per XCD in the (otherwise unused) row_start area
            for (int x = 0; x < 8; ++x) jb.tab[4 + 32 * x] = 0;
            jb.tab[2 + sad::ITEMQ_REFILLS] = 0;  // test instrumentation (common.h)
            jb.tab[2 + sad::ITEMQ_INUSE] = 0;
            jb.tab[2 + sad::ITEMQ_CONFLICT] = 0;
        }
    }
    if (!jb.row_src) return;
    // row map of rows [base, base + blk_rows), by destination row
    for (int q = tid; q < blk_rows; q += SCAN_T) {
        int lo = 0, hi = SCAN_T;                 // largest gi with s_start[gi] <= q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_start[mid] <= q) lo = mid; else hi = mid;
        }
        const int gg = lb * SCAN_T + lo;
        const int r0 = base + s_start[lo], cc = s_start[lo + 1] - s_start[lo];
        const int whole = ((r0 >> 5) == ((r0 + cc - 1) >> 5)) ? WHOLE_BIT : 0;
        const long long b = gg / jb.M;
        jb.row_src[base + q] = (int)(b * jb.N + jb.idx[(long long)gg * jb.S + (q - s_start[lo])]);
        jb.row_gid[base + q] = gg | whole;
    }
}

// Group of the last compact row (used for the clamped rows past the end of the last pass).
__device__ __forceinline__ int s_off_last_group(const int *s_off, int G, int T) {
    int lo = 0, hi = G;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= T - 1) lo = mid; else hi = mid;
    }
    return lo;
}

// Atomic max on floats that are known to be >= +0 (outputs of a ReLU): their bit patterns order
// like unsigned integers.
__device__ __forceinline__ void atomic_max_pos(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned *>(addr), __builtin_bit_cast(unsigned, v));
}

template <int W, int RW, int CW = 1>
__device__ __forceinline__ void mlp_chain_body(const MlpParams &p, const int block_in) {
    static_assert(CW == 1 || RW == 1, "two output tiles per wave only with one row tile");
    int block = block_in;
    if (p.xcd_nb > 0) {                       // XCD-aware chunk order (workgroups go to XCDs round-robin)
        const int per = (p.xcd_nb + 7) >> 3;
        block = (block_in & 7) * per + (block_in >> 3);
        if (block >= p.xcd_nb) return;        // padding workgroup (uniform exit, before any barrier)
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WN = 1 << p.wn_shift;
    const int WM = W >> p.wn_shift;
    const int wn = wave & (WN - 1);
    const int wm = wave >> p.wn_shift;
    const int R = 32 * RW * WM;
    const int PS = 4 * R + 8;             // floats per plane (see act_idx)
    float *bufA = smem;
    float *bufB = smem + (size_t)(p.bufA_rows >> 2) * PS;
    float *sbias = bufB + (size_t)(p.bufB_rows >> 2) * PS;
    int *sm_idx = reinterpret_cast<int *>(sbias + p.bias_total);   // [R] source row / point index
    int *sm_gid = sm_idx + R;                                      // [R] group of the row (-1: none)
    int *s_off = sm_gid + R;                                       // [G+1] first compact row of a group
    const int koff = p.grouped ? 4 : 0;

    // ---- biases -> LDS --------------------------------------------------------------------------
    {
        int bo = 0;
        for (int l = 0; l < p.L; ++l) {
            for (int c = tid; c < p.np[l]; c += W * 64) sbias[bo + c] = p.packed[p.off[l] + c];
            bo += p.np[l];
        }
    }
    // ---- rows of this workgroup -----------------------------------------------------------------
    // plain mode: rows [r0, r0 + R) of the input, one pass.
    // grouped mode: G consecutive (b,m) groups.  The trailing samples of a group that repeat its
    // first index (ball-query padding, SPEC.md §3) are dropped: a duplicate row cannot change the
    // max-pool, so only the leading `cnt` rows of each group are computed.  The surviving rows of the
    // G groups are numbered consecutively ("compact rows") and processed R at a time.
    long long r0 = (long long)block * R;
    long long g0 = 0;
    int T = R, npass = 1, G = p.G;
    const bool dyn = p.grouped && p.rowtab != nullptr;       // global packing + dynamic pass hand-out
    int &s_pass = s_off[p.s_off_entries];   // one int past the offsets (all LDS lives in the dynamic region)
    if (p.grouped && !dyn) {
        g0 = (long long)block * p.G;
        long long left = p.total_groups - g0;
        const int ng = (int)(left < p.G ? left : p.G);
        if (p.cnt) {                                        // counts come from the ball query
            for (int gi = tid; gi < p.G; gi += W * 64) {
                int cnt = 0;
                if (gi < ng) {
                    cnt = p.cnt[g0 + gi];
                    cnt = cnt < 1 ? 1 : (cnt > p.S ? p.S : cnt);
                    if (p.nodedup) cnt = p.S;
                }
                s_off[gi + 1] = cnt;
            }
        } else {
            for (int gi = wave; gi < p.G; gi += W) {       // one wave per group: lane = sample
                int cnt = 0;
                if (gi < ng) {
                    const int32_t *ip = p.idx + (g0 + gi) * p.S;
                    const int v = lane < p.S ? ip[lane] : 0;
                    const int first = __builtin_amdgcn_readfirstlane(v);
                    const unsigned long long diff = __ballot(lane < p.S && v != first);
                    cnt = diff ? 64 - __builtin_clzll(diff) : 1;    // last sample differing from the first, + 1
                    if (p.nodedup) cnt = p.S;
                }
                if (lane == 0) s_off[gi + 1] = cnt;
            }
        }
        if (tid == 0) s_off[0] = 0;
        __syncthreads();
        if (wave == 0) {                                    // inclusive scan of the counts (G <= 1024)
            int carry = 0;
            for (int base = 0; base < p.G; base += 64) {
                const int gi = base + lane;
                int v = gi < p.G ? s_off[gi + 1] : 0;
                for (int off = 1; off < 64; off <<= 1) {
                    const int u = __shfl_up(v, off, 64);
                    if (lane >= off) v += u;
                }
                if (gi < p.G) s_off[gi + 1] = v + carry;
                carry += __builtin_amdgcn_readlane(v, 63);
            }
        }
        __syncthreads();
        T = s_off[p.G];
        npass = (T + R - 1) / R;
    }
    const int nchunks = (p.kp[0] + p.kc - 1) / p.kc;

    // dynamic hand-out: the id of the NEXT pass is fetched (one global atomic, ~1.5 us round trip)
    // while the current pass computes; it reaches LDS at the end of the pass
    int next_pass = 0;
    if (dyn && tid == 0) s_pass = atomicAdd(p.rowtab + 2, 1);
    for (int pass_i = 0; dyn || pass_i < npass; ++pass_i) {
        int pass = pass_i;          // static mode: pass of this workgroup's own rows
        int qoff = pass * R;        // compact-row coordinate of row 0 of the tile in s_off's frame
        if (dyn) {
            __syncthreads();
            pass = s_pass;
            const int total = p.rowtab[0];
            if (pass >= p.rowtab[1]) break;                 // uniform
            T = total - pass * R < R ? total - pass * R : R;
            qoff = 0;
            g0 = 0;                                         // sm_gid holds global group ids in this mode
            for (int r = tid; r < R; r += W * 64) {
                const bool valid = r < T;
                const int q = pass * R + (valid ? r : T - 1);
                sm_idx[r] = p.row_src[q];
                const int gv = p.row_gid[q];
                sm_gid[r] = valid ? gv : ~gv;               // invalid rows: negative, still name a real group
            }
        }
        // ---- per-row source index (static packing / plain rows) ----------------------------------
        for (int r = tid; r < R && !dyn; r += W * 64) {
            if (p.grouped) {
                int q = qoff + r;
                const bool valid = dyn ? r < T : q < T;
                if (!valid) q = T - 1;
                int lo = 0, hi = G;                         // largest gi with s_off[gi] <= q
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_off[mid] <= q) lo = mid; else hi = mid;
                }
                const long long gg = g0 + lo;
                const int b = (int)(gg / p.M);
                sm_idx[r] = (int)((long long)b * p.N + p.idx[gg * p.S + (q - s_off[lo])]);
                sm_gid[r] = valid ? lo : -1;
            } else {
                long long gr = r0 + r;
                if (gr >= p.total_rows) gr = p.total_rows - 1;
                sm_idx[r] = (int)gr;
            }
        }
        __syncthreads();
        if (dyn && tid == 0) next_pass = atomicAdd(p.rowtab + 2, 1);   // everybody has read s_pass by now

        int bias_off = 0;
        for (int l = 0; l < p.L; ++l) {
            const float *in = (l & 1) ? bufB : bufA;
            float *outb = (l & 1) ? bufA : bufB;
            const int n_oc = p.np[l] >> 5;
            const int nrounds = p.flex ? (n_oc * WM + W * CW - 1) / (W * CW) : (n_oc + WN * CW - 1) / (WN * CW);
            const float4 *frags = reinterpret_cast<const float4 *>(p.packed + p.off[l] + p.np[l]);
            const int nT4 = p.kp[l] >> 3;
            const bool last = (l == p.L - 1);
            const bool relu = (p.relu_mask >> l) & 1;
            const int lchunks = (l == 0) ? nchunks : 1;
            // channels of this layer's output the next layer actually reads (its padded K)
            const int keep = last ? p.cout_last : p.kp[l + 1];

            for (int round = 0; round < nrounds; ++round) {
                // classic: wave (wn, wm) owns output tile wn + round*WN of its own RW row tiles;
                // flex: item = wave + round*W -> row tile item % WM (adjacent waves share the weights)
                // (with CW = 2 the wave takes two such tiles per round; they share the row tile)
                int ocs[CW];
                bool haves[CW];
                int rtb = wm * RW;                                      // first row tile of this wave
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const int item = wave + (round * CW + c) * W;
                    ocs[c] = p.flex ? item / WM : wn + ((round * CW + c) << p.wn_shift);
                    if (p.flex) rtb = item - ocs[c] * WM;               // same for every c: WM divides W
                    haves[c] = ocs[c] < n_oc;
                }
                const bool have = haves[0];
                f32x16 accs[CW][RW];
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (!haves[c]) continue;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float4 bv = *reinterpret_cast<const float4 *>(sbias + bias_off + ocs[c] * 32 + 8 * a + 4 * h);
#pragma unroll
                        for (int rt = 0; rt < RW; ++rt) {
                            accs[c][rt][4 * a + 0] = bv.x;
                            accs[c][rt][4 * a + 1] = bv.y;
                            accs[c][rt][4 * a + 2] = bv.z;
                            accs[c][rt][4 * a + 3] = bv.w;
                        }
                    }
                }
                for (int ck = 0; ck < lchunks; ++ck) {
                    const int k0 = (l == 0) ? ck * p.kc : 0;
                    int k1 = (l == 0) ? k0 + p.kc : p.kp[l];
                    if (k1 > p.kp[l]) k1 = p.kp[l];
                    if (l == 0 && (round == 0 || lchunks > 1)) {
                        // ---- stage input channels [k0,k1) of the tile into bufA (channel c - k0) ---
                        if (!(round == 0 && ck == 0)) __syncthreads();  // previous readers are done
                        if (p.grouped && k0 == 0) {  // channels 0..3 = point - centroid, 0 (SPEC §6)
                            for (int r = tid; r < R; r += W * 64) {
                                const int gv = sm_gid[r];
                                const int gi = dyn ? ((gv < 0 ? ~gv : gv) & (WHOLE_BIT - 1))
                                                   : (gv < 0 ? s_off_last_group(s_off, G, T) : gv);
                                const float *q = p.xyz + (long long)sm_idx[r] * 3;
                                const float *c = p.new_xyz + (g0 + gi) * 3;
                                *reinterpret_cast<float2 *>(bufA + r * 4) = make_float2(q[0] - c[0], q[2] - c[2]);
                                *reinterpret_cast<float2 *>(bufA + PS + r * 4) = make_float2(q[1] - c[1], 0.f);
                            }
                        }
                        const int f0 = (k0 > koff ? k0 : koff) - koff;         // first feature channel
                        const int f1 = k1 - koff;                              // one past the last
                        const int fl = f1 < p.C ? f1 : p.C;
                        if (p.cpr > 0) {
                            const int cprp = 1 << p.cshift;
                            const int rpp = 64 >> p.cshift;
                            for (int r = wave * rpp + (lane >> p.cshift); r < R; r += W * rpp) {
                                const float *src = p.feat + (long long)sm_idx[r] * p.ld_feat;
                                for (int ch = (f0 >> 2) + (lane & (cprp - 1)); ch < (fl >> 2); ch += cprp) {
                                    const float4 v = *reinterpret_cast<const float4 *>(src + 4 * ch);
                                    const int c = koff + 4 * ch - k0;  // multiple of 4
                                    float *d = bufA + (size_t)((c >> 3) << 1) * PS + r * 4 + ((c & 7) >> 1);
                                    *reinterpret_cast<float2 *>(d) = make_float2(v.x, v.z);       // even channels
                                    *reinterpret_cast<float2 *>(d + PS) = make_float2(v.y, v.w);  // odd channels
                                }
                            }
                        } else {
                            for (int r = tid; r < R; r += W * 64) {
                                const float *src = p.feat + (long long)sm_idx[r] * p.ld_feat;
                                for (int c = f0; c < fl; ++c) bufA[act_idx(koff + c - k0, r, PS)] = src[c];
                            }
                        }
                        // zero the channel padding [koff + C, kp) that falls inside this chunk
                        const int z0 = (koff + p.C > k0 ? koff + p.C : k0);
                        for (int k = z0 + wave; k < k1; k += W)
                            for (int r = lane; r < R; r += 64) bufA[act_idx(k - k0, r, PS)] = 0.f;
                        __syncthreads();
                    }
                    if (have) {
                        const float4 *bp = reinterpret_cast<const float4 *>(in) + (size_t)h * (PS >> 2) + rtb * 32 + j;
                        const float4 *af = frags + ((size_t)ocs[0] * nT4 + (k0 >> 3)) * 64 + lane;
                        if constexpr (CW == 2) {
                            if (haves[CW - 1]) {
                                const float4 *af1 = frags + ((size_t)ocs[CW - 1] * nT4 + (k0 >> 3)) * 64 + lane;
                                mma_ktile2<SAD_MLP_ADEPTH>(accs[0][0], accs[CW - 1][0], af, af1, (k1 - k0) >> 3, bp, PS >> 1);
                            } else {
                                mma_ktile<RW, SAD_MLP_ADEPTH>(accs[0], af, (k1 - k0) >> 3, bp, PS >> 1);
                            }
                        } else {
                            mma_ktile<RW, SAD_MLP_ADEPTH>(accs[0], af, (k1 - k0) >> 3, bp, PS >> 1);
                        }
                    }
                }
                // ---- epilogue (per output tile of this wave) -----------------------------------
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                if (!haves[cw]) continue;
                const int oc = ocs[cw];
                f32x16 (&acc)[RW] = accs[cw];
                if (relu) {
#pragma unroll
                    for (int rt = 0; rt < RW; ++rt)
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc[rt][g] = acc[rt][g] > 0.f ? acc[rt][g] : 0.f;
                }
                if (!last) {
                    // channel c = oc*32 + 8a + q + 4h -> k-block oc*4 + a, plane q&1, slot (q>>1) + 2h
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (oc * 32 + 8 * a >= keep) continue;   // the next layer never reads these
#pragma unroll
                        for (int rt = 0; rt < RW; ++rt) {
                            float *d = outb + (size_t)((oc * 4 + a) * 2) * PS + ((rtb + rt) * 32 + j) * 4 + 2 * h;
                            *reinterpret_cast<float2 *>(d) = make_float2(acc[rt][4 * a + 0], acc[rt][4 * a + 2]);
                            *reinterpret_cast<float2 *>(d + PS) = make_float2(acc[rt][4 * a + 1], acc[rt][4 * a + 3]);
                        }
                    }
                } else if (!p.grouped) {
                    // plain rows: lane (j,h) holds row j, channels oc*32 + 8a + 4h + (0..3)
#pragma unroll
                    for (int rt = 0; rt < RW; ++rt) {
                        const long long gr = r0 + (rtb + rt) * 32 + j;
                        if (gr >= p.total_rows) continue;
                        float *o = p.out + gr * p.ld_out + p.col_off;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int ch = oc * 32 + 8 * a + 4 * h;
                            if (p.vec_out && ch + 3 < p.cout_last) {
                                *reinterpret_cast<float4 *>(o + ch) = make_float4(acc[rt][4 * a], acc[rt][4 * a + 1], acc[rt][4 * a + 2], acc[rt][4 * a + 3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (ch + e < p.cout_last) o[ch + e] = acc[rt][4 * a + e];
                            }
                        }
                    }
                } else {
                    // max-pool per group: the rows of a group are consecutive lanes (compact rows are
                    // numbered group by group), so a segmented max-scan over the 32 rows of the tile
                    // leaves each group's maximum in its last lane.  A group that lies entirely in
                    // this tile is stored; one that continues in another tile is combined with an
                    // atomic max (outputs are >= 0 after the ReLU and the buffer starts at zero).
#pragma unroll
                    for (int rt = 0; rt < RW; ++rt) {
                        const int rbase = (rtb + rt) * 32;
                        const int gid = sm_gid[rbase + j];
                        int same[5];
#pragma unroll
                        for (int st = 0; st < 5; ++st) {
                            const int og = __shfl_up(gid, 1 << st, 32);
                            same[st] = (j >= (1 << st)) && og == gid;
                        }
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            float v = acc[rt][g];
#pragma unroll
                            for (int st = 0; st < 5; ++st) {
                                const float u = __shfl_up(v, 1 << st, 32);
                                v = (same[st] && u > v) ? u : v;
                            }
                            acc[rt][g] = v;
                        }
                        const int ng = __shfl_down(gid, 1, 32);
                        const bool tail = gid >= 0 && (j == 31 || ng != gid);
                        if (!tail) continue;
                        const int q0 = qoff + rbase;                      // compact row of lane 0 (s_off frame)
                        const bool whole = dyn ? (gid & WHOLE_BIT) != 0 : (s_off[gid] >= q0 && s_off[gid + 1] <= q0 + 32);
                        float *o = p.out + (g0 + (dyn ? (gid & (WHOLE_BIT - 1)) : gid)) * p.ld_out + p.col_off;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int ch = oc * 32 + 8 * a + 4 * h;
                            if (whole && p.vec_out && ch + 3 < p.cout_last) {
                                *reinterpret_cast<float4 *>(o + ch) = make_float4(acc[rt][4 * a], acc[rt][4 * a + 1], acc[rt][4 * a + 2], acc[rt][4 * a + 3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (ch + e >= p.cout_last) continue;
                                    if (whole) o[ch + e] = acc[rt][4 * a + e];
                                    else atomic_max_pos(o + ch + e, acc[rt][4 * a + e]);
                                }
                            }
                        }
                    }
                }
                }   // cw
            }
            bias_off += p.np[l];
            __syncthreads();
        }
        if (dyn && tid == 0) s_pass = next_pass;
    }
}

template <int W, int RW>
__global__ __launch_bounds__(W * 64) void mlp_chain_kernel(const MlpParams p) {
    mlp_chain_body<W, RW>(p, blockIdx.x);
}

template <int W>
__global__ __launch_bounds__(W * 64) void mlp_chain2_kernel(const MlpParams p) {   // two output tiles per wave
    mlp_chain_body<W, 1, 2>(p, blockIdx.x);
}

// Several independent chains (the branches of one multi-radius stage) in ONE dispatch: the
// workgroups of the chains are laid out one after the other (heaviest first), so the light chains
// fill the tail of the heavy one and the launch gaps between them disappear.  All chains share the
// wave count W; the row blocking RW is a per-chain runtime switch (RWMAX bounds the register budget).
constexpr int MULTI_MAX = 4;
struct MultiParams {
    MlpParams p[MULTI_MAX];
    int first[MULTI_MAX + 1];   // first block of chain i; first[n] = grid size
    int rw[MULTI_MAX];
    int cw[MULTI_MAX];          // 2 = two output tiles per wave (rw == 1)
    int n;
};

template <int W, int RWMAX, bool CW2>
__global__ __launch_bounds__(W * 64) void mlp_multi_kernel(const MultiParams mp) {
    int c = 0;
    while (c + 1 < mp.n && (int)blockIdx.x >= mp.first[c + 1]) ++c;
    c = __builtin_amdgcn_readfirstlane(c);
    const int block = blockIdx.x - mp.first[c];
    const int rw = mp.rw[c];
    // (the two-tile body is compiled in only when a chain of the dispatch uses it: it needs more
    // registers than the others and would lower everybody's occupancy)
    if (CW2 && mp.cw[c] == 2) mlp_chain_body<W, 1, (CW2 ? 2 : 1)>(mp.p[c], block);
    else if (rw == 1) mlp_chain_body<W, 1>(mp.p[c], block);
    else if (RWMAX >= 2 && rw == 2) mlp_chain_body<W, (RWMAX >= 2 ? 2 : 1)>(mp.p[c], block);
    else if (RWMAX >= 4 && rw == 4) mlp_chain_body<W, (RWMAX >= 4 ? 4 : 1)>(mp.p[c], block);
}

// ---- narrow chains on the vector ALU: one (b, m, s) row per lane ------------------------------
// For SA1-sized chains (3 layers, widths <= 64, <= 8 input channels) a 32x32 MFMA tile is mostly
// padding and the LDS round trips / barriers of the tiled kernel dominate.  Here every lane carries
// its row through the whole chain in registers with v_fma_f32 (the same k-ascending fmaf chain, so
// still bit-identical), weights arrive through the scalar cache as SGPR operands, the max over the
// nsample lanes of a group is a DPP butterfly, and there is no LDS and no barrier at all.
struct ValuParams {
    const float *xyz, *new_xyz, *feat;
    const int32_t *idx, *cnt;
    const float *w[3], *b[3];
    float *out;
    long long total_groups;
    int ld_feat, N, M, S, G, nodedup, ld_out, col_off, vec_out;
};

constexpr int VALU_T = 256;
constexpr int VALU_GMAX = 2048;

template <int C0, int C1, int C2, int C3>
__global__ __launch_bounds__(VALU_T) void mlp_valu_kernel(const ValuParams p) {
    __shared__ int s_off[VALU_GMAX + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long g0 = (long long)blockIdx.x * p.G;
    const long long left = p.total_groups - g0;
    const int ng = (int)(left < p.G ? left : p.G);
    // rows kept per group (padding rows that repeat the first index are dropped, as in the tiled kernel)
    for (int gi = tid; gi < p.G; gi += VALU_T) {
        int cnt = 0;
        if (gi < ng) {
            if (p.cnt) {
                cnt = p.cnt[g0 + gi];
            } else {
                const int32_t *ip = p.idx + (g0 + gi) * p.S;
                const int first = ip[0];
                cnt = 1;
                for (int s = 1; s < p.S; ++s) cnt = ip[s] != first ? s + 1 : cnt;
            }
            cnt = cnt < 1 ? 1 : (cnt > p.S ? p.S : cnt);
            if (p.nodedup) cnt = p.S;
        }
        s_off[gi + 1] = cnt;
    }
    if (tid == 0) s_off[0] = 0;
    __syncthreads();
    if (tid < 64) {
        int carry = 0;
        for (int base = 0; base < p.G; base += 64) {
            const int gi = base + lane;
            int v = gi < p.G ? s_off[gi + 1] : 0;
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, off, 64);
                if (lane >= off) v += u;
            }
            if (gi < p.G) s_off[gi + 1] = v + carry;
            carry += __builtin_amdgcn_readlane(v, 63);
        }
    }
    __syncthreads();
    const int T = s_off[p.G];
    for (int q0 = 0; q0 < T; q0 += VALU_T) {
    int q = q0 + tid;
    const bool live = q < T;
    if (!live) q = T - 1;
    int glo = 0, ghi = p.G;                          // largest gi with s_off[gi] <= q
    while (ghi - glo > 1) {
        const int mid = (glo + ghi) >> 1;
        if (s_off[mid] <= q) glo = mid; else ghi = mid;
    }
    const int gid = live ? glo : -1;
    const long long bm = g0 + glo;
    const int b = (int)(bm / p.M);
    const long long pt = (long long)b * p.N + p.idx[bm * p.S + (q - s_off[glo])];
    float x[C0];
    {
        const float *qq = p.xyz + pt * 3;
        const float *c = p.new_xyz + bm * 3;
        x[0] = qq[0] - c[0];
        x[1] = qq[1] - c[1];
        x[2] = qq[2] - c[2];
#pragma unroll
        for (int k = 3; k < C0; ++k) x[k] = p.feat[pt * p.ld_feat + (k - 3)];
    }
    // k-major weights Wt[k][o]: the chain of output o is still fma(W[o][k], x[k], .) for k ascending
    float h1[C1], h2[C2], h3[C3];
#pragma unroll
    for (int o = 0; o < C1; ++o) h1[o] = p.b[0][o];
#pragma unroll
    for (int k = 0; k < C0; ++k)
#pragma unroll
        for (int o = 0; o < C1; ++o) h1[o] = __builtin_fmaf(p.w[0][k * C1 + o], x[k], h1[o]);
#pragma unroll
    for (int o = 0; o < C1; ++o) h1[o] = __builtin_fmaxf(h1[o], 0.f);
#pragma unroll
    for (int o = 0; o < C2; ++o) h2[o] = p.b[1][o];
#pragma unroll
    for (int k = 0; k < C1; ++k)
#pragma unroll
        for (int o = 0; o < C2; ++o) h2[o] = __builtin_fmaf(p.w[1][k * C2 + o], h1[k], h2[o]);
#pragma unroll
    for (int o = 0; o < C2; ++o) h2[o] = __builtin_fmaxf(h2[o], 0.f);
#pragma unroll
    for (int o = 0; o < C3; ++o) h3[o] = p.b[2][o];
#pragma unroll
    for (int k = 0; k < C2; ++k)
#pragma unroll
        for (int o = 0; o < C3; ++o) h3[o] = __builtin_fmaf(p.w[2][k * C3 + o], h2[k], h3[o]);
#pragma unroll
    for (int o = 0; o < C3; ++o) h3[o] = __builtin_fmaxf(h3[o], 0.f);
    // max-pool per group: rows of a group are consecutive lanes -> segmented max-scan over the wave;
    // a group inside one wave of one pass is stored, otherwise combined with an atomic max
    int same[6];
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int og = __shfl_up(gid, 1 << st, 64);
        same[st] = (lane >= (1 << st)) && og == gid;
    }
#pragma unroll
    for (int o = 0; o < C3; ++o) {
        float v = h3[o];
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const float u = __shfl_up(v, 1 << st, 64);
            v = (same[st] && u > v) ? u : v;
        }
        h3[o] = v;
    }
    const int ngid = __shfl_down(gid, 1, 64);
    const bool tail = gid >= 0 && (lane == 63 || ngid != gid);
    if (tail) {
        const int w0 = q0 + (tid & ~63);                 // compact row of lane 0 of this wave
        const bool whole = s_off[gid] >= w0 && s_off[gid + 1] <= w0 + 64;
        float *o = p.out + (g0 + gid) * p.ld_out + p.col_off;
        if (whole && p.vec_out) {
#pragma unroll
            for (int c = 0; c < C3; c += 4) *reinterpret_cast<float4 *>(o + c) = make_float4(h3[c], h3[c + 1], h3[c + 2], h3[c + 3]);
        } else if (whole) {
#pragma unroll
            for (int c = 0; c < C3; ++c) o[c] = h3[c];
        } else {
#pragma unroll
            for (int c = 0; c < C3; ++c) atomic_max_pos(o + c, h3[c]);
        }
    }
    }   // pass loop
}

// ---- weight packing --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ Wm,
                                                   const float *__restrict__ bias, int Cin, int Cout,
                                                   int KP, int NP, int has_xyz,
                                                   float *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nfrag = (long long)NP * KP;
    if (t < NP) {
        dst[t] = t < Cout ? bias[t] : 0.f;
        return;
    }
    const long long q = t - NP;
    if (q >= nfrag) return;
    const int e = (int)(q & 3);
    const int lane = (int)((q >> 2) & 63);
    const long long blk = q >> 8;  // oc_tile * nT4 + t4
    const int nT4 = KP >> 3;
    const int t4 = (int)(blk % nT4);
    const int oct = (int)(blk / nT4);
    const int oc = oct * 32 + (lane & 31);
    const int kpad = 8 * t4 + 2 * e + (lane >> 5);
    int k = kpad;
    bool ok = true;
    if (has_xyz) {
        if (kpad == 3) ok = false;       // the zero lane after x,y,z
        else if (kpad > 3) k = kpad - 1;
    }
    ok = ok && (k < Cin) && (oc < Cout);
    dst[t] = ok ? Wm[(size_t)oc * Cin + k] : 0.f;
}

// Stream image of a 3-layer chain (mlp_coop.hip): the A fragments of the three layers copied into the order a tile
// consumes them — per layer-0 output tile o its NT0 fragments, then the 4*NO1 layer-1 fragments it feeds (output tile
// i / 4, k-group 4o + i % 4), then layer 2 tile by tile; zero fragments pad the last stage.
__global__ __launch_bounds__(256) void stream_pack_kernel(const float *__restrict__ packed, long long off0, long long off1, long long off2,
                                                          int NT0, int NO0, int NO1, int NO2, long long nfrag, float *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 per thread
    if (t >= nfrag * 64) return;
    const long long p = t >> 6;
    const int lane = (int)(t & 63);
    const int NI = 4 * NO1, PER = NT0 + NI, TOT = NO0 * PER, NG1 = 4 * NO0, NG2 = 4 * NO1;
    const float4 *f0 = reinterpret_cast<const float4 *>(packed + off0 + NO0 * 32);
    const float4 *f1 = reinterpret_cast<const float4 *>(packed + off1 + NO1 * 32);
    const float4 *f2 = reinterpret_cast<const float4 *>(packed + off2 + NO2 * 32);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < TOT) {
        const int o = (int)(p / PER), r = (int)(p % PER);
        if (r < NT0) v = f0[(long long)(o * NT0 + r) * 64 + lane];
        else {
            const int i = r - NT0;
            v = f1[(long long)((i >> 2) * NG1 + 4 * o + (i & 3)) * 64 + lane];
        }
    } else if (p < TOT + (long long)NO2 * NG2) {
        v = f2[(p - TOT) * 64 + lane];
    }
    reinterpret_cast<float4 *>(dst)[t] = v;
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ Wm, int Cout, int Cin,
                                                        float *__restrict__ Wt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Cout * Cin) return;
    const int k = t / Cout, o = t % Cout;
    Wt[t] = Wm[(size_t)o * Cin + k];
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Geometry {
    int kp[MAXL], np[MAXL];
    long long off[MAXL];     // packed layer l: bias block then A fragments
    long long raw_w[MAXL];   // plain row-major copy W[l][C_out][C_in] (for the VALU kernel)
    long long raw_b[MAXL];
    long long stream_off;    // stream image of the cooperative register-resident kernel (mlp_coop.hip), -1 = none
    long long stream_frags;
    long long total;
};

inline Geometry geometry(int L, const int *dims, int first_has_xyz) {
    Geometry g{};
    long long off = 0;
    for (int l = 0; l < L; ++l) {
        const int cin = dims[l] + ((l == 0 && first_has_xyz) ? 1 : 0);
        g.kp[l] = round_up(cin, 8);
        g.np[l] = round_up(dims[l + 1], 32);
        g.off[l] = off;
        off += (long long)g.np[l] + (long long)g.np[l] * g.kp[l];
    }
    for (int l = 0; l < L; ++l) {
        g.raw_w[l] = off;
        off += round_up(dims[l] * dims[l + 1], 4);
        g.raw_b[l] = off;
        off += round_up(dims[l + 1], 4);
    }
    g.stream_off = -1;
    g.stream_frags = 0;
    if (first_has_xyz && L == 3) {
        const int shape = sad::reg_shape_id(L, g.kp, g.np);
        const long long nf = shape >= 0 ? sad::coop_stream_frags(shape, g.kp, g.np) : 0;
        if (nf > 0) {
            off = (off + 63) / 64 * 64;       // 256-byte aligned
            g.stream_off = off;
            g.stream_frags = nf;
            off += nf * 256;
        }
    }
    g.total = off;
    return g;
}

int check_dims(const char *fn, int L, const int *dims) {
    if (L < 1 || L > MAXL || !dims) return sad::fail(SAD_EINVAL, "%s: L=%d not in 1..%d", fn, L, MAXL);
    for (int l = 0; l <= L; ++l)
        if (dims[l] < 1 || dims[l] > 4096) return sad::fail(SAD_EUNSUPPORTED, "%s: dims[%d]=%d not in 1..4096", fn, l, dims[l]);
    return SAD_OK;
}

template <int W, int RW>
int launch_mlp(const MlpParams &p, size_t lds, long long nblocks, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};   // one mask per template instantiation
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_chain_kernel<W, RW>), 160 * 1024);
    hipLaunchKernelGGL((mlp_chain_kernel<W, RW>), dim3((unsigned)nblocks), dim3(W * 64), lds, st, p);
    return sad::check_launch("sad_mlp_chain_f32");
}

}  // namespace

SAD_API size_t sad_mlp_packed_floats(int L, const int *dims, int first_has_xyz) {
    if (L < 1 || L > MAXL || !dims) return 0;
    return (size_t)geometry(L, dims, first_has_xyz).total;
}

SAD_API int sad_mlp_pack_f32(int L, const int *dims, int first_has_xyz, const float *const *Wm,
                             const float *const *bias, float *packed, sad_stream_t stream) {
    if (int e = check_dims("sad_mlp_pack_f32", L, dims)) return e;
    SAD_REQUIRE(Wm && bias && packed, "sad_mlp_pack_f32: NULL pointer");
    SAD_REQUIRE(!first_has_xyz || dims[0] >= 3, "sad_mlp_pack_f32: first_has_xyz needs dims[0] >= 3");
    const Geometry g = geometry(L, dims, first_has_xyz);
    for (int l = 0; l < L; ++l) {
        SAD_REQUIRE(Wm[l] && bias[l], "sad_mlp_pack_f32: NULL weight pointer for layer %d", l);
        const long long n = (long long)g.np[l] + (long long)g.np[l] * g.kp[l];
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, Wm[l], bias[l], dims[l], dims[l + 1], g.kp[l], g.np[l],
                           (l == 0 && first_has_xyz) ? 1 : 0, packed + g.off[l]);
        // k-major copy Wt[k][o] (for the VALU kernel: (o, o+1) weight pairs are adjacent)
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((dims[l] * dims[l + 1] + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, Wm[l], dims[l + 1], dims[l], packed + g.raw_w[l]);
        if (hipMemcpyAsync(packed + g.raw_b[l], bias[l], sizeof(float) * dims[l + 1],
                           hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return sad::fail(SAD_ELAUNCH, "sad_mlp_pack_f32: device copy of layer %d failed", l);
    }
    if (g.stream_off >= 0) {
        const long long n4 = g.stream_frags * 64;
        hipLaunchKernelGGL(stream_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed, g.off[0], g.off[1],
                           g.off[2], g.kp[0] / 8, g.np[0] / 32, g.np[1] / 32, g.np[2] / 32, g.stream_frags, packed + g.stream_off);
    }
    return sad::check_launch("sad_mlp_pack_f32");
}

namespace sad {
// Fills a ScanJob for one chain (table layout: see sad_mlp_workspace_bytes).
ScanJob make_scan_job(const int32_t *cnt, int ngroups, int S, int R, int *tab, int nodedup, const int32_t *idx, int N, int M) {
    ScanJob jb{};
    jb.cnt = cnt; jb.idx = idx; jb.tab = tab; jb.ngroups = ngroups; jb.S = S; jb.N = N; jb.M = M; jb.nodedup = nodedup; jb.R = R;
    // layout (ints): hdr[4] | row_start[ngroups+1] (unused) | pass_first[ngroups*S/32+2] (unused) | blk_sum[ngroups/1024+2]
    //                | row map: src[ngroups*S] | gid[ngroups*S]   (only written when idx != NULL)
    jb.blk_sum = tab + 4 + (ngroups + 1) + ((long long)ngroups * S / 32 + 2);
    if (idx) {
        jb.row_src = jb.blk_sum + (ngroups / 1024 + 2);
        jb.row_gid = jb.row_src + (long long)ngroups * S;
    }
    return jb;
}

// Prefix-sums the per-group row counts of up to SCAN_MAX_CHAINS chains with two launches (shared with the bf16 chain).
int launch_rowscan_multi(const ScanJob *jobs, int n, hipStream_t st) {
    ScanMulti sm{};
    sm.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        sm.j[i] = jobs[i];
        sm.j[i].blk0 = blocks;
        blocks += (jobs[i].ngroups + SCAN_T - 1) / SCAN_T;
    }
    hipLaunchKernelGGL(rowscan_sums_kernel, dim3(blocks), dim3(SCAN_T), 0, st, sm);
    hipLaunchKernelGGL(rowscan_write_kernel, dim3(blocks), dim3(SCAN_T), 0, st, sm);
    return check_launch("rowscan");
}

int launch_rowscan(const int32_t *cnt, int ngroups, int S, int R, int *tab, hipStream_t st, int nodedup,
                   const int32_t *idx, int N, int M) {
    const ScanJob jb = make_scan_job(cnt, ngroups, S, R, tab, nodedup, idx, N, M);
    return launch_rowscan_multi(&jb, 1, st);
}
}  // namespace sad

SAD_API int sad_mlp_rowscan(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                            int M, void *const *workspace, sad_stream_t stream) {
    SAD_REQUIRE(n >= 1 && n <= sad::SCAN_MAX_CHAINS && cnt && idx && S && workspace, "sad_mlp_rowscan: need 1..%d chains and non-NULL arrays", sad::SCAN_MAX_CHAINS);
    SAD_REQUIRE(B >= 1 && N >= 1 && M >= 1 && (long long)B * M < (1LL << 30), "sad_mlp_rowscan: bad B/N/M");
    sad::ScanJob jobs[sad::SCAN_MAX_CHAINS];
    for (int i = 0; i < n; ++i) {
        SAD_REQUIRE(cnt[i] && idx[i] && workspace[i] && S[i] >= 1 && S[i] <= 64, "sad_mlp_rowscan: chain %d: NULL pointer or bad nsample", i);
        SAD_REQUIRE((uintptr_t)workspace[i] % 16 == 0, "sad_mlp_rowscan: workspace must be 16-byte aligned");
        SAD_REQUIRE((long long)B * M * S[i] < (1LL << 31), "sad_mlp_rowscan: B*M*S too large");
        jobs[i] = sad::make_scan_job(cnt[i], B * M, S[i], 32, (int *)workspace[i], sad::get_option(sad::OPT_MLP_NODEDUP), idx[i], N, M);
    }
    return sad::launch_rowscan_multi(jobs, n, (hipStream_t)stream);
}

SAD_API size_t sad_mlp_scratch_bytes(int B, int M, int S, int L, const int *dims) {
    if (B < 1 || M < 1 || S < 1 || L < 1 || L > MAXL || !dims) return 0;
    const Geometry g = geometry(L, dims, 1);
    int wa = 0, wb = 0;
    for (int l = 0; l + 1 < L; ++l) {
        int &w = (l & 1) ? wb : wa;
        w = g.np[l] > w ? g.np[l] : w;
    }
    const size_t rows_max = ((size_t)B * M * S + 31) / 32 * 32;
    return 64 + sizeof(float) * rows_max * (size_t)(wa + wb);
}

SAD_API int sad_mlp_preferred_geometry(int L, const int *dims) {
    if (L < 1 || L > MAXL || !dims) return 0;
    for (int l = 0; l <= L; ++l)
        if (dims[l] < 1 || dims[l] > 4096) return 0;
    if (dims[0] < 3) return 0;
    const Geometry g = geometry(L, dims, 1);
    const int shape = sad::reg_shape_id(L, g.kp, g.np);
    const int C = dims[0] - 3;
    const bool rows16 = C >= 4 && C % 4 == 0;               // 16-byte feature rows (the caller's ld_feat must agree)
    if (shape >= 0 && sad::coop_shape(shape) && sad::reg_family(shape) == 2 && g.stream_off >= 0 && rows16) return 4;   // (SA3; SA2 is as fast on 2)
    if (shape >= 0 && (C == 0 || C == 1 || rows16)) return 2;
    bool wide = rows16;
    for (int l = 0; l < L; ++l) wide = wide && g.np[l] % 128 == 0 && (l == 0 || g.kp[l] == g.np[l - 1]);
    return wide ? 3 : 0;
}

SAD_API size_t sad_mlp_workspace_bytes(int B, int M, int S) {
    if (B < 1 || M < 1 || S < 1) return 0;
    const size_t ng = (size_t)B * M;
    // hdr, row_start, pass_first (R >= 32), block sums of the two-launch scan
    return sizeof(int) * (4 + (ng + 1) + (ng * S / 32 + 2) + (ng / 1024 + 2) + 2 * ng * S) + 64;   // + row map
}

namespace {
struct Prepared {
    MlpParams p;
    size_t lds;
    long long nblocks;
    int W, RW, CW;
    bool launched;      // the VALU kernel was launched instead (nothing left to do)
    bool reg;           // geometry 2: register-resident chain kernel (csrc/mlp_reg.hip); `rc` is filled, p is not
    bool coop;          // ... geometry 4: its cooperative variant (csrc/mlp_coop.hip)
    sad::RegChain rc;
    sad::ScanJob scan;  // row-packing scan this chain needs before its kernel
    bool prescanned;    // ... unless the caller already ran sad_mlp_rowscan on the workspace
    bool layered;       // geometry 3: layer-streamed chain (csrc/mlp_layer.hip); lj[0..nl) are its launches
    sad::LayerJob lj[MAXL];
    int nl;
    long long layer_items[MAXL];
    int reg_shape;
    long long reg_tiles;   // upper bound of the tile count
};
int launch_prepared(const Prepared &q, hipStream_t st);
}  // namespace

// mlp_layer.hip forms the byte offset of an input row as a 32-bit product (row * ld * 4): the first layer's rows
// (`in_rows` feature rows of stride `in_ld`) and every hidden activation matrix (rows_max x np[l]) must stay below 4 GiB.
static bool layer_offsets_fit(long long in_rows, long long in_ld, long long rows_max, int L, const int *np) {
    const long long lim = 1LL << 32;
    if (in_rows * in_ld * 4 >= lim) return false;
    for (int l = 0; l + 1 < L; ++l)
        if (rows_max * np[l] * 4 >= lim) return false;
    return true;
}

// Validation, geometry choice and the row-packing scan of one chain; fills `q` for the launch.
static int prepare_chain(const sad_mlp_args *a, sad_stream_t stream, Prepared &q) {
    q.launched = false;
    q.reg = false;
    q.coop = false;
    q.layered = false;
    q.prescanned = a && a->prescanned != 0;
    SAD_REQUIRE(a, "sad_mlp_chain_f32: NULL args");
    SAD_REQUIRE(a->struct_size == sizeof(sad_mlp_args), "sad_mlp_chain_f32: struct_size=%zu, this library's sad_mlp_args has %zu bytes "
                "(caller built against another sad_amd.h; ABI version %d)", a->struct_size, sizeof(sad_mlp_args), SAD_ABI_VERSION);
    if (int e = check_dims("sad_mlp_chain_f32", a->L, a->dims)) return e;
    const bool grouped = a->idx != nullptr;
    SAD_REQUIRE(a->packed && a->out, "sad_mlp_chain_f32: NULL packed/out");
    SAD_REQUIRE(a->B >= 1 && a->M >= 1 && a->C >= 0, "sad_mlp_chain_f32: bad B/M/C");
    SAD_REQUIRE(a->C == 0 || a->feat, "sad_mlp_chain_f32: C=%d but feat is NULL", a->C);
    SAD_REQUIRE(a->C == 0 || a->ld_feat >= a->C, "sad_mlp_chain_f32: ld_feat=%d < C=%d", a->ld_feat, a->C);
    if (grouped) {
        SAD_REQUIRE(a->xyz && a->new_xyz, "sad_mlp_chain_f32: grouped mode needs xyz and new_xyz");
        SAD_REQUIRE(a->N >= 1 && a->S >= 1 && a->S <= 64, "sad_mlp_chain_f32: need N>=1, 1<=S<=64 (S=%d)", a->S);
        SAD_REQUIRE(a->dims[0] == a->C + 3, "sad_mlp_chain_f32: dims[0]=%d != C+3=%d", a->dims[0], a->C + 3);
        SAD_REQUIRE((long long)a->B * a->N < (1LL << 31), "sad_mlp_chain_f32: B*N too large");
        SAD_REQUIRE((long long)a->B * a->M * a->S < (1LL << 31), "sad_mlp_chain_f32: B*M*S too large (row numbers are 32-bit)");
        // straddling groups are merged with an unsigned atomic max into a zero-initialised buffer:
        // only valid for non-negative outputs, i.e. a ReLU after every layer (SPEC.md §6 grouped chains)
        SAD_REQUIRE((a->relu_mask & ((1 << a->L) - 1)) == (1 << a->L) - 1,
                    "sad_mlp_chain_f32: grouped chains need a ReLU after every layer (relu_mask=0x%x, L=%d)", a->relu_mask, a->L);
    } else {
        SAD_REQUIRE(a->S == 1, "sad_mlp_chain_f32: plain mode needs S == 1");
        SAD_REQUIRE(a->C >= 1 && a->dims[0] == a->C, "sad_mlp_chain_f32: dims[0]=%d != C=%d", a->dims[0], a->C);
        SAD_REQUIRE((long long)a->B * a->M < (1LL << 31), "sad_mlp_chain_f32: too many rows");
    }
    const int cout = a->dims[a->L];
    SAD_REQUIRE(a->ld_out >= a->col_off + cout && a->col_off >= 0, "sad_mlp_chain_f32: ld_out=%d too small for col_off=%d + C_out=%d", a->ld_out, a->col_off, cout);

    MlpParams p{};
    const Geometry g = geometry(a->L, a->dims, grouped);
    p.xyz = a->xyz; p.new_xyz = a->new_xyz; p.idx = a->idx; p.cnt = a->cnt; p.feat = a->feat; p.packed = a->packed;
    p.out = a->out; p.ld_feat = a->ld_feat; p.N = a->N; p.M = a->M; p.S = a->S; p.C = a->C;
    p.grouped = grouped; p.L = a->L; p.relu_mask = a->relu_mask; p.ld_out = a->ld_out;
    p.col_off = a->col_off; p.cout_last = cout;
    int sp_shift = 0;
    while ((1 << sp_shift) < a->S) ++sp_shift;
    p.sp_shift = sp_shift;
    p.total_rows = (long long)a->B * a->M << sp_shift;   // plain mode: rows; VALU kernel: padded rows
    p.total_groups = (long long)a->B * a->M;
    // (the mlp_force knob of tests / sweeps wins over the per-call field)
    int geom_all = sad::get_option(sad::OPT_MLP_FORCE) ? sad::get_option(sad::OPT_MLP_FORCE) : a->geometry;
    const int flex_code = (geom_all / 100000) % 10;          // bit 0 = flexible item distribution, bit 1 = two output tiles per wave (both need RW == 1)
    const int dyn_code = (geom_all / 10000) % 10;            // 0 = heuristic, 1 = global packing, 2 = per-workgroup packing
    const int fcode = (geom_all / 1000) % 10;                // 0 = default
    const int geom_wg = geom_all % 1000;
    int dedup_f = sad::get_option(sad::OPT_MLP_DEDUP_F) > 0 ? sad::get_option(sad::OPT_MLP_DEDUP_F) : 8;
    if (fcode >= 1 && fcode <= 7) dedup_f = 1 << fcode;
    if (sad::get_option(sad::OPT_MLP_NODEDUP)) dedup_f = 1;
    int max_noc = 1, min_noc = 1 << 30;
    for (int l = 0; l < a->L; ++l) {
        p.kp[l] = g.kp[l]; p.np[l] = g.np[l]; p.off[l] = g.off[l];
        max_noc = g.np[l] / 32 > max_noc ? g.np[l] / 32 : max_noc;
        min_noc = g.np[l] / 32 < min_noc ? g.np[l] / 32 : min_noc;
    }
    p.vec_out = (a->ld_out % 4 == 0 && a->col_off % 4 == 0 && ((uintptr_t)a->out % 16 == 0)) ? 1 : 0;
    // feature staging: 16-B chunks when rows are 16-B aligned
    if (a->C >= 4 && a->C % 4 == 0 && a->ld_feat % 4 == 0 && ((uintptr_t)a->feat % 16 == 0)) {
        p.cpr = a->C / 4;
        int cs = 0;
        while ((1 << cs) < p.cpr && cs < 6) ++cs;
        p.cshift = cs;
    } else {
        p.cpr = 0; p.cshift = 0;
    }
    // ---- geometry 2: register-resident chain (one wave per 32-row tile, no LDS round trips, no barriers) ----
    if (geom_wg == 2 || geom_wg == 4) {
        const int shape = grouped ? sad::reg_shape_id(a->L, g.kp, g.np) : -1;
        if (geom_wg == 4 && (shape < 0 || !sad::coop_shape(shape) || g.stream_off < 0 || p.cpr <= 0))
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 4 (cooperative register-resident chain) is compiled for the SA2 / SA3 shapes");
        const bool all_relu = (a->relu_mask & ((1 << a->L) - 1)) == (1 << a->L) - 1;
        const bool feat_ok = a->C == 0 || a->C == 1 || p.cpr > 0;
        if (shape < 0 || !all_relu || !feat_ok || !a->cnt || !a->workspace)
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 2 (register-resident chain) needs a compiled 3-layer shape, "
                                               "cnt + workspace and 16-byte feature rows");
        SAD_REQUIRE((uintptr_t)a->workspace % 16 == 0, "sad_mlp_chain_f32: workspace must be 16-byte aligned");
        SAD_REQUIRE(p.total_groups < (1LL << 30), "sad_mlp_chain_f32: too many groups");
        int *tab = (int *)a->workspace;
        // (the scan is launched by the caller: the chains of a merged dispatch share its two launches)
        q.scan = sad::make_scan_job(a->cnt, (int)p.total_groups, a->S, 32, tab, sad::get_option(sad::OPT_MLP_NODEDUP), a->idx, a->N, a->M);
        sad::RegChain &rc = q.rc;
        rc.xyz = a->xyz; rc.new_xyz = a->new_xyz; rc.feat = a->feat; rc.packed = a->packed; rc.out = a->out;
        rc.rowtab = tab;
        rc.row_src = tab + 4 + (p.total_groups + 1) + (p.total_groups * a->S / 32 + 2) + (p.total_groups / 1024 + 2);
        rc.row_gid = rc.row_src + p.total_groups * a->S;
        for (int l = 0; l < 3; ++l) { rc.off[l] = g.off[l]; rc.np[l] = g.np[l]; }
        rc.stream_off = g.stream_off;
        q.coop = geom_wg == 4;
        rc.ld_feat = a->ld_feat; rc.C = a->C; rc.cpr = p.cpr;
        rc.ld_out = a->ld_out; rc.col_off = a->col_off; rc.cout_last = cout; rc.vec_out = p.vec_out;
        q.reg = true;
        q.reg_shape = shape;
        q.reg_tiles = (p.total_groups * a->S + 31) / 32;
        q.W = -1;
        return SAD_OK;
    }
    // ---- geometry 3: layer-streamed chain (one launch per layer, activations between layers in scratch) ----
    if (geom_wg == 3 && !grouped) {
        // plain rows: every layer is a row-major GEMM launch (bias + optional ReLU); the last one writes the caller's
        // output slice, whole 128-channel blocks at a time, so C_out must be its own padded width
        bool ok = p.cpr > 0 && a->C % 8 == 0 && p.vec_out && g.np[a->L - 1] == cout;
        for (int l = 0; l < a->L; ++l) ok = ok && g.np[l] % 128 == 0 && (l == 0 || g.kp[l] == g.np[l - 1]);
        ok = ok && (a->L == 1 || (a->scratch && a->scratch_bytes >= sad_mlp_scratch_bytes(a->B, a->M, 1, a->L, a->dims)));
        if (!ok)
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 3 (layer-streamed chain) on plain rows needs 16-byte rows, "
                                               "C %% 8 == 0, layer widths that are multiples of 128 and, for L > 1, scratch");
        SAD_REQUIRE(a->L == 1 || (uintptr_t)a->scratch % 16 == 0, "sad_mlp_chain_f32: scratch must be 16-byte aligned");
        const long long rows_max = (p.total_rows + 31) / 32 * 32;
        int wa = 0, wb = 0;
        for (int l = 0; l + 1 < a->L; ++l) {
            int &w = (l & 1) ? wb : wa;
            w = g.np[l] > w ? g.np[l] : w;
        }
        // mlp_layer_kernel addresses a layer's input rows with 32-bit byte offsets (row * ld * 4)
        if (!layer_offsets_fit(p.total_rows, a->ld_feat, rows_max, a->L, g.np))
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 3 (layer-streamed chain): rows x row stride x 4 must stay below 4 GiB "
                                               "(%lld rows); split the call or use the tiled kernel", p.total_rows);
        float *ha = a->L > 1 ? (float *)((char *)a->scratch + 64) : nullptr;
        float *hb = ha ? ha + rows_max * wa : nullptr;
        q.nl = a->L;
        for (int l = 0; l < a->L; ++l) {
            sad::LayerJob &j = q.lj[l];
            j = sad::LayerJob{};
            j.rows = (int)p.total_rows;
            j.packed = a->packed; j.off = g.off[l]; j.np = g.np[l]; j.kg = g.kp[l] / 8; j.nog = g.np[l] / 128;
            j.relu = (a->relu_mask >> l) & 1;
            if (l == 0) { j.x = a->feat; j.ldx = a->ld_feat; }
            else { j.x = ((l - 1) & 1) ? hb : ha; j.ldx = g.np[l - 1]; }
            if (l + 1 < a->L) { j.y = (l & 1) ? hb : ha; j.ldy = g.np[l]; }
            else { j.y = a->out + a->col_off; j.ldy = a->ld_out; }
            q.layer_items[l] = (p.total_rows + 127) / 128 * j.nog;
        }
        q.prescanned = true;     // no row map
        q.layered = true;
        q.W = -2;
        return SAD_OK;
    }
    if (geom_wg == 3) {
        bool ok = grouped && a->cnt && a->workspace && a->scratch && p.cpr > 0;
        const bool all_relu = (a->relu_mask & ((1 << a->L) - 1)) == (1 << a->L) - 1;
        for (int l = 0; l < a->L; ++l) ok = ok && g.np[l] % 128 == 0 && (l == 0 || g.kp[l] == g.np[l - 1]);
        ok = ok && all_relu && a->scratch_bytes >= sad_mlp_scratch_bytes(a->B, a->M, a->S, a->L, a->dims);
        if (!ok)
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 3 (layer-streamed chain) needs cnt + workspace + scratch, "
                                               "16-byte feature rows and layer widths that are multiples of 128");
        SAD_REQUIRE((uintptr_t)a->workspace % 16 == 0 && (uintptr_t)a->scratch % 16 == 0, "sad_mlp_chain_f32: workspace / scratch must be 16-byte aligned");
        SAD_REQUIRE(p.total_groups < (1LL << 30), "sad_mlp_chain_f32: too many groups");
        int *tab = (int *)a->workspace;
        q.scan = sad::make_scan_job(a->cnt, (int)p.total_groups, a->S, 32, tab, sad::get_option(sad::OPT_MLP_NODEDUP), a->idx, a->N, a->M);
        const long long rows_max = (p.total_groups * a->S + 31) / 32 * 32;
        int wa = 0, wb = 0;                         // widths of the two ping-pong activation buffers
        for (int l = 0; l + 1 < a->L; ++l) {
            int &w = (l & 1) ? wb : wa;
            w = g.np[l] > w ? g.np[l] : w;
        }
        if (!layer_offsets_fit((long long)a->B * a->N, a->ld_feat, rows_max, a->L, g.np))
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 3 (layer-streamed chain): rows x row stride x 4 must stay below 4 GiB "
                                               "(B*N = %lld feature rows, %lld grouped rows); split the call or use the tiled kernel",
                             (long long)a->B * a->N, rows_max);
        float *ha = (float *)((char *)a->scratch + 64);
        float *hb = ha + rows_max * wa;
        q.nl = a->L;
        for (int l = 0; l < a->L; ++l) {
            sad::LayerJob &j = q.lj[l];
            j = sad::LayerJob{};
            j.rowtab = tab; j.row_src = q.scan.row_src; j.row_gid = q.scan.row_gid;
            j.packed = a->packed; j.off = g.off[l]; j.np = g.np[l]; j.kg = g.kp[l] / 8; j.nog = g.np[l] / 128;
            j.relu = 1; j.last = l == a->L - 1;
            if (l == 0) {
                j.gather = 1; j.x = a->feat; j.ldx = a->ld_feat; j.cpr = p.cpr; j.xyz = a->xyz; j.new_xyz = a->new_xyz;
            } else {
                j.x = ((l - 1) & 1) ? hb : ha; j.ldx = g.np[l - 1];
            }
            if (!j.last) { j.y = (l & 1) ? hb : ha; j.ldy = g.np[l]; }
            else { j.out = a->out; j.ld_out = a->ld_out; j.col_off = a->col_off; j.cout_last = cout; }
            q.layer_items[l] = (rows_max + 127) / 128 * j.nog;      // (128-row block) x (128-channel block) work items
        }
        q.layered = true;
        q.W = -2;
        return SAD_OK;
    }
    // ---- narrow 3-layer grouped chains can run on the vector ALU (geometry 1; autotune tries it) ----
    {
        const int gsel = geom_wg;
        const int *d = a->dims;
        const bool all_relu = (a->relu_mask & 7) == 7;
        int shape = 0;
        if (grouped && a->L == 3 && all_relu && d[0] == 4 && d[1] == 16 && d[2] == 16 && d[3] == 32) shape = 1;
        if (grouped && a->L == 3 && all_relu && d[0] == 4 && d[1] == 32 && d[2] == 32 && d[3] == 64) shape = 2;
        if (gsel == 1 && !shape) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: no VALU kernel for this chain");
        if (shape && gsel == 1) {   // only on request: it computes the padding rows the tiled kernel skips
            ValuParams v{};
            v.xyz = a->xyz; v.new_xyz = a->new_xyz; v.feat = a->feat; v.idx = a->idx; v.cnt = a->cnt; v.out = a->out;
            for (int l = 0; l < 3; ++l) { v.w[l] = a->packed + g.raw_w[l]; v.b[l] = a->packed + g.raw_b[l]; }
            v.total_groups = p.total_groups; v.ld_feat = a->ld_feat; v.N = a->N; v.M = a->M; v.S = a->S;
            v.ld_out = a->ld_out; v.col_off = a->col_off; v.vec_out = p.vec_out;
            v.nodedup = sad::get_option(sad::OPT_MLP_NODEDUP);
            long long gq = (long long)dedup_f * VALU_T / a->S;   // groups whose surviving rows fill ~one pass
            v.G = (int)(gq < 1 ? 1 : (gq > VALU_GMAX ? VALU_GMAX : gq));
            const long long nb = (p.total_groups + v.G - 1) / v.G;
            SAD_REQUIRE(nb < (1LL << 31), "sad_mlp_chain_f32: too many workgroups");
            if (shape == 1)
                hipLaunchKernelGGL((mlp_valu_kernel<4, 16, 16, 32>), dim3((unsigned)nb), dim3(VALU_T), 0, (hipStream_t)stream, v);
            else
                hipLaunchKernelGGL((mlp_valu_kernel<4, 32, 32, 64>), dim3((unsigned)nb), dim3(VALU_T), 0, (hipStream_t)stream, v);
            q.launched = true;
            return sad::check_launch("sad_mlp_chain_f32 (valu)");
        }
    }
    // LDS rows: bufA holds inputs of even layers / outputs of odd layers, bufB the others.  A layer's
    // output only needs the channels the next layer reads (its padded K).
    auto lds_rows = [&](int kc, int &ra, int &rb) {
        ra = kc; rb = 8;
        for (int l = 0; l + 1 < a->L; ++l) {  // the last layer's output never touches LDS
            int &dst = (l & 1) ? ra : rb;
            const int keep = g.kp[l + 1];
            dst = keep > dst ? keep : dst;
        }
    };
    int bias_total = 0;
    for (int l = 0; l < a->L; ++l) bias_total += g.np[l];
    p.bias_total = bias_total;
    // grouped mode: a workgroup owns G groups; with ball-query padding dropped their surviving rows
    // usually fit one pass of R rows (dedup_f = assumed ratio of padded to surviving rows)
    auto groups_per_wg = [&](int R) {
        long long gq = (long long)dedup_f * R / a->S;
        return (int)(gq < 1 ? 1 : (gq > 1024 ? 1024 : gq));
    };
    auto lds_bytes = [&](int w, int wns, int rw, int kcc) {
        int ra, rb;
        lds_rows(kcc, ra, rb);
        const size_t R = 32 * (size_t)rw * (w >> wns);
        const size_t G = grouped ? (size_t)groups_per_wg((int)R) : 0;
        const size_t nso = (G > R + 1 ? G : R + 1) + 2;     // s_off entries (static G+1, dynamic <= R+2) + broadcast slot
        return ((size_t)((ra + rb) / 4) * (4 * R + 8) + 2 * R + nso + (size_t)bias_total) * 4 + 16;
    };
    // ---- choose the workgroup geometry -----------------------------------------------------
    // W waves, WN along output tiles; R = 32*RW*(W/WN) rows per workgroup.
    const int bkb = sad::get_option(sad::OPT_MLP_BUDGET_KB);
    const size_t BUDGET2 = (size_t)(bkb > 0 ? bkb : 78) * 1024, BUDGET1 = 156 * 1024;
    const int rw_min = 1;
    int W = 8, wn_shift = 0, RW = 1, kc = g.kp[0];
    int geom = geom_wg;
    if (geom) {
        const int fw = geom / 100, fwns = (geom / 10) % 10, frw = geom % 10;
        const bool ok = (fw == 4 || fw == 8 || fw == 16) && (1 << fwns) <= fw && (frw == 1 || frw == 2 || frw == 4) && frw >= rw_min;
        if (!ok) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry %d is not valid here", geom);
        W = fw; wn_shift = fwns; RW = frw;
        if (lds_bytes(W, wn_shift, RW, kc) > BUDGET1) {
            kc = g.kp[0] < 256 ? g.kp[0] : 256;
            if (kc == g.kp[0] || lds_bytes(W, wn_shift, RW, kc) > BUDGET1)
                return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry %d does not fit LDS", geom);
        }
    } else {
        // Heuristic (PackedMLP.autotune measures instead): 8 waves; WN = flop-weighted mean number
        // of output tiles rounded down to a power of two (few idle waves, weights shared through
        // LDS rows rather than re-read per wave); then the most rows per wave that still leave two
        // workgroups per CU and at least two workgroups per CU-slot of work.
        double wsum = 0, fsum = 0;
        for (int l = 0; l < a->L; ++l) {
            const double f = (double)g.kp[l] * g.np[l];
            wsum += f * (g.np[l] / 32);
            fsum += f;
        }
        int wns0 = 0;
        while (wns0 < 3 && (2 << wns0) <= wsum / fsum + 1e-9) ++wns0;
        bool found = false;
        for (int pass = 0; pass < 3 && !found; ++pass) {
            const size_t budget = pass == 0 ? BUDGET2 : BUDGET1;
            const int kcc = pass < 2 ? g.kp[0] : (g.kp[0] < 256 ? g.kp[0] : 256);
            for (int wns = wns0; wns <= 3 && !found; ++wns) {
                int best_rw = 0;
                for (int rw = 4; rw >= rw_min; rw >>= 1) {
                    if (lds_bytes(8, wns, rw, kcc) > budget) continue;
                    if (!best_rw) best_rw = rw;   // largest that fits
                    const long long R = 32LL * rw * (8 >> wns);
                    const long long nb = grouped ? (p.total_groups * a->S / dedup_f + R - 1) / R : (p.total_rows + R - 1) / R;
                    if (nb >= 1024 || rw == rw_min) { best_rw = rw; break; }
                }
                if (best_rw) { wn_shift = wns; RW = best_rw; kc = kcc; found = true; }
            }
        }
        if (!found) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: layer widths do not fit LDS");
    }
    p.wn_shift = wn_shift;
    if (flex_code > 3 || (flex_code && RW != 1))
        return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: flexible distribution / two tiles per wave need RW == 1");
    if ((flex_code & 2) && W == 16) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: two tiles per wave: 4 or 8 waves");
    p.flex = flex_code & 1;
    p.kc = kc;
    lds_rows(kc, p.bufA_rows, p.bufB_rows);
    const size_t lds = lds_bytes(W, wn_shift, RW, kc);
    const size_t lds_final = lds;
    const long long R = 32LL * RW * (W >> wn_shift);
    p.G = grouped ? groups_per_wg((int)R) : 0;
    p.s_off_entries = (int)((p.G > R + 1 ? p.G : R + 1) + 1);
    p.nodedup = sad::get_option(sad::OPT_MLP_NODEDUP);
    p.rowtab = nullptr;
    long long grid_dyn = 0;
    long long macs_per_row = 0;
    for (int l = 0; l < a->L; ++l) macs_per_row += (long long)g.kp[l] * g.np[l];
    // global packing pays off when the prefix-sum workgroup is cheap (few groups) and rows are
    // expensive (measured: cluster.b1 +7 %, sa3.b2 +3 %, every small chain slower)
    const bool want_dyn = dyn_code == 1 || (dyn_code == 0 && p.total_groups <= 16384 && macs_per_row >= 400000);
    if (grouped && a->cnt && a->workspace && want_dyn && !sad::get_option(sad::OPT_MLP_STATIC)) {
        // global row packing: scan the counts once, then a persistent grid pulls full passes
        SAD_REQUIRE((uintptr_t)a->workspace % 16 == 0, "sad_mlp_chain_f32: workspace must be 16-byte aligned");
        SAD_REQUIRE(p.total_groups < (1LL << 30), "sad_mlp_chain_f32: too many groups");
        p.rowtab = (int *)a->workspace;
        if (int e = sad::launch_rowscan(a->cnt, (int)p.total_groups, a->S, (int)R, p.rowtab, (hipStream_t)stream, p.nodedup,
                                        a->idx, a->N, a->M)) return e;
        p.row_src = p.rowtab + 4 + (p.total_groups + 1) + (p.total_groups * a->S / 32 + 2) + (p.total_groups / 1024 + 2);
        p.row_gid = p.row_src + p.total_groups * a->S;
        const long long upper = (p.total_groups * a->S + R - 1) / R;
        long long per_cu = lds_final > 80 * 1024 ? 1 : (lds_final > 52 * 1024 ? 2 : (lds_final > 39 * 1024 ? 3 : 4));
        if (sad::get_option(sad::OPT_MLP_DYN_SLOTS) > 0 && sad::get_option(sad::OPT_MLP_DYN_SLOTS) < per_cu)
            per_cu = sad::get_option(sad::OPT_MLP_DYN_SLOTS);
        grid_dyn = upper < 256 * per_cu ? upper : 256 * per_cu;
    }
    const long long nblocks = grid_dyn ? grid_dyn : (grouped ? (p.total_groups + p.G - 1) / p.G : (p.total_rows + R - 1) / R);
    SAD_REQUIRE(nblocks < (1LL << 31), "sad_mlp_chain_f32: too many workgroups");
    if (W == 16 && RW == 4) return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: 16 waves support RW 1 or 2");
    q.p = p; q.lds = lds; q.nblocks = nblocks; q.W = W; q.RW = RW; q.CW = (flex_code & 2) ? 2 : 1;
    if (!grid_dyn && !sad::get_option(sad::OPT_MLP_NOXCD)) {       // static chunks: XCD-aware order, grid padded to 8 * ceil(nb / 8)
        q.p.xcd_nb = (int)nblocks;
        q.nblocks = ((nblocks + 7) / 8) * 8;
    }
    return SAD_OK;
}

namespace {
template <int W>
int launch_mlp2(const MlpParams &p, size_t lds, long long nblocks, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_chain2_kernel<W>), 160 * 1024);
    hipLaunchKernelGGL((mlp_chain2_kernel<W>), dim3((unsigned)nblocks), dim3(W * 64), lds, st, p);
    return sad::check_launch("sad_mlp_chain_f32");
}

// scans of the chains that did not come with a table (prescanned = 0), in one pair of launches
int launch_pending_scans(const Prepared *const *qs, int n, hipStream_t st) {
    sad::ScanJob jobs[sad::REG_MAX_CHAINS];
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (!qs[i]->prescanned) jobs[m++] = qs[i]->scan;
    return m ? sad::launch_rowscan_multi(jobs, m, st) : SAD_OK;
}

int launch_reg_chains(const Prepared *const *qs, int n, hipStream_t st) {
    sad::RegMulti mp{};
    mp.n = n;
    mp.max_tiles = 0;
    if (int e = launch_pending_scans(qs, n, st)) return e;
    for (int i = 0; i < n; ++i) {
        mp.c[i] = qs[i]->rc;
        mp.shape[i] = qs[i]->reg_shape;
        mp.max_tiles += qs[i]->reg_tiles;
    }
    mp.counter = const_cast<int *>(mp.c[0].rowtab) + 2;      // zeroed by chain 0's rowscan
    mp.nq = qs[0]->scan.ngroups + 1 >= sad::ITEMQ_INTS ? 8 : 1;
    return qs[0]->coop ? sad::launch_coop(mp, st) : sad::launch_reg(mp, st);
}

// Layer-streamed chains (one or two with the same number of layers): one scan, then one
// launch per layer carrying that layer of every chain.
int launch_layered_chains(const Prepared *const *qs, int n, hipStream_t st) {
    if (int e = launch_pending_scans(qs, n, st)) return e;
    for (int l = 0; l < qs[0]->nl; ++l) {
        sad::LayerMulti lm{};
        lm.n = n;
        long long items = 0;
        for (int i = 0; i < n; ++i) { lm.j[i] = qs[i]->lj[l]; items += qs[i]->layer_items[l]; }
        // (items of a layer launch are dealt statically: pulling them from per-XCD queues as the cooperative kernel does
        // cost the pipeline 2.5 %)
        if (int e = sad::launch_layers(lm, items, st)) return e;
    }
    return SAD_OK;
}

int launch_prepared(const Prepared &q, hipStream_t st) {
    if (q.layered) {
        const Prepared *one = &q;
        return launch_layered_chains(&one, 1, st);
    }
    if (q.reg) {
        const Prepared *one = &q;
        return launch_reg_chains(&one, 1, st);
    }
    const MlpParams &p = q.p;
    const size_t lds = q.lds;
    const long long nblocks = q.nblocks;
    if (q.CW == 2) return q.W == 8 ? launch_mlp2<8>(p, lds, nblocks, st) : launch_mlp2<4>(p, lds, nblocks, st);
    if (q.W == 16) {
        if (q.RW == 1) return launch_mlp<16, 1>(p, lds, nblocks, st);
        return launch_mlp<16, 2>(p, lds, nblocks, st);
    }
    if (q.W == 8) {
        if (q.RW == 1) return launch_mlp<8, 1>(p, lds, nblocks, st);
        if (q.RW == 2) return launch_mlp<8, 2>(p, lds, nblocks, st);
        return launch_mlp<8, 4>(p, lds, nblocks, st);
    }
    if (q.RW == 1) return launch_mlp<4, 1>(p, lds, nblocks, st);
    if (q.RW == 2) return launch_mlp<4, 2>(p, lds, nblocks, st);
    return launch_mlp<4, 4>(p, lds, nblocks, st);
}

template <int W, int RWMAX, bool CW2>
int launch_multi(const MultiParams &mp, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_multi_kernel<W, RWMAX, CW2>), 160 * 1024);
    hipLaunchKernelGGL((mlp_multi_kernel<W, RWMAX, CW2>), dim3((unsigned)mp.first[mp.n]), dim3(W * 64), lds, st, mp);
    return sad::check_launch("sad_mlp_chain_multi_f32");
}
}  // namespace

SAD_API int sad_mlp_chain_f32(const sad_mlp_args *a, sad_stream_t stream) {
    Prepared q;
    if (int e = prepare_chain(a, stream, q)) return e;
    if (q.launched) return SAD_OK;
    return launch_prepared(q, (hipStream_t)stream);
}

SAD_API int sad_mlp_chain_multi_f32(const sad_mlp_args *const *args, int n, sad_stream_t stream) {
    SAD_REQUIRE(args && n >= 1, "sad_mlp_chain_multi_f32: need at least one chain");
    hipStream_t st = (hipStream_t)stream;
    if (n > MULTI_MAX) {                       // more chains than one dispatch carries: in groups
        for (int i = 0; i < n; i += MULTI_MAX)
            if (int e = sad_mlp_chain_multi_f32(args + i, n - i < MULTI_MAX ? n - i : MULTI_MAX, stream)) return e;
        return SAD_OK;
    }
    Prepared q[MULTI_MAX];
    for (int i = 0; i < n; ++i)
        if (int e = prepare_chain(args[i], stream, q[i])) return e;
    // two layer-streamed chains with the same depth: their layers share launches (heavier chain first)
    if (n == 2 && q[0].layered && q[1].layered && q[0].nl == q[1].nl) {
        const bool swap = q[1].layer_items[q[1].nl - 1] * (long long)q[1].lj[q[1].nl - 1].kg > q[0].layer_items[q[0].nl - 1] * (long long)q[0].lj[q[0].nl - 1].kg;
        const Prepared *ord[2] = {swap ? &q[1] : &q[0], swap ? &q[0] : &q[1]};
        return launch_layered_chains(ord, 2, st);
    }
    // register-resident chains of one shape family: one dispatch, tiles of the heaviest chain first
    {
        bool all_reg = n > 1 && n <= sad::REG_MAX_CHAINS;
        for (int i = 0; i < n; ++i) all_reg = all_reg && q[i].reg && q[i].coop == q[0].coop && sad::reg_family(q[i].reg_shape) == sad::reg_family(q[0].reg_shape);
        if (all_reg) {
            const Prepared *ord[MULTI_MAX];
            for (int i = 0; i < n; ++i) ord[i] = &q[i];
            auto heavy = [&](const Prepared *s) {       // MACs per row x rows (upper bound)
                double m = 0;
                for (int l = 0; l < 3; ++l) m += (double)s->rc.np[l] * (l == 0 ? 8.0 * 17 : s->rc.np[l - 1]);
                return m * (double)s->reg_tiles;
            };
            for (int i = 0; i < n; ++i)
                for (int k = i + 1; k < n; ++k)
                    if (heavy(ord[k]) > heavy(ord[i])) { const Prepared *t = ord[i]; ord[i] = ord[k]; ord[k] = t; }
            return launch_reg_chains(ord, n, st);
        }
    }
    // one dispatch needs a common wave count and nothing already launched; otherwise one by one
    bool merge = n > 1;
    for (int i = 0; i < n; ++i) merge = merge && !q[i].launched && !q[i].reg && !q[i].layered && q[i].W == q[0].W && q[i].W != 16;
    if (!merge) {
        for (int i = 0; i < n; ++i)
            if (!q[i].launched)
                if (int e = launch_prepared(q[i], st)) return e;
        return SAD_OK;
    }
    // heaviest chain first (its workgroups start first, the light chains fill its tail)
    int order[MULTI_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    auto weight = [&](int i) {
        double m = 0;
        for (int l = 0; l < q[i].p.L; ++l) m += (double)q[i].p.kp[l] * q[i].p.np[l];
        return m * (double)q[i].p.total_rows;
    };
    for (int i = 0; i < n; ++i)
        for (int k = i + 1; k < n; ++k)
            if (weight(order[k]) > weight(order[i])) { const int t = order[i]; order[i] = order[k]; order[k] = t; }
    MultiParams mp{};
    mp.n = n;
    size_t lds = 0;
    long long total = 0;
    int rwmax = 1;
    for (int i = 0; i < n; ++i) {
        const Prepared &s = q[order[i]];
        mp.p[i] = s.p;
        mp.rw[i] = s.RW;
        mp.cw[i] = s.CW;
        mp.first[i] = (int)total;
        total += (s.nblocks + 7) / 8 * 8;          // chain starts stay multiples of 8 (XCD-aware chunk order)
        lds = s.lds > lds ? s.lds : lds;
        rwmax = s.RW > rwmax ? s.RW : rwmax;
    }
    SAD_REQUIRE(total < (1LL << 31), "sad_mlp_chain_multi_f32: too many workgroups");
    mp.first[n] = (int)total;
    bool cw2 = false;
    for (int i = 0; i < n; ++i) cw2 = cw2 || mp.cw[i] == 2;
    if (q[0].W == 8) {
        if (cw2) {
            if (rwmax == 1) return launch_multi<8, 1, true>(mp, lds, st);
            if (rwmax == 2) return launch_multi<8, 2, true>(mp, lds, st);
            return launch_multi<8, 4, true>(mp, lds, st);
        }
        if (rwmax == 1) return launch_multi<8, 1, false>(mp, lds, st);
        if (rwmax == 2) return launch_multi<8, 2, false>(mp, lds, st);
        return launch_multi<8, 4, false>(mp, lds, st);
    }
    if (cw2) {
        if (rwmax == 1) return launch_multi<4, 1, true>(mp, lds, st);
        if (rwmax == 2) return launch_multi<4, 2, true>(mp, lds, st);
        return launch_multi<4, 4, true>(mp, lds, st);
    }
    if (rwmax == 1) return launch_multi<4, 1, false>(mp, lds, st);
    if (rwmax == 2) return launch_multi<4, 2, false>(mp, lds, st);
    return launch_multi<4, 4, false>(mp, lds, st);
}
