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
#include "mlp_chain.h"

namespace {

using namespace sad::chain;

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SAD_MLP_BDEPTH
#define SAD_MLP_BDEPTH 2
#endif

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
template <int W, int RW>
int launch_mlp(const MlpParams &p, size_t lds, long long nblocks, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};   // one mask per template instantiation
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_chain_kernel<W, RW>), 160 * 1024);
    hipLaunchKernelGGL((mlp_chain_kernel<W, RW>), dim3((unsigned)nblocks), dim3(W * 64), lds, st, p);
    return sad::check_launch("sad_mlp_chain_f32");
}

template <int W>
int launch_mlp2(const MlpParams &p, size_t lds, long long nblocks, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_chain2_kernel<W>), 160 * 1024);
    hipLaunchKernelGGL((mlp_chain2_kernel<W>), dim3((unsigned)nblocks), dim3(W * 64), lds, st, p);
    return sad::check_launch("sad_mlp_chain_f32");
}

template <int W, int RWMAX, bool CW2>
int launch_multi(const MultiParams &mp, size_t lds, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    sad::lds_attr_once(attr_done, reinterpret_cast<const void *>(&mlp_multi_kernel<W, RWMAX, CW2>), 160 * 1024);
    hipLaunchKernelGGL((mlp_multi_kernel<W, RWMAX, CW2>), dim3((unsigned)mp.first[mp.n]), dim3(W * 64), lds, st, mp);
    return sad::check_launch("sad_mlp_chain_multi_f32");
}

}  // namespace

namespace sad {
namespace chain {

int launch_tiled(const MlpParams &p, int W, int RW, int CW, size_t lds, long long nblocks, hipStream_t st) {
    if (CW == 2) return W == 8 ? launch_mlp2<8>(p, lds, nblocks, st) : launch_mlp2<4>(p, lds, nblocks, st);
    if (W == 16) {
        if (RW == 1) return launch_mlp<16, 1>(p, lds, nblocks, st);
        return launch_mlp<16, 2>(p, lds, nblocks, st);
    }
    if (W == 8) {
        if (RW == 1) return launch_mlp<8, 1>(p, lds, nblocks, st);
        if (RW == 2) return launch_mlp<8, 2>(p, lds, nblocks, st);
        return launch_mlp<8, 4>(p, lds, nblocks, st);
    }
    if (RW == 1) return launch_mlp<4, 1>(p, lds, nblocks, st);
    if (RW == 2) return launch_mlp<4, 2>(p, lds, nblocks, st);
    return launch_mlp<4, 4>(p, lds, nblocks, st);
}

int launch_tiled_multi(const MultiParams &mp, int W, int rwmax, bool cw2, size_t lds, hipStream_t st) {
    if (W == 8) {
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

int launch_valu(const ValuParams &v, int shape, long long nb, hipStream_t st) {
    if (shape == 1)
        hipLaunchKernelGGL((mlp_valu_kernel<4, 16, 16, 32>), dim3((unsigned)nb), dim3(VALU_T), 0, st, v);
    else
        hipLaunchKernelGGL((mlp_valu_kernel<4, 32, 32, 64>), dim3((unsigned)nb), dim3(VALU_T), 0, st, v);
    return sad::check_launch("sad_mlp_chain_f32 (valu)");
}

}  // namespace chain
}  // namespace sad
