// Global row packing for the grouped-MLP kernels (SPEC.md §6: ball-query padding rows cannot change the max-pool and are
// never computed).  No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
#include "mlp_chain.h"

namespace {

using sad::chain::WHOLE_BIT;

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
// 256 groups per workgroup: four waves of ~40 registers find room on any CU — also beside an FPS workgroup of another batch
// (which leaves 64 registers per SIMD), where the 1 024-thread workgroups of the first version did not fit
constexpr int SCAN_T = 256;
constexpr int SCAN_LOG = 8;
constexpr int SCAN_WAVES = SCAN_T / 64;
static_assert(SCAN_T == 1 << SCAN_LOG, "the row search of rowscan_write_kernel is SCAN_LOG halvings");

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
    __shared__ int wsum[SCAN_WAVES];
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
        for (int w = 0; w < SCAN_WAVES; ++w) t += wsum[w];
        jb.blk_sum[lb] = t;
    }
}

__global__ __launch_bounds__(SCAN_T) void rowscan_write_kernel(const sad::ScanMulti sm) {
    __shared__ int wsum[SCAN_WAVES];
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
        for (int w = 0; w < SCAN_WAVES; ++w) t += wsum[w];
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
    if (jb.split) {
        // split pooling: first packed row of every group (the consumer of the pooled rows derives a group's continuation tiles
        // from it), and the all-zero row 0 of the continuation buffer (what a group without continuation reads)
        if (g < jb.ngroups) jb.gstart[g] = base + run;
        if (g == jb.ngroups - 1) jb.gstart[jb.ngroups] = base + run + c;
        if (lb == 0)
            for (int i = tid; i < jb.cont_cols / 2; i += SCAN_T) reinterpret_cast<unsigned *>(jb.cont0)[i] = 0u;
    }
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
    if (jb.zout) {
        // zero the output rows of the groups that straddle a tile boundary (the only ones combined with an atomic max)
        __shared__ int s_nw[SCAN_T];
        __shared__ int s_nnw;
        if (tid == 0) s_nnw = 0;
        __syncthreads();
        if (g < jb.ngroups) {
            const int r0 = base + run;
            if ((r0 >> 5) != ((r0 + c - 1) >> 5)) s_nw[atomicAdd(&s_nnw, 1)] = g;
        }
        __syncthreads();
        const int nnw = s_nnw;
        if (((jb.zld | jb.zcols) & 3) == 0 && (reinterpret_cast<uintptr_t>(jb.zout) & 15) == 0) {
            const int per = jb.zcols >> 2;
            for (int i = tid; i < nnw * per; i += SCAN_T)
                reinterpret_cast<float4 *>(jb.zout + (long long)s_nw[i / per] * jb.zld)[i % per] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int i = tid; i < nnw * jb.zcols; i += SCAN_T) jb.zout[(long long)s_nw[i / jb.zcols] * jb.zld + i % jb.zcols] = 0.f;
        }
    }
    if (!jb.row_src) return;
    // row map of rows [base, base + blk_rows), by destination row
    // Four rows per thread and step (eight: slower, the clamped surplus searches cost more than the overlap buys): a row is a chain of ten dependent LDS reads (the search) and a dependent gather of its
    // index, and a block has 5 - 30 rows per thread — one at a time, a block of the SA3 stage took 17 us of pure latency.
    constexpr int U = 4;
    for (int q0 = tid; q0 < blk_rows; q0 += U * SCAN_T) {
        int lo[U], src[U], gid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * SCAN_T < blk_rows ? q0 + u * SCAN_T : blk_rows - 1;     // (clamped: searched, not stored)
            int l = 0, h = SCAN_T;               // largest gi with s_start[gi] <= q
#pragma unroll
            for (int it = 0; it < SCAN_LOG; ++it) {
                const int mid = (l + h) >> 1;
                const bool le = s_start[mid] <= q;
                l = le ? mid : l;
                h = le ? h : mid;
            }
            lo[u] = l;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * SCAN_T < blk_rows ? q0 + u * SCAN_T : blk_rows - 1;
            const int gg = lb * SCAN_T + lo[u];
            const int r0 = base + s_start[lo[u]], cc = s_start[lo[u] + 1] - s_start[lo[u]];
            const int whole = ((r0 >> 5) == ((r0 + cc - 1) >> 5)) ? WHOLE_BIT : 0;
            const int cont = (jb.split && ((base + q) >> 5) != (r0 >> 5)) ? sad::CONT_BIT : 0;
            const long long b = gg / jb.M;
            src[u] = (int)(b * jb.N + jb.idx[(long long)gg * jb.S + (q - s_start[lo[u]])]);
            gid[u] = gg | whole | cont;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * SCAN_T;
            if (q < blk_rows) {
                jb.row_src[base + q] = src[u];
                jb.row_gid[base + q] = gid[u];
            }
        }
    }
}
}  // namespace

namespace sad {
// Fills a ScanJob for one chain (table layout: see sad_mlp_workspace_bytes).
ScanJob make_scan_job(const int32_t *cnt, int ngroups, int S, int R, int *tab, int nodedup, const int32_t *idx, int N, int M) {
    ScanJob jb{};
    jb.cnt = cnt; jb.idx = idx; jb.tab = tab; jb.ngroups = ngroups; jb.S = S; jb.N = N; jb.M = M; jb.nodedup = nodedup; jb.R = R;
    // layout (ints): hdr[4] | row_start[ngroups+1] (item queues in its first 258 ints, else unused) | pass_first[ngroups*S/32+2]
    //                (holds the block sums: ngroups/256 + 1 of them) | [ngroups/1024+2] (unused: the block sums of 1 024-group
    //                blocks lived here) | row map: src[ngroups*S] | gid[ngroups*S]   (only written when idx != NULL)
    jb.blk_sum = tab + 4 + (ngroups + 1);
    if (idx) {
        jb.row_src = tab + 4 + (ngroups + 1) + ((long long)ngroups * S / 32 + 2) + (ngroups / 1024 + 2);
        jb.row_gid = jb.row_src + (long long)ngroups * S;
    }
    jb.gstart = tab + scan_gstart_off(ngroups, S);      // (written by a split-pooling scan only)
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
#ifdef SAD_SCAN_TWICE       // measurement build: the scan's work doubled (both kernels are idempotent: same tables, same zero fill) — what its time costs the step
    hipLaunchKernelGGL(rowscan_sums_kernel, dim3(blocks), dim3(SCAN_T), 0, st, sm);
    hipLaunchKernelGGL(rowscan_write_kernel, dim3(blocks), dim3(SCAN_T), 0, st, sm);
#endif
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

SAD_API int sad_mlp_rowscan_init(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                                 int M, void *const *workspace, float *const *out, const int *ld_out, const int *col_off,
                                 const int *cout, sad_stream_t stream) {
    SAD_REQUIRE(n >= 1 && n <= sad::SCAN_MAX_CHAINS && cnt && idx && S && workspace && out && ld_out && col_off && cout,
                "sad_mlp_rowscan_init: need 1..%d chains and non-NULL arrays", sad::SCAN_MAX_CHAINS);
    SAD_REQUIRE(B >= 1 && N >= 1 && M >= 1 && (long long)B * M < (1LL << 30), "sad_mlp_rowscan_init: bad B/N/M");
    sad::ScanJob jobs[sad::SCAN_MAX_CHAINS];
    for (int i = 0; i < n; ++i) {
        SAD_REQUIRE(cnt[i] && idx[i] && workspace[i] && S[i] >= 1 && S[i] <= 64, "sad_mlp_rowscan_init: chain %d: NULL pointer or bad nsample", i);
        SAD_REQUIRE((uintptr_t)workspace[i] % 16 == 0, "sad_mlp_rowscan_init: workspace must be 16-byte aligned");
        SAD_REQUIRE((long long)B * M * S[i] < (1LL << 31), "sad_mlp_rowscan_init: B*M*S too large");
        SAD_REQUIRE(out[i] && cout[i] >= 1 && col_off[i] >= 0 && ld_out[i] >= col_off[i] + cout[i],
                    "sad_mlp_rowscan_init: chain %d: need out != NULL and col_off + cout <= ld_out", i);
        jobs[i] = sad::make_scan_job(cnt[i], B * M, S[i], 32, (int *)workspace[i], sad::get_option(sad::OPT_MLP_NODEDUP), idx[i], N, M);
        jobs[i].zout = out[i] + col_off[i];
        jobs[i].zld = ld_out[i];
        jobs[i].zcols = cout[i];
    }
    return sad::launch_rowscan_multi(jobs, n, (hipStream_t)stream);
}

SAD_API int sad_mlp_rowscan_split(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                                  int M, void *const *workspace, void *const *cont, const int *cout, sad_stream_t stream) {
    SAD_REQUIRE(n >= 1 && n <= sad::SCAN_MAX_CHAINS && cnt && idx && S && workspace && cont && cout,
                "sad_mlp_rowscan_split: need 1..%d chains and non-NULL arrays", sad::SCAN_MAX_CHAINS);
    SAD_REQUIRE(B >= 1 && N >= 1 && M >= 1 && (long long)B * M < (long long)sad::CONT_BIT, "sad_mlp_rowscan_split: bad B/N/M");
    sad::ScanJob jobs[sad::SCAN_MAX_CHAINS];
    for (int i = 0; i < n; ++i) {
        SAD_REQUIRE(cnt[i] && idx[i] && workspace[i] && S[i] >= 1 && S[i] <= 64, "sad_mlp_rowscan_split: chain %d: NULL pointer or bad nsample", i);
        SAD_REQUIRE((uintptr_t)workspace[i] % 16 == 0, "sad_mlp_rowscan_split: workspace must be 16-byte aligned");
        SAD_REQUIRE((long long)B * M * S[i] < (1LL << 31), "sad_mlp_rowscan_split: B*M*S too large");
        SAD_REQUIRE(cont[i] && (uintptr_t)cont[i] % 16 == 0 && cout[i] >= 8 && cout[i] % 8 == 0,
                    "sad_mlp_rowscan_split: chain %d: need a 16-byte aligned continuation buffer and cout %% 8 == 0", i);
        jobs[i] = sad::make_scan_job(cnt[i], B * M, S[i], 32, (int *)workspace[i], 0, idx[i], N, M);
        jobs[i].split = 1;
        jobs[i].cont0 = cont[i];
        jobs[i].cont_cols = cout[i];
    }
    return sad::launch_rowscan_multi(jobs, n, (hipStream_t)stream);
}

SAD_API size_t sad_mlp_cont_bytes(int B, int M, int S, int cout) {
    if (B < 1 || M < 1 || S < 1 || cout < 1) return 0;
    const size_t tiles = ((size_t)B * M * S + 31) / 32;          // a tile t >= 1 may hold the continuation of one group: row t; row 0 stays zero
    return (tiles + 1) * (size_t)cout * 2;
}

SAD_API size_t sad_mlp_workspace_bytes(int B, int M, int S) {
    if (B < 1 || M < 1 || S < 1) return 0;
    const size_t ng = (size_t)B * M;
    // hdr, row_start, pass_first (R >= 32), block sums of the two-launch scan, row map, first packed row of every group (split pooling)
    return sizeof(int) * ((size_t)sad::scan_gstart_off((long long)ng, S) + ng + 1) + 64;
}
