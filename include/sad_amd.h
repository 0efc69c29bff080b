/*
 * sad_amd.h — C-ABI of libsad_amd.so: hand-written gfx950 (MI355X) HIP kernels for the
 * set-abstraction + size-adaptive-clustering hot path.
 *
 * Reference interface replaced: none exists.  The upstream reference is a two-line README
 * (/root/reference/README.md:1-2) and defines no operator/FFI surface; the entry points below are
 * the ones BASELINE.json north_star names ("fps / ball_query / group_points / sa_module ... through
 * a thin C-ABI extension"), with the semantics frozen in this repository's SPEC.md (section cited
 * per function).  The Python binding a maintainer would add is in INTEGRATION.md.
 *
 * Contract (all functions):
 *  - plain pointers and sizes only; no torch / C++ types cross the boundary;
 *  - every data pointer is DEVICE memory owned by the caller (inputs, outputs and workspace); the
 *    library never allocates, frees or retains device memory; pointer ARRAYS and small parameter
 *    arrays (dims, radii, nsamples) are HOST memory read before the call returns;
 *  - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) of the
 *    current device and the call returns without synchronising;
 *  - returns 0 on success, a negative SAD_E* code otherwise; sad_last_error() gives a
 *    thread-local message; nothing throws across the boundary; re-entrant, no global mutable
 *    state other than the tuning options below.
 */
#ifndef SAD_AMD_H
#define SAD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAD_OK 0
#define SAD_EINVAL (-1)       /* bad argument (NULL pointer, size out of range, ...) */
#define SAD_EUNSUPPORTED (-2) /* shape outside what the kernels implement */
#define SAD_ELAUNCH (-3)      /* HIP reported a launch error */

#define SAD_MAX_LAYERS 4
#define SAD_MAX_RADII 4

typedef void *sad_stream_t; /* hipStream_t */

/* ABI version: bumped whenever a public struct layout or a signature changes (2 = sad_mlp_args / sad_mlp_bf16_args
 * start with struct_size; 3 = sad_mlp_args.c_out in the former tail padding, sad_mlp_padded_dims, sad_copy_rows_u32;
 * 4 = split pooling: sad_mlp_bf16_args.cont / n_pool / pool_*, sad_mlp_rowscan_split, sad_mlp_cont_bytes, larger row-packing tables).  The structs additionally carry their own size: a caller built against another header is
 * refused with SAD_EINVAL instead of being read past its end. */
#define SAD_ABI_VERSION 4
int sad_version(void);
const char *sad_last_error(void);
/* Tuning / A-B knobs (process-wide; defaults 0 = automatic).  Returns SAD_EINVAL for an unknown key.
 *   fps_variant  1 pair kernel, 2 key kernel, 3 wave buckets, 4 cell buckets (second form of the kernel: the default for
 *                N <= 16384), 5 cell buckets over sorted records (default for 16384 < N <= 65536), 6 cell buckets, FIRST form of
 *                the kernel (lane-elected publish, five writelanes per bucket update: kept as the A/B baseline of round 4)
 *   fps_threads  cell-bucket geometry waves*100 + slots (e.g. 1616, 832); old kernels: 1024/512/256 threads
 *   fps_dpp      1 = DPP reductions in the pair kernel
 *   bq_variant   1 = grid query: always the LDS-bitmap path (no 64-lane sort for centroids with <= 64 candidates);
 *                2 = grid query: ONE LDS bitmap set per workgroup (round 3's layout) instead of one per two waves;
 *                grid vs scan is the caller's choice: sad_ball_query_grid_f32 / sad_ball_query_multi_f32
 *   group_variant 1 = L2-gather group_points kernel only (no LDS staging)
 *   mlp_rw, mlp_budget_kb, mlp_force, mlp_dedup_f, mlp_nodedup, mlp_static, mlp_dyn_slots, mlp_noxcd: f32 chain geometry overrides
 *                (see sad_mlp_args.geometry; mlp_nodedup = 1 computes the padding rows too)
 *   mlp_layer_queue 1 = the layer-streamed chain kernel (geometry 3) pulls its items from per-XCD queues owned by the
 *                launching stream instead of a static round-robin.  Beside an FPS kernel of another stream the cluster
 *                dispatch of the benchmark is 15 % shorter (905 vs 1 080 us), alone 2 % longer, and the pipelined step
 *                (two main streams always busy) 1.5 % slower: off by default.  Same results either way.
 *   mlp_rows_form 1 = the bf16 row-streaming layer (sad_mlp_chain_bf16, one plain layer) always in its first form (rows straight to
 *                registers, register-staged weights) instead of the tiled GEMM fed by LDS-DMA that serves bf16 rows with
 *                K % 64 == 0.  Same results either way (the same products in the same order).
 * Test knobs of the item queues of the cooperative chain kernel (geometry 4; tables of >= 257 groups):
 *   mlp_steal_after v > 0: a workgroup treats the queue of its own XCD as empty after v - 1 items and takes its items from
 *                the other queues, one at a time and with nothing prefetched, so the cross-queue / weight-ring refill
 *                path that normally runs only at the tail of a dispatch runs all the time (same results);
 *   mlp_check_inuse 1: every dispatch stamps its id into the queue header and flags a conflict when it finds another
 *                dispatch's id there (two dispatches sharing one workspace at the same time).
 * Instrumentation ints of a row-packing table (sad_mlp_args.workspace as int32[]; zeroed by the row-packing scan,
 * accumulated by the launches that reuse the table): [5] weight-ring refills, [6] id of the dispatch that owns the
 * queues right now (0 = none), [7] != 0: a conflict was seen. */
#define SAD_WS_REFILLS 5
#define SAD_WS_INUSE 6
#define SAD_WS_CONFLICT 7
int sad_set_option(const char *key, int value);

/* SPEC.md §2.  xyz[B,N,3] -> idx[B,M].  N < 2048 needs no workspace; otherwise pass
 * sad_fps_workspace_bytes(B,N) bytes of 16-byte aligned device workspace (Z-order permutation for
 * the bucketed kernels; + sorted point records for 16384 < N <= 65536).  Without a workspace,
 * N <= 16384 falls back to the plain register-resident scan. */
size_t sad_fps_workspace_bytes(int B, int N);
int sad_fps_f32(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace,
                sad_stream_t stream);

/* SPEC.md §15.  Feature-distance FPS: xyz[B,N,3], feat point-major [B,N,C]
 * with row stride ld_feat (floats) -> idx[B,M]; N <= 16384.  Two phases on the given stream: all
 * N x N distances in the exact §15 order (sad_pairdist_f32, parallel over the chip) into
 * `workspace` (sad_ffps_workspace_bytes(B,N) = B*N*N*4 bytes), then the serial sampling chain
 * reading one matrix row per step. */
size_t sad_ffps_workspace_bytes(int B, int N);
int sad_pairdist_f32(const float *xyz, const float *feat, int ld_feat, int B, int N, int C,
                     float w_xyz, float *dmat, sad_stream_t stream);
int sad_ffps_f32(const float *xyz, const float *feat, int ld_feat, int B, int N, int C, int M,
                 float w_xyz, int32_t *idx, void *workspace, sad_stream_t stream);

/* SPEC.md §5.  gather_xyz: xyz[B,N,3], idx[B,M] -> out[B,M,3].
 * gather_points: src[B,C,N], idx[B,M] -> out[B,C,M]; elem_size 2 or 4 bytes.
 * group_points:  feat[B,C,N], idx[B,M,S] -> out[B,C,M,S]; elem_size 2 or 4 bytes. */
int sad_gather_xyz_f32(const float *xyz, const int32_t *idx, int B, int N, int M, float *out,
                       sad_stream_t stream);
int sad_gather_points(const void *src, const int32_t *idx, int B, int C, int N, int M,
                      int elem_size, void *out, sad_stream_t stream);
int sad_group_points(const void *feat, const int32_t *idx, int B, int C, int N, int M, int S,
                     int elem_size, void *out, sad_stream_t stream);

/* SPEC.md §16: backward of the unfused operators.
 * group_points_grad: grad_out[B,C,M,S], idx[B,M,S] -> grad_feat[B,C,N] += scatter-add (float
 * atomics; grad_feat must be zero, or hold a gradient to accumulate into, on entry); S = 1 is
 * gather_points_grad.  max_pool_s: x[B,C,M,S] -> out[B,C,M], arg[B,C,M] (ties -> lowest s);
 * max_pool_s_grad: grad_out[B,C,M], arg -> grad_x[B,C,M,S]. */
int sad_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                              int S, float *grad_feat, sad_stream_t stream);
/* Same sum into a POINT-MAJOR gradient grad_feat_pm[B,N,C] (zero / accumulate on entry): channels on
 * the lanes, so the float atomics are 256 contiguous bytes per wave instruction (full memory-side
 * rate, ~16x the channel-major scatter).  Transpose afterwards if a [B,C,N] gradient is needed. */
int sad_group_points_grad_pm_f32(const float *grad_out, const int32_t *idx, int B, int C, int N,
                                 int M, int S, float *grad_feat_pm, sad_stream_t stream);
int sad_max_pool_s_f32(const float *x, int B, int C, int M, int S, float *out, int32_t *arg,
                       sad_stream_t stream);
int sad_max_pool_s_grad_f32(const float *grad_out, const int32_t *arg, int B, int C, int M, int S,
                            float *grad_x, sad_stream_t stream);

/* SPEC.md §3.  xyz[B,N,3], new_xyz[B,M,3] -> idx[B,M,S], 1 <= S <= 64.
 * radius_pc == NULL: scalar `radius`; else per-centroid radius_pc[B,M] (adaptive) and `radius`
 * is ignored. */
int sad_ball_query_f32(const float *xyz, const float *new_xyz, float radius,
                       const float *radius_pc, int B, int N, int M, int S, int32_t *idx,
                       sad_stream_t stream);
/* Multi-radius form: d2 evaluated once per pair, SPEC.md §3 applied per radius.  radii[n_radii] and
 * nsamples[n_radii] are host arrays, idx[n_radii] a host array of device pointers (idx[r] is
 * [B,M,nsamples[r]]).  With radius_pc != NULL the radius of branch r for centroid (b,m) is
 * radii[r] * radius_pc[b,m] (one binary32 multiply; SPEC.md §8 step 5).
 * cnt (may be NULL, entries may be NULL): host array of device pointers; cnt[r][b,m] receives the
 * number of accepted points capped at nsamples[r] — the rows of the group that are not padding. */
int sad_ball_query_multi_f32(const float *xyz, const float *new_xyz, int n_radii,
                             const float *radii, const float *radius_pc, const int *nsamples,
                             int32_t *const *idx, int32_t *const *cnt, int B, int N, int M,
                             sad_stream_t stream);

/* Same results as sad_ball_query_multi_f32 with scalar radii, computed through a per-scene uniform
 * grid (cell edge > max radius): each centroid tests only the 27 cells around it and index order is
 * restored through an LDS bitmap.  Needs sad_ball_query_grid_workspace_bytes(B,N) bytes of 16-byte
 * aligned device workspace (rebuilt on every call); N <= 65536. */
size_t sad_ball_query_grid_workspace_bytes(int B, int N);
int sad_ball_query_grid_f32(const float *xyz, const float *new_xyz, int n_radii, const float *radii,
                            const int *nsamples, int32_t *const *idx, int32_t *const *cnt, int B,
                            int N, int M, void *workspace, sad_stream_t stream);

/* SPEC.md §17 (the step before the path).  Ragged scenes -> fixed point count:
 * points[total, C] (C floats per point), offsets[B+1] (device, int32; scene b owns rows offsets[b] ..
 * offsets[b+1]-1) -> out[B, n_points, C].  Deterministic for (seed, scene); integer arithmetic only, so the
 * rows equal those of the numpy loader (io.fix_size) and of the oracle. */
int sad_subsample_pad_f32(const float *points, const int32_t *offsets, int B, int C, int n_points,
                          unsigned seed, float *out, sad_stream_t stream);

/* Strided row copy in 4-byte words (no arithmetic: exact): row r of dst = words [0, row_words) of row r of src, rows
 * src_stride_words / dst_stride_words apart (dst_stride_words >= row_words; src rows may overlap dst rows never).  What the
 * host side uses instead of a framework copy wherever a strided view has to become a packed operand on the step — the
 * coordinates / features of a [B,N,3+C] point array, the centroid prefix of a nested sampling stage, the candidate rows of the
 * cluster layer — so that every device operation of a step is a launch of this library (and can be recorded and replayed:
 * 3dsad-main_amd/plan.py).  No reference counterpart (/root/reference/README.md:1-2). */
int sad_copy_rows_u32(const void *src, long long src_stride_words, void *dst, long long dst_stride_words,
                      long long n_rows, long long row_words, sad_stream_t stream);

/* SPEC.md §4.  -> idx[B,M,K] sorted by (d2, index); K <= 64, K <= N. */
int sad_knn_f32(const float *xyz, const float *new_xyz, int B, int N, int M, int K, int32_t *idx,
                sad_stream_t stream);

/* ---- grouped point-feature MLP (SPEC.md §6) ------------------------------------------------
 * Weights are repacked once into MFMA A-fragment order.  dims[L+1] = {C_in, C_1, ..., C_L} (host).
 * first_has_xyz != 0: the first layer's input is [rel_xyz(3) || feat(C_in-3)].
 * W[l] is [dims[l+1], dims[l]] row-major, bias[l] is [dims[l+1]] (host arrays of device pointers).
 * `packed` needs sad_mlp_packed_floats() floats. */
size_t sad_mlp_packed_floats(int L, const int *dims, int first_has_xyz);
int sad_mlp_pack_f32(int L, const int *dims, int first_has_xyz, const float *const *W,
                     const float *const *bias, float *packed, sad_stream_t stream);

typedef struct sad_mlp_args {
    size_t struct_size;   /* = sizeof(sad_mlp_args) of the header the caller was built against */
    /* grouped mode (idx != NULL): rows are (b, m, s); input row = [xyz[idx]-new_xyz || feat[idx]] */
    const float *xyz;     /* [B,N,3]                                   (grouped mode) */
    const float *new_xyz; /* [B,M,3]                                   (grouped mode) */
    const int32_t *idx;   /* [B,M,S] or NULL                                          */
    /* optional [B,M]: leading rows of each group that are not padding (from ball query); NULL =
     * derived from idx as "last sample that differs from the first, + 1" */
    const int32_t *cnt;
    /* optional, grouped mode with cnt: sad_mlp_workspace_bytes(B, M, S) bytes of 16-byte aligned
     * device scratch.  The surviving rows of ALL groups are then packed globally (one prefix-sum
     * workgroup) and a persistent grid pulls full passes from a work counter; without it each
     * workgroup packs only its own groups.  The table also holds the work counters / item queues of the
     * kernel that consumes it, so ONE dispatch at a time may use a given workspace (dispatches that run
     * side by side on different streams need a workspace each; a prescanned table may be reused by
     * consecutive launches of geometry 2 / 3 / 4 on one stream — they re-arm what they use). */
    void *workspace;
    /* features: grouped mode: point-major [B,N,C] with row stride ld_feat (NULL iff C == 0);
     * plain mode (idx == NULL): rows [B*M, C] with row stride ld_feat */
    const float *feat;
    int ld_feat;
    int B, N, M, S, C; /* plain mode: N ignored, S must be 1 */
    /* layers */
    int L;                        /* 1..SAD_MAX_LAYERS */
    int dims[SAD_MAX_LAYERS + 1]; /* dims[0] = C (+3 in grouped mode) */
    const float *packed;          /* from sad_mlp_pack_f32 with the same L, dims, first_has_xyz */
    int relu_mask;                /* bit l set = ReLU after layer l */
    /* output: point-major rows.  grouped mode: max over the S samples of each (b,m) ->
     * out[(b*M+m)*ld_out + col_off + o]; plain mode: out[row*ld_out + col_off + o].
     * GROUPED MODE NEEDS THE OUTPUT SLICE ZERO-INITIALISED: groups whose rows straddle two row tiles
     * are combined with an atomic max (every layer of a grouped chain must carry a ReLU, so outputs
     * are >= 0).  Trailing samples that repeat a group's first index (ball-query padding) are not
     * computed at all — a duplicate row cannot change the max. */
    float *out;
    int ld_out;
    int col_off;
    /* workgroup geometry: 0 = built-in heuristic; else W*100 + log2(WN)*10 + RW with W in {4,8}
     * waves, WN waves along the 32-channel output tiles, RW in {1,2,4} row tiles of 32 rows per
     * wave.  A geometry that does not fit LDS returns SAD_EUNSUPPORTED (autotuners skip it).
     * + 1000*f (f = 1..7): grouped mode, a workgroup owns 2^f * R / S groups (default 8: it assumes
     * about one row in eight survives the padding removal).
     * + 10000*d: d = 1 forces global row packing (needs cnt + workspace), d = 2 forbids it.
     * + 100000: with RW = 1, the (output tile, row tile) items of every layer are dealt round-robin
     *   to all W waves instead of the fixed WN x WM grid (no wave idles in a layer narrower than WN tiles).
     * + 200000: with RW = 1 (4 or 8 waves), a wave computes two output tiles per round as two
     *   independent accumulator chains sharing the activation operand (+300000 = both).
     * Other kernels (grouped mode with cnt + workspace; SAD_EUNSUPPORTED where they do not apply):
     *   1 = row-per-lane vector-ALU kernel (two narrow SA1 shapes);
     *   2 = register-resident chain: one wave carries a 32-row tile through a 3-layer chain in registers
     *       (compiled shapes: the SA stages of the KITTI / nuScenes topology and BASELINE configs[0]);
     *   3 = layer-streamed chain: one launch per layer, (128 rows x 128 output channels) work items,
     *       activations between layers in `scratch` (wide chains with few rows: the cluster layer); also
     *       on plain rows (idx == NULL) when C % 8 == 0 and every width, C_out included, is a multiple of 128;
     *   4 = cooperative register-resident chain: as 2, but the four waves of a workgroup walk four tiles
     *       through the chain in lockstep and share the weight stream through an LDS ring (one global load per
     *       16 MFMAs instead of one per 4; compiled for the SA2 / SA3 shapes 67 -> 64 -> {64,96} -> 128 and
     *       131 -> 128 -> {128,192,256} -> 256);
     *   5 = row-streaming layer: ONE plain layer (idx == NULL, L == 1, C % 8 == 0, 16-byte aligned rows): a wave owns 32 rows
     *       and all output channels of its item, its input rows never pass through LDS (the stage aggregations). */
    int geometry;
    /* geometry 3 only: sad_mlp_scratch_bytes(B, M, S, L, dims) bytes of 16-byte aligned device scratch */
    void *scratch;
    size_t scratch_bytes;
    /* != 0: `workspace` already holds the row-packing table of (cnt, idx) from sad_mlp_rowscan (geometries 2,
     * 3 and 4 only): the chain launches no scan of its own.  The scan needs coordinates-side data only, so a
     * caller can run it right behind the ball query on another stream, off the MLP stream's critical path. */
    int prescanned;
    /* ABI 3 (occupies what was tail padding of the ABI-2 struct: sizeof is unchanged, a zeroed ABI-2 struct means 0).
     * != 0: the chain was packed ZERO-PADDED onto wider dims (sad_mlp_padded_dims): `dims` are the padded ones, C + 3 <= dims[0]
     * feature channels exist per row and only the first c_out <= dims[L] output channels are stored.  The padded weights
     * and biases are zeros, so every fmaf chain gains only exact +0 terms behind its real ones: same results.  Grouped mode,
     * geometries 2 and 4 (register-resident / cooperative chain) only — that is what it is for: a chain that is not a
     * compiled shape runs on the compiled shape that dominates it. */
    int c_out;
} sad_mlp_args;
size_t sad_mlp_workspace_bytes(int B, int M, int S);
/* Row-packing tables of n (<= SAD_MAX_RADII) chains over the same (B, N, M): cnt[i] [B,M] and idx[i] [B,M,S[i]] from the
 * ball query -> workspace[i] (sad_mlp_workspace_bytes(B, M, S[i]) bytes each, 16-byte aligned).  Two launches
 * for all n.  Pass the workspaces to sad_mlp_chain_f32 with prescanned = 1. */
int sad_mlp_rowscan(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                    int M, void *const *workspace, sad_stream_t stream);
/* The same, and it prepares the pooled-output buffers so that they need NO zero fill: for chain i, columns
 * [col_off[i], col_off[i] + cout[i]) of the rows out[i][(b*M+m)*ld_out[i] + ...] are set to zero for exactly the groups
 * whose packed rows straddle a 32-row tile — the only ones the chain kernels of geometries 2 / 3 / 4 (bf16: 2) combine
 * with an atomic max; every other group is written with plain stores.  (A 32-scene KITTI-shaped step otherwise zero-fills
 * 217 MB of pooling buffers.)  A scan that sad_mlp_chain[_multi]_f32 / _bf16 launches itself (prescanned == 0, geometries
 * 2 / 3 / 4) does the same with the call's own `out`.  The tiled kernels (geometry 0 / tile heights) still need `out`
 * zeroed by the caller. */
int sad_mlp_rowscan_init(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                         int M, void *const *workspace, float *const *out, const int *ld_out, const int *col_off,
                         const int *cout, sad_stream_t stream);
size_t sad_mlp_scratch_bytes(int B, int M, int S, int L, const int *dims);
/* The `geometry` a caller that does not autotune should pass for a GROUPED chain of this shape when it provides
 * cnt + workspace (+ scratch for 3): 4 (cooperative register-resident chain: the SA3 shapes), 2 (register-resident
 * chain: the other compiled shapes), 3 (layer-streamed chain: every padded width a multiple of 128), else a code of the tiled
 * kernel (eight waves, round-robin items, workgroup-local row packing: what the autotuner picks for chains that are not
 * compiled shapes; a caller retries with 0 = built-in heuristic when it is refused with SAD_EUNSUPPORTED: LDS).  Matches what
 * the autotuner picks on the KITTI-shaped benchmark; never 0 for a valid chain. */
int sad_mlp_preferred_geometry(int L, const int *dims);
/* Is there a compiled shape of the register-resident chain kernels that DOMINATES this grouped 3-layer chain (every padded
 * width >= the chain's) at no more than 1.6 x its flops?  Returns 1 and the dims to pack for in padded[0..L] (pack the chain's
 * weights zero-padded to them, call with sad_mlp_args.dims = padded, C = the true feature channels, c_out = the true output
 * width), else 0.  C must be 0, 1 or a multiple of 4 (the kernels' feature-row layouts).  A chain that IS a compiled shape
 * returns 0 (nothing to pad). */
int sad_mlp_padded_dims(int L, const int *dims, int *padded);
int sad_mlp_chain_f32(const sad_mlp_args *args, sad_stream_t stream);
/* n independent chains (typically the branches of one multi-radius stage, each writing its own
 * column slice) in one dispatch when they share a wave count: the light chains fill the tail of the
 * heavy one and the launch gaps disappear.  Same results as n sad_mlp_chain_f32 calls. */
int sad_mlp_chain_multi_f32(const sad_mlp_args *const *args, int n, sad_stream_t stream);

/* SPEC.md §14 — the same chain in bfloat16 on the matrix cores (BASELINE.json configs[4]).
 * Weights are rounded to bf16 and laid out in MFMA fragment order by sad_mlp_pack_bf16 (W/bias as
 * for sad_mlp_pack_f32; `packed` needs sad_mlp_packed_bytes_bf16() bytes, 16-byte aligned).
 * Fields as in sad_mlp_args, except: feat is bf16 (feat_bf16 = 1) or f32 (0, rounded on load) with
 * ld_feat in ELEMENTS; grouped output is f32 and must be zero on entry (atomic max merge); plain
 * output is f32 or bf16 (out_bf16). */
size_t sad_mlp_packed_bytes_bf16(int L, const int *dims, int first_has_xyz);
int sad_mlp_pack_bf16(int L, const int *dims, int first_has_xyz, const float *const *W,
                      const float *const *bias, void *packed, sad_stream_t stream);
typedef struct sad_mlp_bf16_args {
    size_t struct_size;   /* = sizeof(sad_mlp_bf16_args) */
    const float *xyz;     /* [B,N,3] f32                               (grouped mode) */
    const float *new_xyz; /* [B,M,3] f32                               (grouped mode) */
    const int32_t *idx;   /* [B,M,S] or NULL (plain mode)                             */
    const void *feat;     /* grouped: point-major [B,N,C]; plain: rows [B*M, C]        */
    int feat_bf16;
    int ld_feat;
    int B, N, M, S, C;
    int L;
    int dims[SAD_MAX_LAYERS + 1];
    const void *packed;
    int relu_mask;
    void *out;
    int out_bf16;
    int ld_out;
    int col_off;
    /* optional, grouped mode: cnt[B,M] from the ball query + sad_mlp_workspace_bytes(B,M,S) bytes of
     * 16-byte aligned scratch -> only the leading cnt rows of each group are computed (rows that
     * repeat the first neighbour cannot change the max); both NULL = dense rows */
    const int32_t *cnt;
    void *workspace;
    /* 0 = built-in choice (128 rows per tile, fewer if LDS demands); else rows per tile 32/64/128/256
     * (more rows amortise the per-tile gather latency of narrow chains; SAD_EUNSUPPORTED if LDS is short);
     * 2 = register-resident chain (grouped 3-layer chains with cnt + workspace whose shape is compiled: the SA stages
     *     and the cluster layer of the KITTI / nuScenes / TINY topologies and configs[0]; 16-byte bf16 feature rows,
     *     or at most 13 feature channels of either type): one wave carries a 32-row tile through the chain in
     *     registers, weights through an LDS ring, pooled rows staged in LDS and stored once per group;
     * 3 = row-streaming layer (plain single layers: C % 8 == 0, 16-byte aligned rows; also what 0 picks for them):
     *     every input row is read once per 128 output channels, weights shared through LDS */
    int geometry;
    /* != 0: `workspace` already holds the row-packing table of (cnt, idx) from sad_mlp_rowscan (geometry 2 only) */
    int prescanned;
    /* Split pooling (ABI 4; grouped mode, geometry 2, out_bf16 = 1): the pooled rows leave as bf16 with PLAIN stores and nothing
     * needs a zero fill.  A group whose packed rows lie in several 32-row tiles (about one in eight) is not combined in memory:
     * its rows in the tile where they begin pool into out[g], its rows in a later tile t into row t of `cont`
     * (sad_mlp_cont_bytes() bytes, 16-byte aligned, row stride = dims[3] elements; row 0 is all zero).  The true pooled row is
     * the element-wise maximum of the two or three — taken by the layer that reads them: a plain-mode call with n_pool > 0.
     * Exact: rounding to bf16 is monotone, so max(bf16(a), bf16(b)) = bf16(max(a, b)), and every consumer of pooled rows
     * rounds them to bf16 on load.  Needs dims[3], ld_out and col_off multiples of 8, `out` 16-byte aligned, and a table made
     * by this call's own scan (prescanned = 0) or by sad_mlp_rowscan_split. */
    void *cont;
    /* plain mode, one layer (geometry 0 / 3), feat_bf16 = 1: the input rows are the split-pooled outputs of n_pool chains side
     * by side — columns of width pool_cols[i] (multiples of 16, adding up to C) pooled by the chain with nsample pool_S[i],
     * row-packing table pool_ws[i] and continuation rows pool_cont[i]; B * M = the chains' groups.  0 = ordinary rows. */
    int n_pool;
    const void *pool_ws[SAD_MAX_RADII];
    const void *pool_cont[SAD_MAX_RADII];
    int pool_S[SAD_MAX_RADII];
    int pool_cols[SAD_MAX_RADII];
} sad_mlp_bf16_args;
/* bytes of the continuation-row buffer of a split-pooled chain with B * M groups of at most S rows and cout output channels */
size_t sad_mlp_cont_bytes(int B, int M, int S, int cout);
/* sad_mlp_rowscan for chains whose pooled output is split (sad_mlp_bf16_args.cont): also marks the continuation rows in the row
 * map, records the first packed row of every group behind it and zeroes row 0 of cont[i] (cout[i] = the chain's output channels). */
int sad_mlp_rowscan_split(int n, const int32_t *const *cnt, const int32_t *const *idx, const int *S, int B, int N,
                          int M, void *const *workspace, void *const *cont, const int *cout, sad_stream_t stream);
/* 2 when sad_mlp_chain_bf16 has a register-resident kernel for this grouped chain (dims[0] = C + 3), else 0 */
int sad_mlp_preferred_geometry_bf16(int L, const int *dims);
int sad_mlp_chain_bf16(const sad_mlp_bf16_args *args, sad_stream_t stream);
/* n independent bf16 chains in one dispatch (see sad_mlp_chain_multi_f32). */
int sad_mlp_chain_multi_bf16(const sad_mlp_bf16_args *const *args, int n, sad_stream_t stream);

/* SPEC.md §8 steps 2-4.  xyz3[B,M3,3], c[B,K,6] -> cand[B,K,3], radius[B,K]; anchor[3] host. */
int sad_candidates_f32(const float *xyz3, const float *c, int B, int M3, int K, float shift_max,
                       float r_min, float r_max, const float *anchor, float *cand, float *radius,
                       sad_stream_t stream);
/* SPEC.md §9.  cand[B,K,3], o[B,K,10] -> boxes[B,K,9]; anchors[9] host (3 classes x lwh). */
int sad_decode_boxes_f32(const float *cand, const float *o, int B, int K, const float *anchors,
                         float *boxes, sad_stream_t stream);

/* SPEC.md §13.  Rotated-box NMS in bird's-eye view, one scene per workgroup.
 * boxes[B,K,9] (x,y,z,l,w,h,yaw,score,label), K <= 512 -> keep[B,K] (0/1), order[B,K] (kept indices
 * in rank order, -1 padded), count[B]. */
int sad_nms_bev_f32(const float *boxes, int B, int K, float iou_thr, float score_thr, int32_t *keep,
                    int32_t *order, int32_t *count, sad_stream_t stream);
/* Same result with sad_nms_bev_workspace_bytes(B,K) bytes of 16-byte aligned scratch: the K x K
 * suppression matrix is computed by B*K waves spread over the chip instead of one workgroup per scene. */
size_t sad_nms_bev_workspace_bytes(int B, int K);
int sad_nms_bev_ws_f32(const float *boxes, int B, int K, float iou_thr, float score_thr, int32_t *keep,
                       int32_t *order, int32_t *count, void *workspace, sad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAD_AMD_H */
