// Shared host-side helpers for the libsad_amd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sad_amd.h"

#define SAD_API extern "C" __attribute__((visibility("default")))

namespace sad {

void set_error(const char *fmt, ...);
int get_option(int which);
// row-packing scan of one chain / of up to SAD_MAX_RADII chains in the same two launches (csrc/mlp_chain.hip)
struct ScanJob {
    const int32_t *cnt, *idx;
    int *tab, *blk_sum, *row_src, *row_gid;
    int ngroups, S, N, M, nodedup, R, blk0;
    // optional: the pooled-output slice of this chain, [ngroups][zld] floats, columns 0 .. zcols-1 of which the scan zero-fills
    // for the groups the MLP kernels combine with an atomic max (packed rows straddling a 32-row tile: WHOLE_BIT clear) — the
    // others are overwritten by plain stores, so the caller need not initialise the buffer at all
    float *zout;
    int zld, zcols;
    // split pooling (bf16 mode, sad_mlp_rowscan_split): nothing is zero-filled; the scan marks in the row map the rows whose group
    // began in an EARLIER 32-row tile (CONT_BIT), writes the first packed row of every group (gstart[ngroups + 1], what the layer
    // that consumes the pooled rows finds a group's continuation rows with) and zeroes row 0 of the chain's continuation buffer
    int split;
    int *gstart;
    void *cont0;
    int cont_cols;
};
// ints of a row-packing table in front of gstart (hdr, row_start, pass_first, block sums, row map: make_scan_job)
inline long long scan_gstart_off(long long ng, int S) {
    const long long o = 4 + (ng + 1) + (ng * S / 32 + 2) + (ng / 1024 + 2) + 2 * ng * S;
    return (o + 3) & ~3LL;
}
constexpr int CONT_BIT = 1 << 29;          // row map: this row's group began in an earlier tile (split pooling only)
constexpr int GID_MASK = CONT_BIT - 1;
constexpr int SCAN_MAX_CHAINS = SAD_MAX_RADII;     // the branches of one multi-radius stage
struct ScanMulti { ScanJob j[SCAN_MAX_CHAINS]; int n; };
ScanJob make_scan_job(const int32_t *cnt, int ngroups, int S, int R, int *tab, int nodedup, const int32_t *idx, int N, int M);
int launch_rowscan_multi(const ScanJob *jobs, int n, hipStream_t st);
int launch_rowscan(const int32_t *cnt, int ngroups, int S, int R, int *tab, hipStream_t st, int nodedup = 0,
                   const int32_t *idx = nullptr, int N = 0, int M = 1);
int launch_fps_cellg(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace, hipStream_t st);
int launch_fps_bucket(const float *xyz, int B, int N, int M, int32_t *idx, void *workspace, hipStream_t st);
// Register-resident chain kernel (csrc/mlp_reg.hip): one chain of a dispatch / a dispatch of up to
// REG_MAX_CHAINS chains of one shape family (prepared in mlp_chain.hip, launched by launch_reg).
constexpr int REG_MAX_CHAINS = 3;
struct RegChain {
    const float *xyz, *new_xyz, *feat, *packed;
    float *out;
    const int *rowtab;             // hdr of the row-packing table: [0] = packed rows
    const int *row_src, *row_gid;  // row map (see rowscan_kernel)
    long long off[3];              // float offset of layer l inside `packed` (bias block, then A fragments)
    long long stream_off;          // float offset of the stream image (mlp_coop.hip), -1 if the shape has none
    int np[3];                     // padded output channels of layer l
    int ld_feat, C, cpr;           // cpr = C / 4 when feature rows are read as 16-byte chunks, else 0
    int ld_out, col_off, cout_last, vec_out;
};
struct RegMulti {
    RegChain c[REG_MAX_CHAINS];
    int shape[REG_MAX_CHAINS];
    int n;
    int *counter;                  // zeroed item queues (see ItemQueue; tiles of all chains form one list)
    int nq;                        // 8: one queue per XCD, 1: a single counter
    long long max_tiles;           // upper bound of the tile count (grid sizing)
    int steal_after;               // test knob mlp_steal_after (0 = off): see launch_coop
    int check_id;                  // test knob mlp_check_inuse: this dispatch's id for the in-use marker (0 = off)
};
// Layer-streamed chain kernel (csrc/mlp_layer.hip): one layer of one chain / one layer of up to two chains
struct LayerJob {
    const float *x;                // layer input: feature matrix (gather) or the previous layer's row-major output
    const float *xyz, *new_xyz;    // gather only
    const int *rowtab, *row_src, *row_gid;   // rowtab == NULL: plain rows, `rows` of them (no row map, no pooling)
    int rows;
    const float *packed;
    long long off;                 // float offset of this layer inside `packed`
    int ldx, np, kg, nog;          // input row stride, padded output channels, k-groups of 8, groups of 4 output tiles
    int cpr, gather, relu, last;
    float *y;                      // hidden layer: row-major output [rows][ldy]
    int ldy;
    float *out;                    // last layer: pooled output
    int ld_out, col_off, cout_last;
};
struct LayerMulti {
    LayerJob j[2];
    int n;
    int *queue;                    // zeroed item queues of the launching stream (ItemQueue, eight counters), NULL: static deal
};
int launch_layers(const LayerMulti &lm, long long max_items, hipStream_t st);
// ITEMQ_INTS zeroed ints owned by (device, stream): launches of one stream run one after the other, and the last workgroup
// of a launch re-arms the queues, so one set per stream is enough.  NULL while the stream is being captured into a graph
// (no allocation then) or when the allocation fails: the caller falls back to a static deal.
int *stream_item_queue(hipStream_t st);
// Row-streaming plain f32 layer (csrc/mlp_rows.hip, geometry 5)
struct RowsJob {
    const float *x;                // rows [rows, ldx]
    int ldx;
    long long rows;
    const float *packed;
    long long off;                 // float offset of the layer inside `packed` (bias block, then A fragments)
    int np, kg, ct, cout;          // padded output channels, k-groups of 8 (C / 8), channel tiles of 32, true output channels
    int relu;
    float *out;
    int ld_out, col_off, vec_out;
    int nrb, ncb;                  // (filled by launch_rows) row blocks of 128, channel blocks
};
int launch_rows(const RowsJob &job, hipStream_t st);
int reg_shape_id(int L, const int *kp, const int *np);   // -1: no compiled shape
int reg_family(int shape);
int launch_reg(const RegMulti &mp, hipStream_t st);
// Cooperative variant (csrc/mlp_coop.hip, geometry 4): the four waves of a workgroup share the weight stream through LDS
bool coop_shape(int shape);
long long coop_stream_frags(int shape, const int *kp, const int *np);   // fragments (1 KB) of the stream image, 0 = none
int launch_coop(const RegMulti &mp, hipStream_t st);
// Register-resident bf16 chain kernel (csrc/mlp_bf16_reg.hip): one chain of a dispatch / up to REG_MAX_CHAINS chains
struct BfRegChain {
    const float *xyz, *new_xyz;
    const void *feat;              // point-major rows, bf16 (feat_bf16) or f32
    int feat_bf16, ld_feat, C;     // ld_feat in elements
    const void *stream;            // fragment stream image (sad_mlp_pack_bf16)
    int stream_frags;              // its 1-KB fragments (filled by launch_bfreg)
    const float *bias[3];          // padded to multiples of 32
    int np[3];                     // padded output channels of layer l
    float *out;
    int ld_out, col_off, cout_last;
    const int *rowtab, *row_src, *row_gid;
    // split pooling: `out` holds bf16 rows; a group's rows in the tile where it begins pool into out[g], its rows in a later tile t
    // into cont[t] (row stride ld_cont elements) — every pooled row leaves with plain stores, nothing is combined in memory
    int out_bf16, ld_cont;
    void *cont;
};
struct BfRegMulti {
    BfRegChain c[REG_MAX_CHAINS];
    int shape[REG_MAX_CHAINS];
    int n;
    long long max_tiles;
    int static_f4;                 // family 0: float4 of all chains' stream images held in LDS (filled by launch_bfreg)
};
// One plain-row bf16 layer (csrc/mlp_bf16_rows.hip)
struct BfRowsJob {
    const void *x;                 // rows [rows, ldx], bf16 (x_bf16) or f32
    int x_bf16, ldx, kin;          // kin = input channels (a multiple of 8)
    long long rows;
    const void *w;                 // fragment image of the layer: [channel tile][k-step][lane] x 8 bf16
    const float *bias;             // padded to a multiple of 32
    int ks, ct, cout;              // k-steps of 16, channel tiles of 32, true output channels
    int relu;
    void *out;                     // rows [rows, ld_out], f32 or bf16 (out_bf16)
    int out_bf16, ld_out, col_off, vec_out;
    int nrb, ncb;                  // (filled by launch_bf16_rows) row blocks of 128, channel blocks of 128
    // x = split-pooled rows of n_pool chains side by side (columns pool_col0[i] .. pool_col0[i + 1] - 1 from chain i): row g of chain i
    // is max(x[g], cont_i[t]) over the tiles t after the one its packed rows begin in (pool_gstart[i][g] >> 5)
    int n_pool;
    const int *pool_gstart[SAD_MAX_RADII];
    const void *pool_cont[SAD_MAX_RADII];
    int pool_col0[SAD_MAX_RADII + 1];
    int pool_ld[SAD_MAX_RADII];
};
int launch_bf16_rows(const BfRowsJob &job, hipStream_t st);
int bfreg_shape_id(int L, const int *dims);        // dims = {C + 3, C1, C2, C3}; -1: no compiled shape
int bfreg_family(int shape);
long long bfreg_stream_frags(int shape);           // 1-KB fragments of the stream image (whole stages)
int bfreg_pack(int shape, const int *dims, int first_has_xyz, const float *const *W, void *dst, hipStream_t st);
int launch_bfreg(const BfRegMulti &mp, hipStream_t st);
enum { OPT_FPS_DPP = 0, OPT_MLP_RW = 1, OPT_MLP_BUDGET_KB = 2, OPT_BQ_VARIANT = 3, OPT_FPS_VARIANT = 4, OPT_MLP_FORCE = 5, OPT_MLP_DEDUP_F = 6, OPT_MLP_NODEDUP = 7, OPT_FPS_THREADS = 8, OPT_MLP_STATIC = 9, OPT_GROUP_VARIANT = 10, OPT_MLP_DYN_SLOTS = 11, OPT_MLP_NOXCD = 12, OPT_MLP_STEAL_AFTER = 13, OPT_MLP_CHECK_INUSE = 14, OPT_MLP_LAYER_QUEUE = 15, OPT_MLP_ROWS_FORM = 16, OPT_COUNT };

inline int fail(int code, const char *fmt, ...) {
    char buf[480];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device).  `done` is the call
// site's mask of devices already served.  Thread-safe (two threads racing set the same value twice,
// which is harmless) and per device (a process that drives several GPUs gets the attribute on each);
// a refused attribute is not allowed to poison the launch check — the launch then reports it.
// Compute units of the current device, asked once per process (the devices of a node are the same part; the launchers that size a
// persistent grid used to ask at every launch: two runtime calls per dispatch on a path whose bf16 steps are within ~25 % of the
// host's enqueue time)
inline int device_cus() {
    static std::atomic<int> cached{0};
    int c = cached.load(std::memory_order_relaxed);
    if (c > 0) return c;
    int dev = 0;
    c = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) c = v;
    }
    (void)hipGetLastError();
    cached.store(c, std::memory_order_relaxed);
    return c;
}

inline void lds_attr_once(std::atomic<uint64_t> &done, const void *fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) (void)hipGetLastError();
    done.fetch_or(bit, std::memory_order_release);
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SAD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SAD_OK;
}

// ---- work-item queues of the persistent MLP kernels (mlp_coop.hip, mlp_layer.hip) ------------------------------
// Items are dealt dynamically: a static deal is ~1 % faster on an empty chip, but an FPS kernel of another stream (32
// workgroups x 1024 threads for 2.8 ms) leaves room for fewer of these workgroups on its 32 CUs; the ones that do
// not fit start when others finish, and with a static deal they still own a full share of the items (+40 % on the
// SA3 dispatch, tools/fps_overlap_probe.py).  Returning atomics on ONE address are served at ~30 ns each chip-wide,
// so there are eight queues, one per XCD: item i belongs to queue i % 8, a workgroup pulls from the queue of the XCD
// it runs on (HW_REG_XCC_ID: correct under any placement) and, when that queue is empty, looks at the others
// (plain loads first: a stale count is a smaller count, so a non-empty queue is never missed) — an XCD that received
// no workgroup has its items taken by the rest.  Layout (ints, in the header of the row-packing table): q[0] the
// single counter used when nq == 1 (small tables / grids), q[1] workgroups finished (the last one out re-arms
// everything for the next launch), q[2 + 32 x] the counter of XCD x (one 128-byte line each).
// With nq == 8 three more ints of the header are test instrumentation (zeroed by the row-packing scan, never re-armed:
// they accumulate over the launches that reuse a table; ints 5, 6, 7 of the workspace, see include/sad_amd.h):
// q[ITEMQ_REFILLS] weight-ring refills of the cooperative kernel (an item of another chain than the one prefetched),
// q[ITEMQ_INUSE] id of the dispatch that owns the queues right now (mlp_check_inuse), q[ITEMQ_CONFLICT] set when a
// dispatch found another one's id there.
constexpr int ITEMQ_REFILLS = 3, ITEMQ_INUSE = 4, ITEMQ_CONFLICT = 5;
struct ItemQueue {
    int *q;
    int nq, own;
};
constexpr int ITEMQ_INTS = 2 + 32 * 8;     // ints behind `q` when nq == 8
__device__ __forceinline__ ItemQueue itemq_init(int *q, int nq) {
    unsigned xcc = 0;
    if (nq == 8) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ItemQueue{q, nq, (int)(xcc & 7u)};
}
__device__ __forceinline__ int *itemq_counter(const ItemQueue &Q, int x) { return Q.nq == 8 ? Q.q + 2 + 32 * x : Q.q; }
__device__ __forceinline__ int itemq_item(const ItemQueue &Q, int x, int pos) { return Q.nq == 8 ? pos * 8 + x : pos; }
// next item of the own queue (may be >= nitems: queue empty); returning atomic: issue it early, use it late
__device__ __forceinline__ int itemq_pull(const ItemQueue &Q, int n = 1) { return atomicAdd(itemq_counter(Q, Q.own), n); }
// the own queue is empty: an item of another queue, or `nitems`
__device__ __forceinline__ int itemq_steal(const ItemQueue &Q, int nitems) {
    if (Q.nq != 8) return nitems;
    int seen[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)      // (seven independent loads: one round trip, not seven)
        seen[k] = __hip_atomic_load(itemq_counter(Q, (Q.own + 1 + k) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int x = (Q.own + 1 + k) & 7;
        if (itemq_item(Q, x, seen[k]) >= nitems) continue;
        const int it = itemq_item(Q, x, atomicAdd(itemq_counter(Q, x), 1));
        if (it < nitems) return it;
    }
    return nitems;
}
// one thread per workgroup, after its last pull has returned
__device__ __forceinline__ void itemq_done(const ItemQueue &Q, int workgroups) {
    if (atomicAdd(Q.q + 1, 1) != workgroups - 1) return;
    Q.q[0] = 0;
    Q.q[1] = 0;
    if (Q.nq == 8) {
        for (int x = 0; x < 8; ++x) Q.q[2 + 32 * x] = 0;
        Q.q[ITEMQ_INUSE] = 0;
    }
}
// test knob mlp_check_inuse: one thread per workgroup, before its first pull — claims the queues for dispatch `id`
__device__ __forceinline__ void itemq_claim(const ItemQueue &Q, int id) {
    if (Q.nq != 8 || id == 0) return;
    const int old = atomicCAS(Q.q + ITEMQ_INUSE, 0, id);
    if (old != 0 && old != id) atomicExch(Q.q + ITEMQ_CONFLICT, 1);
}

// SPEC.md §1 — the one squared-distance expression every index decision uses.  The library is
// built with -ffp-contract=off so none of these multiplies/adds is fused.
__device__ __forceinline__ float d2f(float px, float py, float pz, float cx, float cy, float cz) {
    float dx = px - cx, dy = py - cy, dz = pz - cz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

}  // namespace sad

#define SAD_REQUIRE(cond, ...)                                \
    do {                                                      \
        if (!(cond)) return sad::fail(SAD_EINVAL, __VA_ARGS__); \
    } while (0)
