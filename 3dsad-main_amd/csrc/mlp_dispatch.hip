// Validation, kernel choice and dispatch behind sad_mlp_chain_f32 / sad_mlp_chain_multi_f32 (SPEC.md §6): the tiled
// kernel (mlp_chain.hip), the register-resident chain (mlp_reg.hip), the layer-streamed chain (mlp_layer.hip), the
// cooperative register-resident chain (mlp_coop.hip) and the vector-ALU kernel.
// No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
#include "mlp_chain.h"

using namespace sad::chain;

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
    bool rows;          // geometry 5: row-streaming plain layer (csrc/mlp_rows.hip); `rj` is filled
    sad::RowsJob rj;
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
    q.rows = false;
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
        if (a->c_out)       // a chain packed zero-padded onto wider dims (sad_mlp_padded_dims): fewer feature channels than dims[0] - 3
            SAD_REQUIRE(a->C + 3 <= a->dims[0] && a->c_out >= 1 && a->c_out <= a->dims[a->L],
                        "sad_mlp_chain_f32: zero-padded chain needs C+3=%d <= dims[0]=%d and 1 <= c_out=%d <= dims[L]=%d", a->C + 3, a->dims[0], a->c_out, a->dims[a->L]);
        else
            SAD_REQUIRE(a->dims[0] == a->C + 3, "sad_mlp_chain_f32: dims[0]=%d != C+3=%d", a->dims[0], a->C + 3);
        SAD_REQUIRE((long long)a->B * a->N < (1LL << 31), "sad_mlp_chain_f32: B*N too large");
        SAD_REQUIRE((long long)a->B * a->M * a->S < (1LL << 31), "sad_mlp_chain_f32: B*M*S too large (row numbers are 32-bit)");
        // straddling groups are merged with an unsigned atomic max into a zero-initialised buffer:
        // only valid for non-negative outputs, i.e. a ReLU after every layer (SPEC.md §6 grouped chains)
        SAD_REQUIRE((a->relu_mask & ((1 << a->L) - 1)) == (1 << a->L) - 1,
                    "sad_mlp_chain_f32: grouped chains need a ReLU after every layer (relu_mask=0x%x, L=%d)", a->relu_mask, a->L);
    } else {
        SAD_REQUIRE(a->S == 1, "sad_mlp_chain_f32: plain mode needs S == 1");
        SAD_REQUIRE(a->c_out == 0, "sad_mlp_chain_f32: c_out (zero-padded chain) is a grouped-mode field");
        SAD_REQUIRE(a->C >= 1 && a->dims[0] == a->C, "sad_mlp_chain_f32: dims[0]=%d != C=%d", a->dims[0], a->C);
        SAD_REQUIRE((long long)a->B * a->M < (1LL << 31), "sad_mlp_chain_f32: too many rows");
    }
    const int cout = a->c_out ? a->c_out : a->dims[a->L];
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
    if (a->c_out && geom_wg != 2 && geom_wg != 4)
        return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: a zero-padded chain (c_out != 0) runs on geometries 2 / 4 only (geometry %d asked)", geom_wg);
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
        q.scan.zout = a->out + a->col_off;         // (a scan launched by the dispatch itself zero-fills the groups that need it)
        q.scan.zld = a->ld_out;
        q.scan.zcols = cout;                       // (a zero-padded chain: its own output channels only)
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
    // ---- geometry 5: row-streaming plain layer (one layer, every input row read once per 128 output channels) ----
    if (geom_wg == 5) {
        const bool ok = !grouped && a->L == 1 && p.cpr > 0 && a->C % 8 == 0 && g.kp[0] == a->C;
        if (!ok)
            return sad::fail(SAD_EUNSUPPORTED, "sad_mlp_chain_f32: geometry 5 (row-streaming layer) takes one plain layer with C %% 8 == 0 "
                                               "and 16-byte aligned rows");
        sad::RowsJob &j = q.rj;
        j = sad::RowsJob{};
        j.x = a->feat; j.ldx = a->ld_feat; j.rows = p.total_rows;
        j.packed = a->packed; j.off = g.off[0]; j.np = g.np[0]; j.kg = g.kp[0] / 8; j.ct = g.np[0] / 32; j.cout = cout;
        j.relu = a->relu_mask & 1;
        j.out = a->out; j.ld_out = a->ld_out; j.col_off = a->col_off; j.vec_out = p.vec_out;
        q.rows = true;
        q.W = -3;
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
        q.scan.zout = a->out + a->col_off;         // (a scan launched by the dispatch itself zero-fills the groups that need it)
        q.scan.zld = a->ld_out;
        q.scan.zcols = cout;                       // (a zero-padded chain: its own output channels only)
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
            if (int e = launch_valu(v, shape, nb, (hipStream_t)stream)) return e;
            q.launched = true;
            return SAD_OK;
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
    if (q.rows) return sad::launch_rows(q.rj, st);
    if (q.layered) {
        const Prepared *one = &q;
        return launch_layered_chains(&one, 1, st);
    }
    if (q.reg) {
        const Prepared *one = &q;
        return launch_reg_chains(&one, 1, st);
    }
    return launch_tiled(q.p, q.W, q.RW, q.CW, q.lds, q.nblocks, st);
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
    for (int i = 0; i < n; ++i) merge = merge && !q[i].launched && !q[i].reg && !q[i].layered && !q[i].rows && q[i].W == q[0].W && q[i].W != 16;
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
    return launch_tiled_multi(mp, q[0].W, rwmax, cw2, lds, st);
}
