// Weight repacking of the f32 grouped-MLP chains into MFMA A-fragment order and the geometry of the packed image
// (SPEC.md §6).  No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
#include "mlp_chain.h"

namespace {

using namespace sad::chain;

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
}  // namespace

namespace sad {
namespace chain {

Geometry geometry(int L, const int *dims, int first_has_xyz) {
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

}  // namespace chain
}  // namespace sad

using namespace sad::chain;

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
    if (wide) return 3;
    // Any other chain runs on the tiled kernel.  Its built-in heuristic (0) was made for dense rows; on grouped rows with counts
    // the autotuner's picks for chains that are not compiled shapes (tools/generality_bench.py, round 5: [32,32,64], [64,96,128],
    // [64,128], [128,196,256]) all have eight waves, one row tile per wave with the (output tile, row tile) items dealt
    // round-robin, workgroup-local row packing and 32 R / S groups per workgroup; the waves along the output tiles follow the
    // widest layer.  1.3 - 3 x the heuristic's throughput on those chains; the caller falls back to 0 if it does not fit LDS.
    int tiles = 1;
    for (int l = 0; l < L; ++l) tiles = g.np[l] / 32 > tiles ? g.np[l] / 32 : tiles;
    const int n = tiles >= 8 ? 3 : (tiles >= 4 ? 2 : (tiles >= 2 ? 1 : 0));
    return 100000 + 20000 + 5000 + 800 + 10 * n + 1;
}

// A compiled shape of the register-resident kernels that dominates the chain (include/sad_amd.h)
SAD_API int sad_mlp_padded_dims(int L, const int *dims, int *padded) {
    if (L != 3 || !dims || !padded) return 0;
    for (int l = 0; l <= L; ++l)
        if (dims[l] < 1 || dims[l] > 4096) return 0;
    if (dims[0] < 3) return 0;
    const int C = dims[0] - 3;
    if (!(C == 0 || C == 1 || C % 4 == 0)) return 0;
    {
        const Geometry g = geometry(L, dims, 1);
        if (sad::reg_shape_id(L, g.kp, g.np) >= 0) return 0;      // already a compiled shape
    }
    // logical dims of the compiled shapes (mlp_reg.hip, reg_shape_id): C + 3, C1, C2, C3
    static const int tab[8][4] = {{7, 16, 16, 32}, {7, 32, 32, 64}, {67, 64, 64, 128}, {67, 64, 96, 128},
                                  {131, 128, 128, 256}, {131, 128, 192, 256}, {131, 128, 256, 256}, {7, 64, 64, 128}};
    auto flops = [](const int *d) { return (double)d[0] * d[1] + (double)d[1] * d[2] + (double)d[2] * d[3]; };
    const double own = flops(dims);
    int best = -1;
    double best_f = 0.0;
    for (int i = 0; i < 8; ++i) {
        const int *t = tab[i];
        // the narrow shapes (7 -> ...) take C <= 4 as 0 / 1 strided channels or one 16-byte chunk; the others 16-byte chunks
        const bool feat_ok = t[0] == 7 ? (C == 0 || C == 1 || C == 4) : (C >= 4 && C % 4 == 0);
        if (!feat_ok || dims[0] > t[0] || dims[1] > t[1] || dims[2] > t[2] || dims[3] > t[3]) continue;
        const double f = flops(t);
        if (best < 0 || f < best_f) { best = i; best_f = f; }
    }
    if (best < 0 || best_f > 1.6 * own) return 0;
    for (int l = 0; l <= L; ++l) padded[l] = tab[best][l];
    return 1;
}
