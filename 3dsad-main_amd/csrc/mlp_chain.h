// Internal interface between the translation units of the f32 grouped-MLP path (SPEC.md §6):
//   mlp_chain.hip    the tiled kernel (mlp_chain_kernel / mlp_multi_kernel), the vector-ALU kernel, their launchers
//   mlp_rowscan.hip  global row packing (prefix sum of the per-group counts + row map)
//   mlp_pack.hip     weight repacking into MFMA fragment order, packed-image geometry
//   mlp_dispatch.hip validation, kernel choice and dispatch behind sad_mlp_chain_f32 / sad_mlp_chain_multi_f32
// No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
#pragma once
#include "common.h"

namespace sad {
namespace chain {

constexpr int MAXL = SAD_MAX_LAYERS;
constexpr int WHOLE_BIT = 1 << 30;

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

// narrow chains on the vector ALU (mlp_valu_kernel): one (b, m, s) row per lane
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

struct Geometry {
    int kp[MAXL], np[MAXL];
    long long off[MAXL];     // packed layer l: bias block then A fragments
    long long raw_w[MAXL];   // plain row-major copy W[l][C_out][C_in] (for the VALU kernel)
    long long raw_b[MAXL];
    long long stream_off;    // stream image of the cooperative register-resident kernel (mlp_coop.hip), -1 = none
    long long stream_frags;
    long long total;
};
Geometry geometry(int L, const int *dims, int first_has_xyz);     // mlp_pack.hip
int check_dims(const char *fn, int L, const int *dims);

// launchers of the tiled kernels (mlp_chain.hip): W waves, RW row tiles per wave, CW = 2: two output tiles per wave
int launch_tiled(const MlpParams &p, int W, int RW, int CW, size_t lds, long long nblocks, hipStream_t st);
int launch_tiled_multi(const MultiParams &mp, int W, int rwmax, bool cw2, size_t lds, hipStream_t st);
// vector-ALU kernel: shape 1 = 4 -> 16 -> 16 -> 32, 2 = 4 -> 32 -> 32 -> 64
int launch_valu(const ValuParams &v, int shape, long long nblocks, hipStream_t st);

}  // namespace chain
}  // namespace sad
