// One plain-row f32 layer  out[row, :] = act(W . x[row, :] + b)  on v_mfma_f32_32x32x2_f32: "geometry 5" of sad_mlp_chain_f32
// (SPEC.md §6; the stage aggregations of the detector: 128 -> 64 on 131 072 rows, 384 -> 128 on 32 768, 768 -> 256 on 16 384
// per 32-scene step).  No reference source exists (/root/reference/README.md:1-2 is the whole upstream repository).
//
// These layers read 50 - 67 MB of pooled features for 2 - 6 GFLOP: the two narrow ones are memory-bound, and the tiled kernel
// (activations transposed into an LDS [channel][row] image, barriers per layer) and the layer-streamed kernel (128 x 128 items,
// both operands through LDS) spend their time moving rows, not multiplying.  Here a wave owns 32 rows and ALL output channels of
// its item and its x operand never touches LDS: lane (j, h) loads 16 bytes of its row per k-group (c0..c3 on the lower lane half,
// c4..c7 on the upper) and two v_permlane32_swap turn them into the four B operands of the k-group (reg_common.h); the weight
// fragments of a chunk of four k-groups are shared by the four waves of the workgroup through two LDS stages (each wave fetches a
// quarter, one barrier per chunk); the next chunk's rows are in flight during the MFMAs.  The accumulator starts at the bias and k
// ascends: every output is SPEC §6's fmaf chain bit for bit, like every other kernel behind this entry point.
#include "common.h"

namespace {

#include "reg_common.h"

using sad::RowsJob;
constexpr int KC = 4;                 // k-groups (of 8) per chunk

template <int NT>                     // channel tiles (of 32) per item: 2 (C_out <= 64) or 4
__global__ __launch_bounds__(256, NT == 2 ? 4 : 2) void mlp_rows_kernel(const RowsJob jb) {   // (memory-bound: as many waves as the accumulators allow)
    constexpr int STAGE_F4 = KC * NT * 64;
    __shared__ __attribute__((aligned(16))) float4 lds[2 * STAGE_F4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int KG = jb.kg;
    const int NC = (KG + KC - 1) / KC;
    const int ncb = jb.ncb, nrb = jb.nrb;
    // XCD-aware item order (workgroup g runs on XCD g % 8): the channel blocks of a row block follow each other on one XCD
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    const int rb = xcd + 8 * (slot / ncb), cb = slot % ncb;
    if (rb >= nrb) return;
    const int nt = jb.ct - cb * NT < NT ? jb.ct - cb * NT : NT;        // channel tiles of this item
    const long long row = (long long)rb * 128 + wave * 32 + j;
    const bool live = row < jb.rows;
    const long long rowc = live ? row : jb.rows - 1;
    const float4 *wimg = reinterpret_cast<const float4 *>(jb.packed + jb.off + jb.np) + (size_t)(cb * NT) * KG * 64;   // [tile][k-group][lane]
    const unsigned ulane = (unsigned)lane;
    const float *xrow = jb.x + (size_t)rowc * jb.ldx + 4 * h;

    struct XRaw { float4 v[KC]; };
    auto load_x = [&](int c) -> XRaw {
        XRaw r;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            int kg = c * KC + s;
            kg = kg < KG ? kg : KG - 1;
            r.v[s] = *reinterpret_cast<const float4 *>(xrow + 8 * kg);
        }
        return r;
    };
    struct WRaw { float4 f[NT]; };
    auto load_w = [&](int c) -> WRaw {      // fragment f = wave * NT + i of the chunk -> (k-group f / NT, tile f % NT)
        WRaw v;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = wave * NT + i;
            int kg = c * KC + (f / NT);
            kg = kg < KG ? kg : KG - 1;
            const int t = (f % NT) < nt ? (f % NT) : 0;
            v.f[i] = (wimg + ((size_t)t * KG + kg) * 64)[ulane];
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
        const float *bias = jb.packed + jb.off + (cb * NT + (t < nt ? t : 0)) * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + 8 * a + 4 * h);
            acc[t][4 * a] = bv.x; acc[t][4 * a + 1] = bv.y; acc[t][4 * a + 2] = bv.z; acc[t][4 * a + 3] = bv.w;
        }
    }
    XRaw xn = load_x(0);
    {
        const WRaw w0 = load_w(0);
        store_w(w0, lds);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        const float4 *cur = lds + (c & 1) * STAGE_F4;
        const XRaw xc = xn;
        const int cn = c + 1 < NC ? c + 1 : c;
        xn = load_x(cn);                            // the next chunk's rows and weights are in flight during the MFMAs
        const WRaw wn = load_w(cn);
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (c * KC + s < KG) {                  // (wave-uniform: the last chunk may be partial; its k-groups are never padded)
                float ops[4];
                to_operands(xc.v[s].x, xc.v[s].y, xc.v[s].z, xc.v[s].w, ops);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t < nt) acc[t] = mma4(acc[t], cur[(s * NT + t) * 64 + lane], ops);
            }
        }
        if (c + 1 < NC) store_w(wn, lds + ((c + 1) & 1) * STAGE_F4);
        __syncthreads();
    }
    // ---- epilogue: lane = row, registers 4a .. 4a+3 = channels 32 t + 8 a + 4 h .. + 3 ----
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= nt) continue;
        const f32x16 v = jb.relu ? relu16(acc[t]) : acc[t];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int co = (cb * NT + t) * 32 + 8 * a + 4 * h;
            float *o = jb.out + (size_t)row * jb.ld_out + jb.col_off + co;
            if (co + 3 < jb.cout && jb.vec_out) {
                *reinterpret_cast<float4 *>(o) = make_float4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < jb.cout) o[e] = v[4 * a + e];
            }
        }
    }
}


// ---- second form (round 5): the same layer as a tiled GEMM fed by LDS-DMA (the f32 twin of bf16_rows2_kernel, csrc/mlp_bf16_rows.hip) ----
// The first form's rows come fragment-shaped (every load instruction touches 32 lines for 16 bytes each), its weights through registers
// and a write -> barrier -> read turn-around of the LDS per chunk; at 0.43 - 0.54 of the f32 peak on layers whose bytes would allow more.
// Here both operands go global -> LDS with global_load_lds_dwordx4 into a ring of three stages, two chunks ahead: a stage holds the chunk's
// weight fragments (four k-groups x 2 NTW channel tiles, lane-linear as packed) and the 128 x 32 row tile as 128-byte rows whose 16-byte
// columns are XOR-ed with the row's low three bits (on the SOURCE side: a DMA's LDS image is lane-linear); 8 waves = 4 row tiles x 2
// channel groups; lane (j, h) reads c0..c3 / c4..c7 of its row per k-group and two v_permlane32_swap make the four B operands, exactly as
// in the first form.  The DMA goes through inline asm (the compiler keeps no count of it), is retired by a counted vmcnt one chunk later and
// published by the chunk's barrier.  Same MFMAs in the same k order from the same bias: SPEC section 6's fmaf chains bit for bit.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;       // (M0 is the compiler's: saved and restored in the statement that uses it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N)); }

template <int NTW>
__global__ __launch_bounds__(512) void mlp_rows2_kernel(const RowsJob jb) {
    constexpr int TW = 2 * NTW;                       // channel tiles per workgroup
    constexpr int WB = KC * TW * 1024;                // bytes of a stage: weight fragments
    constexpr int XB = 128 * 128;                     //                   row tile (128 rows x 32 floats)
    constexpr int SB = WB + XB;
    constexpr int PW = KC * TW / 8;                   // weight pieces (1 KB) per wave and chunk
    constexpr int PV = PW + 2;                        // DMA instructions per wave and chunk
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 3, cw = wave >> 2;
    const int j = lane & 31, h = lane >> 5;
    const int KG = jb.kg, NC = KG / KC;               // (host: KG % KC == 0)
    const int ncb = jb.ncb, nrb = jb.nrb;
    const int g = blockIdx.x;
    const int xcd = g & 7, slot = g >> 3;
    const int rb = xcd + 8 * (slot / ncb), cb = slot % ncb;
    if (rb >= nrb) return;                            // (workgroup-uniform)
    const unsigned lds0 = (unsigned)(size_t)smem2;
    const char *wsrc[PW];
    unsigned wdst[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = wave + 8 * i, kgl = p / TW, t = p % TW;
        int tile = cb * TW + t;
        tile = tile < jb.ct ? tile : jb.ct - 1;        // (a ragged last channel block: its surplus tiles repeat the last one and store nothing)
        wsrc[i] = reinterpret_cast<const char *>(jb.packed + jb.off + jb.np) + (((size_t)tile * KG + kgl) * 64 + lane) * 16;
        wdst[i] = p * 1024;
    }
    const char *xsrc[2];
    unsigned xdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave + 8 * i;                    // piece q: rows 8 q .. 8 q + 7 of the tile, one 128-byte line each
        long long row = (long long)rb * 128 + 8 * q + (lane >> 3);
        row = row < jb.rows ? row : jb.rows - 1;
        xsrc[i] = reinterpret_cast<const char *>(jb.x) + (size_t)row * jb.ldx * 4 + (((lane & 7) ^ (lane >> 3)) * 16);
        xdst[i] = WB + q * 1024;
    }
    auto issue = [&](int c) __attribute__((always_inline)) {      // the DMA of chunk c into stage c % 3
        const unsigned st = lds0 + (unsigned)(c % 3) * SB;
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(wsrc[i] + (size_t)c * (KC * 1024), st + wdst[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(xsrc[i] + (size_t)c * 128, st + xdst[i]);
    };
    issue(0);
    if (NC > 1) issue(1);
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        int tile = cb * TW + cw * NTW + t;
        tile = tile < jb.ct ? tile : jb.ct - 1;
        const float *bias = jb.packed + jb.off + tile * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + 8 * a + 4 * h);
            acc[t][4 * a] = bv.x; acc[t][4 * a + 1] = bv.y; acc[t][4 * a + 2] = bv.z; acc[t][4 * a + 3] = bv.w;
        }
    }
    // the bias reads are consumed HERE: left to their first real use (the first MFMA of the loop), the compiler's wait for them — a
    // vmcnt(0), which also drains the DMA two chunks ahead — would run in every chunk
#pragma unroll
    for (int t = 0; t < NTW; ++t) asm volatile("" : "+v"(acc[t]));
    if (NC > 1) wait_vm<PV>(); else wait_vm<0>();
    __syncthreads();                                   // (lgkmcnt(0) + s_barrier: the compiler knows of no vector-memory operation in flight)
    const int xrow = rw * 32 + j;
    const unsigned xoff = WB + xrow * 128, xsw = (unsigned)(xrow & 7);
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        if (c + 2 < NC) issue(c + 2);
        const unsigned char *st = smem2 + (c % 3) * SB;
        float4 xv[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) xv[s] = *reinterpret_cast<const float4 *>(st + xoff + (((unsigned)(2 * s + h) ^ xsw) * 16));
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            float ops[4];
            to_operands(xv[s].x, xv[s].y, xv[s].z, xv[s].w, ops);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                acc[t] = mma4(acc[t], *reinterpret_cast<const float4 *>(st + ((s * TW + cw * NTW + t) * 64 + lane) * 16), ops);
        }
        if (c + 1 < NC) {
            if (c + 2 < NC) wait_vm<PV>(); else wait_vm<0>();      // retire chunk c + 1's DMA, leave chunk c + 2's in flight
            __syncthreads();
        }
    }
    const long long row = (long long)rb * 128 + rw * 32 + j;
    if (row >= jb.rows) return;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tile = cb * TW + cw * NTW + t;
        if (tile >= jb.ct) continue;
        const f32x16 v = jb.relu ? relu16(acc[t]) : acc[t];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int co = tile * 32 + 8 * a + 4 * h;
            float *o = jb.out + (size_t)row * jb.ld_out + jb.col_off + co;
            if (co + 3 < jb.cout && jb.vec_out) {
                *reinterpret_cast<float4 *>(o) = make_float4(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < jb.cout) o[e] = v[4 * a + e];
            }
        }
    }
}


}  // namespace

namespace sad {

int launch_rows(const RowsJob &job, hipStream_t st) {
    RowsJob jb = job;
    jb.nrb = (int)((jb.rows + 127) / 128);
    // second form (tiled GEMM fed by LDS-DMA): K a whole number of 32-deep chunks, 16-byte aligned row starts; mlp_rows_form = 1: first form
    if (get_option(OPT_MLP_ROWS_FORM) != 1 && jb.kg % KC == 0 && (jb.ldx * 4) % 16 == 0 && (reinterpret_cast<uintptr_t>(jb.x) & 15) == 0) {
        const int ntw = jb.ct > 2 ? 2 : 1;
        jb.ncb = (jb.ct + 2 * ntw - 1) / (2 * ntw);
        const long long grid2 = 8LL * ((jb.nrb + 7) / 8) * jb.ncb;
        if (grid2 >= (1LL << 31)) return fail(SAD_EINVAL, "sad_mlp_chain_f32: too many rows");
        const size_t lds = 3 * (size_t)(KC * 2 * ntw * 1024 + 128 * 128);
        if (ntw == 2) {
            static std::atomic<uint64_t> done2{0};
            lds_attr_once(done2, reinterpret_cast<const void *>(&mlp_rows2_kernel<2>), 160 * 1024);
            hipLaunchKernelGGL(mlp_rows2_kernel<2>, dim3((unsigned)grid2), dim3(512), lds, st, jb);
        } else {
            static std::atomic<uint64_t> done1{0};
            lds_attr_once(done1, reinterpret_cast<const void *>(&mlp_rows2_kernel<1>), 160 * 1024);
            hipLaunchKernelGGL(mlp_rows2_kernel<1>, dim3((unsigned)grid2), dim3(512), lds, st, jb);
        }
        return check_launch("sad_mlp_chain_f32 (row-streaming layer, second form)");
    }
    // two channel tiles per item: four were slower on every aggregation of the detector (39 / 68 / 123 us against 42 / 73 / 135 on
    // 384 -> 128, 768 -> 256, 1536 -> 512: the rows are read once more, from L2, but twice the workgroups hide the row loads'
    // latency); mlp_rw = 4 is the A/B knob
    const int nt = get_option(OPT_MLP_RW) == 4 && jb.ct > 2 ? 4 : 2;
    jb.ncb = (jb.ct + nt - 1) / nt;
    const long long grid = 8LL * ((jb.nrb + 7) / 8) * jb.ncb;
    if (grid >= (1LL << 31)) return fail(SAD_EINVAL, "sad_mlp_chain_f32: too many rows");
    if (nt == 4) hipLaunchKernelGGL(mlp_rows_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, jb);
    else hipLaunchKernelGGL(mlp_rows_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, jb);
    return check_launch("sad_mlp_chain_f32 (row-streaming layer)");
}

}  // namespace sad
