// Helpers shared by the register-resident chain kernels (mlp_reg.hip, mlp_coop.hip): MFMA operand shuffles,
// the DPP segmented max-scan and the staged pooled-output path.  Included inside each file's anonymous namespace.
// No reference source exists (/root/reference/README.md:1-2).
#pragma once

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WHOLE_BIT = 1 << 30;
using sad::RegChain;
using sad::RegMulti;

struct Swapped { float lo, hi; };
// lanes 32-63 of `a` <-> lanes 0-31 of `b`:  lo = (a.lo | b.lo), hi = (a.hi | b.hi) read as (lanes 0-31 | lanes 32-63)
__device__ __forceinline__ Swapped swap32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    // (copy the elements out first: __builtin_bit_cast applied to r[1] directly reads element 0 with this hipcc)
    const unsigned r0 = r[0], r1 = r[1];
    return {__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}

// (c0..c3 on lanes 0-31 | c4..c7 on lanes 32-63) in v[0..3]  ->  B operands of the four MFMAs of the k-group:
// out[e] = (c_{2e} | c_{2e+1})
__device__ __forceinline__ void to_operands(float v0, float v1, float v2, float v3, float *out) {
    const Swapped s01 = swap32(v0, v1);   // (c0|c1), (c4|c5)
    const Swapped s23 = swap32(v2, v3);   // (c2|c3), (c6|c7)
    out[0] = s01.lo;
    out[1] = s23.lo;
    out[2] = s01.hi;
    out[3] = s23.hi;
}

__device__ __forceinline__ f32x16 mma4(f32x16 acc, const float4 a, const float *b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x16 bias_tile(const float *sb, int h) {   // sb: 32 biases of the tile (LDS)
    f32x16 t;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 bv = *reinterpret_cast<const float4 *>(sb + 8 * a + 4 * h);
        t[4 * a + 0] = bv.x; t[4 * a + 1] = bv.y; t[4 * a + 2] = bv.z; t[4 * a + 3] = bv.w;
    }
    return t;
}

// x > 0 ? x : 0 as ONE vector-ALU operation per register: v_med3_f32(x, 0, +inf).  Written as `x > 0 ? x : 0` the compiler
// emits a canonicalising v_max_f32 v, v, v in front of the compare, 16 more operations per tile.  Until round 3 this was a
// v_max_f32 in an asm statement; hipcc pads no MFMA -> VALU wait states for instructions inside asm (the bf16 chain kernel
// read stale accumulators that way), so the operation is a builtin the hazard recogniser knows.  Same bits for every
// finite x and for -0 (both give +0).
__device__ __forceinline__ f32x16 relu16(f32x16 t) {
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = __builtin_amdgcn_fmed3f(t[g], 0.f, __builtin_inff());
    return t;
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, true);   // out-of-range / masked rows read 0
}

// Segmented inclusive max-scan over the 32 rows (lanes j = 0..31 of each half) for values >= 0:
// m[st] = all-ones where lane j - 2^st lies in the same 16-lane DPP row AND the same group, m[4] = all-ones
// on the upper 16 lanes whose group continues from lane 15 of the lower 16.
struct PoolMasks {
    bool m[5];       // this lane takes part in step k
    bool any[5];     // (wave-uniform) some lane takes part in step k: a step nobody takes part in is the identity and is skipped
};

__device__ __forceinline__ PoolMasks pool_masks(int key) {   // key >= 1 for live rows, 0 for rows past the end
    PoolMasks pm;
    pm.m[0] = dpp_i<0x111, 0xF>(key) == key;   // row_shr:1
    pm.m[1] = dpp_i<0x112, 0xF>(key) == key;   // row_shr:2
    pm.m[2] = dpp_i<0x114, 0xF>(key) == key;   // row_shr:4
    pm.m[3] = dpp_i<0x118, 0xF>(key) == key;   // row_shr:8
    pm.m[4] = dpp_i<0x142, 0xA>(key) == key;   // row_bcast:15 into DPP rows 1 and 3 (rows 0, 2 read 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) pm.any[k] = __ballot(pm.m[k]) != 0;   // groups of one or two rows need no or one step
    return pm;
}

// All 16 registers of a tile at once, step by step: consecutive instructions are independent, so the DPP
// read-after-write wait states are covered by the other registers' work (a register-by-register scan is a
// chain of ten dependent instructions per register: ~4000 cycles per tile instead of ~700).
__device__ __forceinline__ f32x16 seg_max16(f32x16 t, const PoolMasks &pm) {
#ifdef SAD_NOSCAN           // measurement build: no pooling arithmetic (wrong results)
    return t;
#endif
    unsigned x[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float f = t[g];          // (copy first: __builtin_bit_cast on a vector element reads element 0 with this hipcc)
        x[g] = __builtin_bit_cast(unsigned, f);
    }
    // (values are >= +0: their bit patterns order like unsigned integers and 0 is the neutral element, so a lane whose
    // DPP source is out of range or masked reads 0 = "no change").  Written as max-then-select: two vector-ALU
    // operations per register and step (v_max_u32_dpp + v_cndmask_b32); the and-then-max form compiled to three
#define SAD_STEP(CTRL, RM, K)                                                   \
    if (pm.any[K]) {                                                            \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                        \
            const unsigned u = (unsigned)dpp_i<CTRL, RM>((int)x[g]);            \
            const unsigned r = u > x[g] ? u : x[g];                             \
            x[g] = pm.m[K] ? r : x[g];                                          \
        }                                                                       \
    }
    SAD_STEP(0x111, 0xF, 0)
    SAD_STEP(0x112, 0xF, 1)
    SAD_STEP(0x114, 0xF, 2)
    SAD_STEP(0x118, 0xF, 3)
    SAD_STEP(0x142, 0xA, 4)
#undef SAD_STEP
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = __builtin_bit_cast(float, x[g]);
    return t;
}

__device__ __forceinline__ void atomic_max_pos(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned *>(addr), __builtin_bit_cast(unsigned, v));
}

// max-pool of one finished 32-channel tile over the rows of each group + store (whole groups) / atomic max (groups
// that continue in another tile; outputs are >= 0 and the buffer starts at zero)
__device__ __forceinline__ void pool_store(f32x16 t, const PoolMasks &pm, bool tail, bool whole, float *orow, int o, int h,
                                           const RegChain &c) {
#ifdef SAD_REG_NOPOOL       // measurement build: no pooling, one store per tile keeps the chain alive
    if (t[0] == 123.f) orow[0] = t[0];
    return;
#endif
    t = seg_max16(t, pm);
#ifdef SAD_REG_NOSTORE      // measurement build: pooling arithmetic only
    {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += t[g];
        if (s == 123.456f) orow[0] = s;
        return;
    }
#endif
#ifdef SAD_REG_NOATOMIC     // measurement build: plain stores where the product merges with an atomic max (wrong results)
    whole = true;
#endif
    if (tail) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ch = o * 32 + 8 * a + 4 * h;
            if (whole && c.vec_out && ch + 3 < c.cout_last) {
                *reinterpret_cast<float4 *>(orow + ch) = make_float4(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ch + e >= c.cout_last) continue;
                    if (whole) orow[ch + e] = t[4 * a + e];
                    else atomic_max_pos(orow + ch + e, t[4 * a + e]);
                }
            }
        }
    }
}

// Staged output.  A tile's pooled results are first collected in the wave's LDS (slot = ordinal of the group
// inside the tile, COUT floats per slot) and written out when the tile is complete: one coalesced store per
// whole group, or COUT / 64 atomic-max instructions of 256 contiguous bytes for a group that continues in
// another tile — instead of 4 stores + 16 scattered single-lane atomics per 32-channel tile (measured on
// sa3.b2: the scattered atomics alone were 10 % of the kernel; a wave stalls once ~16 of them are in flight).
struct Stage {
    float *lds;          // [slots][COUT] of this wave, or nullptr: direct stores (more groups in the tile than slots)
    int slot;            // this lane's group ordinal inside the tile
    unsigned tails;      // bit j: lane j holds the last row of its group in this tile
    int ngroups;
};

template <int COUT>
__device__ __forceinline__ void pool_stage(f32x16 t, const PoolMasks &pm, bool tail, const Stage &sg, int o, int h) {
#ifdef SAD_REG_NOPOOL       // measurement build (as in pool_store): no pooling, no staging; the comparison keeps the chain alive
    if (t[0] == 123.f) sg.lds[0] = t[0];
    return;
#endif
    t = seg_max16(t, pm);
#ifdef SAD_REG_NOSTORE      // measurement build (as in pool_store): the pooling arithmetic only, nothing staged or stored
    {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += t[g];
        if (s == 123.456f) sg.lds[0] = s;
        return;
    }
#endif
    if (tail) {
        float *d = sg.lds + sg.slot * COUT + o * 32 + 4 * h;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4 *>(d + 8 * a) = make_float4(t[4 * a], t[4 * a + 1], t[4 * a + 2], t[4 * a + 3]);
    }
}

template <int COUT>
__device__ __forceinline__ void stage_flush(const Stage &sg, int grp, bool whole, int lane, const RegChain &c) {
#if defined(SAD_REG_NOPOOL) || defined(SAD_REG_NOSTORE)
    return;
#endif
    unsigned rem = sg.tails;
    for (int s = 0; s < sg.ngroups; ++s) {                 // wave-uniform loop over the groups that end in this tile
        const int p = __builtin_ctz(rem);
        rem &= rem - 1;
        const int g = __builtin_amdgcn_readlane(grp, p);
        const bool w = __builtin_amdgcn_readlane((int)whole, p) != 0;
        float *orow = c.out + (long long)g * c.ld_out + c.col_off;
        const float *src = sg.lds + s * COUT;
        if (w && c.vec_out && COUT == c.cout_last) {
            if (lane * 4 < COUT) *reinterpret_cast<float4 *>(orow + lane * 4) = *reinterpret_cast<const float4 *>(src + lane * 4);
        } else {
#pragma unroll
            for (int k = 0; k < (COUT + 63) / 64; ++k) {
                const int ch = lane + 64 * k;
                if (ch < COUT && ch < c.cout_last) {
                    if (w) orow[ch] = src[ch];
                    else atomic_max_pos(orow + ch, src[ch]);
                }
            }
        }
    }
}

