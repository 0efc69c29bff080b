/*
 * sad_oracle.c — CPU spec-oracle for the set-abstraction + size-adaptive-clustering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (3dsad-main_amd/) never does.
 *
 * PARITY UNPINNED: the upstream reference is a two-line README (/root/reference/README.md:1-2) with
 * no implementation, tests or golden vectors, so there is no reference file:line to follow.  Every
 * function below restates a section of this repository's own SPEC.md (cited per function), which
 * freezes the semantics BASELINE.json's north_star names.
 *
 * Build: gcc -O3 -fopenmp -ffp-contract=off -mavx2 -mfma (see oracle/Makefile).  -ffp-contract=off
 * keeps every a*b+c that is not written as fmaf() un-fused, exactly as the HIP kernels are built.
 * Results do not depend on the thread count: every output element is computed by one thread.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* SPEC.md §1 */
static inline float d2f(const float *p, const float *c) {
    float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

ORC_API int orc_version(void) { return 1; }

/* SPEC.md §2 — farthest point sampling */
ORC_API void orc_fps(const float *xyz, int B, int N, int M, int32_t *idx) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        int32_t *o = idx + (size_t)b * M;
        float *mind = (float *)malloc(sizeof(float) * (size_t)N);
        for (int j = 0; j < N; ++j) mind[j] = INFINITY;
        int last = 0;
        o[0] = 0;
        for (int i = 1; i < M; ++i) {
            const float *c = p + (size_t)last * 3;
            float best = -1.0f;
            int besti = 0;
            for (int j = 0; j < N; ++j) {
                float d = d2f(p + (size_t)j * 3, c);
                float m = mind[j] < d ? mind[j] : d;
                mind[j] = m;
                if (m > best) { best = m; besti = j; } /* strict: ties keep the lowest j */
            }
            last = besti;
            o[i] = last;
        }
        free(mind);
    }
}

/* SPEC.md §15 — feature-distance FPS; feat point-major [B,N,C] */
static inline float dff(const float *p, const float *q, const float *fp, const float *fq, int C, float w) {
    float d = d2f(p, q) * w;
    for (int c = 0; c < C; ++c) {
        float t = fp[c] - fq[c];
        float tt = t * t;
        d = d + tt;
    }
    return d;
}
ORC_API void orc_ffps(const float *xyz, const float *feat, int B, int N, int C, int M, float w_xyz, int32_t *idx) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        const float *f = feat + (size_t)b * N * C;
        int32_t *o = idx + (size_t)b * M;
        float *mind = (float *)malloc(sizeof(float) * (size_t)N);
        for (int j = 0; j < N; ++j) mind[j] = INFINITY;
        int last = 0;
        o[0] = 0;
        for (int i = 1; i < M; ++i) {
            float best = -1.0f;
            int besti = 0;
            for (int j = 0; j < N; ++j) {
                float d = dff(p + (size_t)j * 3, p + (size_t)last * 3, f + (size_t)j * C, f + (size_t)last * C, C, w_xyz);
                float m = mind[j] < d ? mind[j] : d;
                mind[j] = m;
                if (m > best) { best = m; besti = j; }
            }
            last = besti;
            o[i] = last;
        }
        free(mind);
    }
}

/* SPEC.md §3 — ball query; radius_pc == NULL → scalar radius, else per-centroid radius[B,M] */
ORC_API void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int M, int S,
                            float radius, const float *radius_pc, int32_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            const float *p = xyz + (size_t)b * N * 3;
            const float *c = new_xyz + ((size_t)b * M + m) * 3;
            int32_t *o = idx + ((size_t)b * M + m) * S;
            float r = radius_pc ? radius_pc[(size_t)b * M + m] : radius;
            float r2 = r * r;
            int cnt = 0;
            for (int j = 0; j < N && cnt < S; ++j) {
                if (d2f(p + (size_t)j * 3, c) < r2) o[cnt++] = j;
            }
            int fill = cnt ? o[0] : 0;
            for (int s = cnt; s < S; ++s) o[s] = fill;
        }
}

/* SPEC.md §4 — k nearest neighbours, sorted by (d2, index) */
ORC_API void orc_knn(const float *xyz, const float *new_xyz, int B, int N, int M, int K,
                     int32_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            const float *p = xyz + (size_t)b * N * 3;
            const float *c = new_xyz + ((size_t)b * M + m) * 3;
            int32_t *o = idx + ((size_t)b * M + m) * K;
            float bd[64];
            int32_t bi[64];
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                float d = d2f(p + (size_t)j * 3, c);
                if (cnt == K && !(d < bd[K - 1])) continue; /* equal d, larger j: stays out */
                int pos = cnt < K ? cnt : K - 1;
                while (pos > 0 && d < bd[pos - 1]) { /* strict: equal d keeps the earlier j first */
                    bd[pos] = bd[pos - 1];
                    bi[pos] = bi[pos - 1];
                    --pos;
                }
                bd[pos] = d;
                bi[pos] = j;
                if (cnt < K) ++cnt;
            }
            for (int s = 0; s < K; ++s) o[s] = bi[s];
        }
}

/* SPEC.md §5 — pure copies; esz = element size in bytes (2 or 4) */
ORC_API void orc_gather_points(const void *src, const int32_t *idx, int B, int C, int N, int M,
                               int esz, void *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const char *s = (const char *)src + ((size_t)b * C + c) * N * esz;
            char *o = (char *)out + ((size_t)b * C + c) * M * esz;
            const int32_t *ix = idx + (size_t)b * M;
            for (int m = 0; m < M; ++m) memcpy(o + (size_t)m * esz, s + (size_t)ix[m] * esz, esz);
        }
}

ORC_API void orc_gather_xyz(const float *xyz, const int32_t *idx, int B, int N, int M, float *out) {
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            const float *s = xyz + ((size_t)b * N + idx[(size_t)b * M + m]) * 3;
            float *o = out + ((size_t)b * M + m) * 3;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        }
}

ORC_API void orc_group_points(const void *feat, const int32_t *idx, int B, int C, int N, int M,
                              int S, int esz, void *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const char *s = (const char *)feat + ((size_t)b * C + c) * N * esz;
            char *o = (char *)out + ((size_t)b * C + c) * M * S * esz;
            const int32_t *ix = idx + (size_t)b * M * S;
            for (size_t t = 0; t < (size_t)M * S; ++t)
                memcpy(o + t * esz, s + (size_t)ix[t] * esz, esz);
        }
}

/* ------------------------------------------------------------------------------------------
 * SPEC.md §6 — one 1x1-conv layer on a block of rows: y[r][o] = act(fmaf-chain over k ascending).
 * x: rows x Cin (row stride ldx), wT: Cin x Cout (transposed copy of W[Cout][Cin]), y: rows x Cout
 * (row stride ldy).  The o-loop is the vector loop (contiguous wT row), the k-loop is the chain.
 * ---------------------------------------------------------------------------------------- */
static void layer_rows(const float *x, int ldx, int rows, const float *wT, const float *bias,
                       int Cin, int Cout, int relu, float *y, int ldy) {
    /* Register-blocked so the CPU baseline is an honest one: a 6-row x 16-channel tile of chains
     * lives in 12 ymm accumulators across the whole k loop; _mm256_fmadd_ps is fmaf per element
     * (one rounding), so the result is the same k-ascending chain as the scalar tail below. */
    int r = 0;
    for (; r + 6 <= rows; r += 6) {
        const float *xr = x + (size_t)r * ldx;
        int o = 0;
        for (; o + 16 <= Cout; o += 16) {
            __m256 acc[6][2];
            const __m256 b0 = _mm256_loadu_ps(bias + o), b1 = _mm256_loadu_ps(bias + o + 8);
            for (int i = 0; i < 6; ++i) { acc[i][0] = b0; acc[i][1] = b1; }
            for (int k = 0; k < Cin; ++k) {
                const __m256 w0 = _mm256_loadu_ps(wT + (size_t)k * Cout + o);
                const __m256 w1 = _mm256_loadu_ps(wT + (size_t)k * Cout + o + 8);
                for (int i = 0; i < 6; ++i) {
                    const __m256 a = _mm256_broadcast_ss(xr + (size_t)i * ldx + k);
                    acc[i][0] = _mm256_fmadd_ps(w0, a, acc[i][0]);
                    acc[i][1] = _mm256_fmadd_ps(w1, a, acc[i][1]);
                }
            }
            for (int i = 0; i < 6; ++i) {
                _mm256_storeu_ps(y + (size_t)(r + i) * ldy + o, acc[i][0]);
                _mm256_storeu_ps(y + (size_t)(r + i) * ldy + o + 8, acc[i][1]);
            }
        }
        for (; o < Cout; ++o)
            for (int i = 0; i < 6; ++i) {
                float acc = bias[o];
                for (int k = 0; k < Cin; ++k) acc = fmaf(wT[(size_t)k * Cout + o], xr[(size_t)i * ldx + k], acc);
                y[(size_t)(r + i) * ldy + o] = acc;
            }
    }
    for (; r < rows; ++r) {
        float *y0 = y + (size_t)r * ldy;
        for (int o = 0; o < Cout; ++o) y0[o] = bias[o];
        for (int k = 0; k < Cin; ++k) {
            const float *w = wT + (size_t)k * Cout;
            float a0 = x[(size_t)r * ldx + k];
#pragma omp simd
            for (int o = 0; o < Cout; ++o) y0[o] = fmaf(w[o], a0, y0[o]);
        }
    }
    if (relu)
        for (int rr = 0; rr < rows; ++rr) {
            float *yy = y + (size_t)rr * ldy;
            for (int o = 0; o < Cout; ++o) yy[o] = yy[o] > 0.0f ? yy[o] : 0.0f;
        }
}

static float *transpose_w(const float *W, int Cout, int Cin) {
    float *t = (float *)malloc(sizeof(float) * (size_t)Cin * Cout);
    for (int o = 0; o < Cout; ++o)
        for (int k = 0; k < Cin; ++k) t[(size_t)k * Cout + o] = W[(size_t)o * Cin + k];
    return t;
}

/* SPEC.md §6 — plain MLP chain on rows (aggregation layers, candidate MLP, head).
 * x: R x dims[0] (point-major rows); W[l]: dims[l+1] x dims[l]; relu_mask bit l = ReLU after layer l.
 * out: R x dims[L] written with row stride ld_out at column offset col_off. */
ORC_API void orc_mlp_rows(const float *x, int64_t R, int L, const int *dims, const float *const *W,
                          const float *const *bias, int relu_mask, float *out, int ld_out,
                          int col_off) {
    float *wT[8];
    int maxc = 0;
    for (int l = 0; l < L; ++l) {
        wT[l] = transpose_w(W[l], dims[l + 1], dims[l]);
        if (dims[l + 1] > maxc) maxc = dims[l + 1];
    }
    const int RB = 96;
#pragma omp parallel
    {
        float *buf0 = (float *)malloc(sizeof(float) * (size_t)RB * maxc);
        float *buf1 = (float *)malloc(sizeof(float) * (size_t)RB * maxc);
#pragma omp for schedule(static)
        for (int64_t r0 = 0; r0 < R; r0 += RB) {
            int rows = (int)((R - r0) < RB ? (R - r0) : RB);
            const float *cur = x + (size_t)r0 * dims[0];
            int ldc = dims[0];
            for (int l = 0; l < L; ++l) {
                float *dst = (l & 1) ? buf1 : buf0;
                layer_rows(cur, ldc, rows, wT[l], bias[l], dims[l], dims[l + 1],
                           (relu_mask >> l) & 1, dst, dims[l + 1]);
                cur = dst;
                ldc = dims[l + 1];
            }
            for (int r = 0; r < rows; ++r)
                memcpy(out + (size_t)(r0 + r) * ld_out + col_off, cur + (size_t)r * ldc,
                       sizeof(float) * (size_t)dims[L]);
        }
        free(buf0);
        free(buf1);
    }
    for (int l = 0; l < L; ++l) free(wT[l]);
}

/* SPEC.md §6 — fused group -> MLP -> max-pool, never materialising the grouped tensor.
 * xyz[B,N,3]; feat_pm[B,N,C] point-major (NULL when C == 0); new_xyz[B,M,3]; idx[B,M,S];
 * dims[0] must equal 3 + C; all L layers carry ReLU.
 * out[(b*M+m)*ld_out + col_off + o] = max_s y_L[o] (point-major rows, so branches concatenate by
 * writing at different col_off into one [B,M,ld_out] buffer). */
static void sa_group_mlp_max_impl(const float *xyz, const float *feat_pm, const float *new_xyz,
                                  const int32_t *idx, int B, int N, int M, int S_all, int C, int L,
                                  const int *dims, const float *const *W, const float *const *bias,
                                  float *out, int ld_out, int col_off, int skip_padding) {
    float *wT[8];
    int maxc = dims[0];
    for (int l = 0; l < L; ++l) {
        wT[l] = transpose_w(W[l], dims[l + 1], dims[l]);
        if (dims[l + 1] > maxc) maxc = dims[l + 1];
    }
    const int Cin = dims[0], Cout = dims[L];
#pragma omp parallel
    {
        float *buf0 = (float *)malloc(sizeof(float) * (size_t)S_all * maxc);
        float *buf1 = (float *)malloc(sizeof(float) * (size_t)S_all * maxc);
#pragma omp for collapse(2) schedule(dynamic, 16)
        for (int b = 0; b < B; ++b)
            for (int m = 0; m < M; ++m) {
                const float *c = new_xyz + ((size_t)b * M + m) * 3;
                const int32_t *ix = idx + ((size_t)b * M + m) * S_all;
                /* skip_padding: only the leading rows up to the last sample that differs from the
                 * first are computed.  Every dropped row repeats sample 0 (SPEC.md §3 padding), and a
                 * duplicate row cannot change a max — the same exact rule the HIP kernel applies. */
                int S = S_all;
                if (skip_padding) {
                    S = 1;
                    for (int s = 1; s < S_all; ++s)
                        if (ix[s] != ix[0]) S = s + 1;
                }
                for (int s = 0; s < S; ++s) {
                    const float *p = xyz + ((size_t)b * N + ix[s]) * 3;
                    float *g = buf0 + (size_t)s * Cin;
                    g[0] = p[0] - c[0]; g[1] = p[1] - c[1]; g[2] = p[2] - c[2];
                    if (C) memcpy(g + 3, feat_pm + ((size_t)b * N + ix[s]) * C, sizeof(float) * (size_t)C);
                }
                float *cur = buf0;
                int ldc = Cin;
                for (int l = 0; l < L; ++l) {
                    float *dst = (cur == buf0) ? buf1 : buf0;
                    layer_rows(cur, ldc, S, wT[l], bias[l], dims[l], dims[l + 1], 1, dst, dims[l + 1]);
                    cur = dst;
                    ldc = dims[l + 1];
                }
                float *o = out + ((size_t)b * M + m) * ld_out + col_off;
                for (int ch = 0; ch < Cout; ++ch) {
                    float mx = cur[ch];
                    for (int s = 1; s < S; ++s) {
                        float v = cur[(size_t)s * ldc + ch];
                        mx = v > mx ? v : mx;
                    }
                    o[ch] = mx;
                }
            }
        free(buf0);
        free(buf1);
    }
    for (int l = 0; l < L; ++l) free(wT[l]);
}

ORC_API void orc_sa_group_mlp_max(const float *xyz, const float *feat_pm, const float *new_xyz,
                                  const int32_t *idx, int B, int N, int M, int S, int C, int L,
                                  const int *dims, const float *const *W, const float *const *bias,
                                  float *out, int ld_out, int col_off) {
    sa_group_mlp_max_impl(xyz, feat_pm, new_xyz, idx, B, N, M, S, C, L, dims, W, bias, out, ld_out, col_off, 0);
}

/* The same result with the ball-query padding rows not computed (bench.py's like-for-like CPU leg:
 * the HIP kernel skips exactly these rows).  Bit-identical to orc_sa_group_mlp_max. */
ORC_API void orc_sa_group_mlp_max_skip(const float *xyz, const float *feat_pm, const float *new_xyz,
                                       const int32_t *idx, int B, int N, int M, int S, int C, int L,
                                       const int *dims, const float *const *W, const float *const *bias,
                                       float *out, int ld_out, int col_off) {
    sa_group_mlp_max_impl(xyz, feat_pm, new_xyz, idx, B, N, M, S, C, L, dims, W, bias, out, ld_out, col_off, 1);
}

/* SPEC.md §8 steps 2-4 — candidate centres and per-candidate adaptive radius.
 * xyz3[B,M3,3], c[B,K,6] (candidate MLP output), outputs cand[B,K,3], radius[B,K]. */
ORC_API void orc_candidates(const float *xyz3, const float *c, int B, int M3, int K,
                            float shift_max, float r_min, float r_max, const float *anchor,
                            float *cand, float *radius) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < K; ++i) {
            const float *ci = c + ((size_t)b * K + i) * 6;
            const float *p = xyz3 + ((size_t)b * M3 + i) * 3;
            float *o = cand + ((size_t)b * K + i) * 3;
            float sz[3];
            for (int d = 0; d < 3; ++d) {
                float sh = ci[d];
                sh = sh < -shift_max ? -shift_max : sh;
                sh = sh > shift_max ? shift_max : sh;
                o[d] = p[d] + sh;
                float s = ci[3 + d];
                s = s < -1.0f ? -1.0f : s;
                s = s > 1.0f ? 1.0f : s;
                float t1 = s * s;
                float t2 = 0.5f * t1;
                float t3 = 1.0f + s;
                float q = t3 + t2;
                sz[d] = anchor[d] * q;
            }
            float ll = sz[0] * sz[0], ww = sz[1] * sz[1], hh = sz[2] * sz[2];
            float sum = ll + ww;
            sum = sum + hh;
            float r = 0.5f * sqrtf(sum);
            r = r < r_min ? r_min : r;
            r = r > r_max ? r_max : r;
            radius[(size_t)b * K + i] = r;
        }
}

/* SPEC.md §9 — decode head output o[B,K,10] into boxes[B,K,9]. anchors: 3 x 3. */
ORC_API void orc_decode_boxes(const float *cand, const float *o, int B, int K, const float *anchors,
                              float *boxes) {
    for (size_t t = 0; t < (size_t)B * K; ++t) {
        const float *oi = o + t * 10;
        const float *p = cand + t * 3;
        float *bx = boxes + t * 9;
        int label = 0;
        float best = oi[0];
        for (int k = 1; k < 3; ++k)
            if (oi[k] > best) { best = oi[k]; label = k; }
        float score = 1.0f / (1.0f + expf(-best));
        for (int d = 0; d < 3; ++d) {
            bx[d] = p[d] + oi[3 + d];
            float e = oi[6 + d];
            e = e < -2.0f ? -2.0f : e;
            e = e > 2.0f ? 2.0f : e;
            bx[3 + d] = anchors[label * 3 + d] * expf(e);
        }
        bx[6] = oi[9];
        bx[7] = score;
        bx[8] = (float)label;
    }
}

/* ------------------------------------------------------------------------------------------
 * SPEC.md §13 — rotated-box NMS in bird's-eye view.
 * ---------------------------------------------------------------------------------------- */
static void sincos_r(float th, float *s_out, float *c_out) {
    float n = rintf(th * 0.63661975f);
    float r = th - n * 1.5703125f;
    r = r - n * 4.8375129699707031e-4f;
    r = r - n * 7.5497899548918861e-8f;
    int q = ((int)n) & 3;
    float r2 = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * r2; ps = ps + 8.3321608736e-3f;
    ps = ps * r2; ps = ps + -1.6666654611e-1f;
    float S = r * r2; S = S * ps; S = r + S;
    float pc = 2.443315711809948e-5f;
    pc = pc * r2; pc = pc + -1.388731625493765e-3f;
    pc = pc * r2; pc = pc + 4.166664568298827e-2f;
    float C = r2 * r2; C = C * pc;
    float h = 0.5f * r2;
    float one = 1.0f - h;
    C = one + C;
    switch (q) {
        case 0: *s_out = S; *c_out = C; break;
        case 1: *s_out = C; *c_out = -S; break;
        case 2: *s_out = -S; *c_out = -C; break;
        default: *s_out = -C; *c_out = S; break;
    }
}

static void box_corners(const float *bx, float *cx, float *cy) {
    float s, c;
    sincos_r(bx[6], &s, &c);
    const float hl = 0.5f * bx[3], hw = 0.5f * bx[4];
    const float dx[4] = {hl, -hl, -hl, hl}, dy[4] = {hw, hw, -hw, -hw};
    for (int k = 0; k < 4; ++k) {
        float a = c * dx[k], b = s * dy[k];
        float t = bx[0] + a;
        cx[k] = t - b;
        a = s * dx[k]; b = c * dy[k];
        t = bx[1] + a;
        cy[k] = t + b;
    }
}

static float poly_clip_area(const float *ax, const float *ay, const float *bxs, const float *bys) {
    float px[16], py[16], qx[16], qy[16];
    int n = 4;
    for (int i = 0; i < 4; ++i) { px[i] = ax[i]; py[i] = ay[i]; }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const float q0x = bxs[e], q0y = bys[e], q1x = bxs[(e + 1) & 3], q1y = bys[(e + 1) & 3];
        const float ex = q1x - q0x, ey = q1y - q0y;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int ip = (i + n - 1) % n;
            float a = py[i] - q0y, b = px[i] - q0x;
            float t1 = ex * a, t2 = ey * b;
            const float cc = t1 - t2;
            a = py[ip] - q0y; b = px[ip] - q0x;
            t1 = ex * a; t2 = ey * b;
            const float cp = t1 - t2;
            const int in_c = cc >= 0.0f, in_p = cp >= 0.0f;
            if (in_c != in_p) {
                const float den = cp - cc;
                const float t = cp / den;
                float d = px[i] - px[ip];
                d = t * d;
                qx[m] = px[ip] + d;
                d = py[i] - py[ip];
                d = t * d;
                qy[m] = py[ip] + d;
                ++m;
            }
            if (in_c) { qx[m] = px[i]; qy[m] = py[i]; ++m; }
        }
        n = m;
        for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
    }
    if (n < 3) return 0.0f;
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        const float t1 = px[i] * py[j], t2 = px[j] * py[i];
        const float d = t1 - t2;
        sum = sum + d;
    }
    sum = sum < 0.0f ? -sum : sum;
    return 0.5f * sum;
}

static float iou_bev(const float *a, const float *b) {
    float ax[4], ay[4], bx[4], by[4];
    box_corners(a, ax, ay);
    box_corners(b, bx, by);
    const float inter = poly_clip_area(ax, ay, bx, by);
    const float aa = a[3] * a[4], ab = b[3] * b[4];
    float den = aa + ab;
    den = den - inter;
    if (!(den > 0.0f)) return 0.0f;
    return inter / den;
}

ORC_API void orc_sincos_r(const float *th, int n, float *s, float *c) {
    for (int i = 0; i < n; ++i) sincos_r(th[i], s + i, c + i);
}

ORC_API void orc_iou_bev(const float *a, const float *b, int n, float *out) {
    for (int i = 0; i < n; ++i) out[i] = iou_bev(a + (size_t)i * 9, b + (size_t)i * 9);
}

ORC_API void orc_nms_bev(const float *boxes, int B, int K, float iou_thr, float score_thr,
                         int32_t *keep, int32_t *order, int32_t *count) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *bx = boxes + (size_t)b * K * 9;
        int32_t *kp = keep + (size_t)b * K, *od = order + (size_t)b * K;
        int *rank = (int *)malloc(sizeof(int) * (size_t)K);
        int n = 0;
        for (int i = 0; i < K; ++i) { kp[i] = 0; od[i] = -1; if (bx[i * 9 + 7] >= score_thr) rank[n++] = i; }
        /* insertion sort: score descending, index ascending (stable) */
        for (int i = 1; i < n; ++i) {
            const int v = rank[i];
            int j = i;
            while (j > 0 && bx[rank[j - 1] * 9 + 7] < bx[v * 9 + 7]) { rank[j] = rank[j - 1]; --j; }
            rank[j] = v;
        }
        int nk = 0;
        for (int r = 0; r < n; ++r) {
            const int i = rank[r];
            int ok = 1;
            for (int k = 0; k < nk && ok; ++k)
                if (iou_bev(bx + (size_t)od[k] * 9, bx + (size_t)i * 9) > iou_thr) ok = 0;
            if (ok) { od[nk++] = i; kp[i] = 1; }
        }
        count[b] = nk;
        free(rank);
    }
}

/* SPEC.md §17 — ragged scenes -> fixed point count (integer arithmetic only: every implementation agrees).
 * points[total, C], offsets[B+1] -> out[B, n_points, C]. */
static inline uint32_t mix32(uint32_t a) {
    a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
    return a;
}
static inline uint32_t h17(uint32_t seed, uint32_t b, uint32_t i) {
    return mix32(seed * 0x9E3779B1u + b * 0x85EBCA77u + i * 0xC2B2AE3Du + 0x27D4EB2Fu);
}
ORC_API void orc_subsample_pad(const float *points, const int32_t *offsets, int B, int C, int n_points,
                               uint32_t seed, float *out) {
    for (int b = 0; b < B; ++b) {
        const int64_t o = offsets[b], n = (int64_t)offsets[b + 1] - o;
        float *dst = out + (size_t)b * n_points * C;
        if (n <= 0) { memset(dst, 0, sizeof(float) * (size_t)n_points * C); continue; }
        const int64_t r = (int64_t)(h17(seed, (uint32_t)b, 0xFFFFFFFFu) % (uint32_t)n);
        for (int64_t i = 0; i < n_points; ++i) {
            int64_t j;
            if (n == n_points) j = i;
            else if (n > n_points) j = (i * n + r) / n_points;
            else j = i < n ? i : (int64_t)(h17(seed, (uint32_t)b, (uint32_t)i) % (uint32_t)n);
            memcpy(dst + (size_t)i * C, points + (size_t)(o + j) * C, sizeof(float) * (size_t)C);
        }
    }
}
