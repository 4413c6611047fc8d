// Row-owned CSR device primitives shared by the SpMM, the baseline forward and the probe
// kernels.  Everything that has to be BIT-IDENTICAL between the baseline and a perturbed
// forward (so that rows outside a probe's 2-hop set difference to exactly 0, as they do in
// the reference: SURVEY.md section 7.2-1) goes through the functions in this header:
//
//   row_dot             : acc = init + sum_e val[e] * S[col[e], :]  as one k-ordered fmaf chain per column
//                         that STARTS from `init` (layer 1 passes the bias: z = fma(a_k, s_k, ... fma(a_0, s_0, b1)),
//                         which costs no instruction, where "+ b1" at the end costs one per column)
//   relu_w2_partial     : this lane's share of relu(z) . W2
//   group_sum<LPR>      : fixed butterfly (xor LPR/2 ... 1) -- fp add commutes, so every lane of
//                         the group ends with the same bits
//   row2_dot<C>         : layer-2 row (C <= 8 columns): 8 lanes per row, lane q owns entries
//                         e0+q, e0+q+8, ... (fmaf chain), then the xor 4,2,1 butterfly
//
// Built with -ffp-contract=off: the only fused operations are the explicit fmaf calls.
#pragma once
#include "lt_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LT_L2_LANES 8  // lanes cooperating on one layer-2 row

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

__device__ __forceinline__ f32x4 fma4(float a, f32x4 s, f32x4 acc) {
    acc.x = fmaf(a, s.x, acc.x);
    acc.y = fmaf(a, s.y, acc.y);
    acc.z = fmaf(a, s.z, acc.z);
    acc.w = fmaf(a, s.w, acc.w);
    return acc;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// Lane-local part of relu(z) . W2[:, c] for the 4 hidden columns this lane owns (z = pre-activation, bias
// included).  w2 points at W2p[(4*gl) * C]; rows beyond H are zero-padded so inactive columns add 0.
template <int CP>
__device__ __forceinline__ void relu_w2_partial(f32x4 z, const float *w2, int C, float (&part)[CP]) {
    const float h0 = fmaxf(z.x, 0.f);
    const float h1 = fmaxf(z.y, 0.f);
    const float h2 = fmaxf(z.z, 0.f);
    const float h3 = fmaxf(z.w, 0.f);
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        float p = 0.f;
        if (c < C) {
            p = h0 * w2[c];
            p = fmaf(h1, w2[C + c], p);
            p = fmaf(h2, w2[2 * C + c], p);
            p = fmaf(h3, w2[3 * C + c], p);
        }
        part[c] = p;
    }
}

// One CSR row against a dense [*, ld] matrix, 4 columns per lane, accumulated onto `init`.  `subst_col`/`subst_row`:
// entries whose column equals subst_col read subst_row instead of S (the perturbed S1 row of a
// probe); pass subst_col = -1 for none.  Pointer select, so the arithmetic is the same chain.
//
// Canonical order: the row is cut into segments of LT_ROW_SEG entries; a segment is one k-ordered fmaf chain
// (the first starts from `init`, the others from +0) and the segment sums are added in segment order,
// z = ((seg0 + seg1) + seg2) + ...  A row of up to LT_ROW_SEG entries is therefore one plain chain.  The cut is
// what lets FULL stage A give the segments of a hub row to different waves and still produce these bits.
// AHEAD = gathers issued before their FMAs run (the FMAs stay in entry order whatever AHEAD is: same chain, same
// bits).  4 suits the probe kernels (registers); the baseline kernels, whose run time on a graph with hub rows is
// the latency tail of their longest chains, use 8 / 16.
template <int AHEAD = 4>
__device__ __forceinline__ f32x4 seg_chain(const int32_t *__restrict__ col, const float *__restrict__ val,
                                           int e0, int e1, const float *__restrict__ S, int ld, int coff,
                                           bool active, int subst_col, const float *__restrict__ subst_row,
                                           f32x4 init) {
    f32x4 acc = init;
    int e = e0;
    for (; e + AHEAD <= e1; e += AHEAD) {
        int c[AHEAD];
        float a[AHEAD];
        f32x4 s[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            c[k] = col[e + k];
            a[k] = val[e + k];
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const float *src = (c[k] == subst_col) ? subst_row : S + (size_t)c[k] * ld;
            s[k] = active ? ld4(src + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) acc = fma4(a[k], s[k], acc);
    }
    if (AHEAD > 4) {   // the tail in blocks of 4, then singly
        for (; e + 4 <= e1; e += 4) {
            int c[4];
            float a[4];
            f32x4 s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c[k] = col[e + k];
                a[k] = val[e + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *src = (c[k] == subst_col) ? subst_row : S + (size_t)c[k] * ld;
                s[k] = active ? ld4(src + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma4(a[k], s[k], acc);
        }
    }
    for (; e < e1; ++e) {
        const int c = col[e];
        const float a = val[e];
        const float *src = (c == subst_col) ? subst_row : S + (size_t)c * ld;
        const f32x4 s = active ? ld4(src + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc = fma4(a, s, acc);
    }
    return acc;
}
// The same chain with every trip -- the last one too -- issuing AHEAD gathers at once: a trip past the end re-reads the
// last entry (clamped index, never used) instead of falling back to one gather per round trip.  For callers whose run time
// IS this chain's latency (FULL stage A recomputing a substituted probe: a d-entry row is ceil(d / AHEAD) round trips
// instead of d / 4 + d % 4).  e1 > e0; e0, e1 wave-uniform.
template <int AHEAD>
__device__ __forceinline__ f32x4 seg_chain_clamped(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                   int e0, int e1, const float *__restrict__ S, int ld, int coff,
                                                   int subst_col, const float *__restrict__ subst_row, f32x4 init) {
    f32x4 acc = init;
    for (int e = e0; e < e1; e += AHEAD) {
        int c[AHEAD];
        float a[AHEAD];
        f32x4 s[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const int ee = min(e + k, e1 - 1);
            c[k] = col[ee];
            a[k] = val[ee];
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const float *src = (c[k] == subst_col) ? subst_row : S + (size_t)c[k] * ld;
            s[k] = ld4(src + coff);
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k)
            if (e + k < e1) acc = fma4(a[k], s[k], acc);   // wave-uniform
    }
    return acc;
}
template <int AHEAD = 4>
__device__ __forceinline__ f32x4 row_dot(const int32_t *__restrict__ col,
                                         const float *__restrict__ val, int e0, int e1,
                                         const float *__restrict__ S, int ld, int coff,
                                         bool active, int subst_col,
                                         const float *__restrict__ subst_row,
                                         f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f}) {
    f32x4 total = seg_chain<AHEAD>(col, val, e0, min(e1, e0 + LT_ROW_SEG), S, ld, coff, active, subst_col, subst_row, init);
    for (int s0 = e0 + LT_ROW_SEG; s0 < e1; s0 += LT_ROW_SEG) {   // long rows only
        const f32x4 t = seg_chain<AHEAD>(col, val, s0, min(e1, s0 + LT_ROW_SEG), S, ld, coff, active, subst_col, subst_row,
                                         f32x4{0.f, 0.f, 0.f, 0.f});
        total.x += t.x; total.y += t.y; total.z += t.z; total.w += t.w;
    }
    return total;
}

// Layer-2 row: out[c] = sum_e val[e] * T[col[e]*C + c] for c < C, cooperative over LT_L2_LANES
// lanes (q = lane id inside the 8-lane group).  `lookup(col) -> const float*` returns the row of
// T to read for that column, which lets the probe kernels substitute perturbed rows without
// touching the arithmetic.  All 8 lanes return the full sums.
template <int CP, typename Lookup>
__device__ __forceinline__ void row2_dot(const int32_t *__restrict__ col,
                                         const float *__restrict__ val, int e0, int e1, int q,
                                         int C, Lookup lookup, float (&out)[CP]) {
#pragma unroll
    for (int c = 0; c < CP; ++c) out[c] = 0.f;
    int e = e0 + q;
    // four entries of the lane's chain in flight (a hub row is hundreds of entries per lane); the FMAs stay in
    // entry order, so the chain -- and its bits -- are unchanged
    for (; e + 3 * LT_L2_LANES < e1; e += 4 * LT_L2_LANES) {
        float a[4], tv[4][CP];
        const float *t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = val[e + k * LT_L2_LANES];
            t[k] = lookup(col[e + k * LT_L2_LANES], e + k * LT_L2_LANES);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < CP; ++c) tv[k][c] = (t[k] != nullptr && c < C) ? t[k][c] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (t[k] == nullptr) continue;  // DELTA mode: entries outside the probe's row set add nothing
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) out[c] = fmaf(a[k], tv[k][c], out[c]);
        }
    }
    for (; e < e1; e += LT_L2_LANES) {
        const float a = val[e];
        const float *t = lookup(col[e], e);
        if (t == nullptr) continue;
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) out[c] = fmaf(a, t[c], out[c]);
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) out[c] = group_sum<LT_L2_LANES>(out[c]);
}

// compile-time dispatch over lanes-per-row / padded class count
#define LT_DISPATCH_LPR(lpr, ...)                                         \
    switch (lpr) {                                                        \
        case 1: { constexpr int LPR_ = 1; __VA_ARGS__; } break;           \
        case 2: { constexpr int LPR_ = 2; __VA_ARGS__; } break;           \
        case 4: { constexpr int LPR_ = 4; __VA_ARGS__; } break;           \
        case 8: { constexpr int LPR_ = 8; __VA_ARGS__; } break;           \
        case 16: { constexpr int LPR_ = 16; __VA_ARGS__; } break;         \
        case 32: { constexpr int LPR_ = 32; __VA_ARGS__; } break;         \
        default: { constexpr int LPR_ = 64; __VA_ARGS__; } break;         \
    }
#define LT_DISPATCH_CP(cp, ...)                                           \
    switch (cp) {                                                         \
        case 2: { constexpr int CP_ = 2; __VA_ARGS__; } break;            \
        case 4: { constexpr int CP_ = 4; __VA_ARGS__; } break;            \
        default: { constexpr int CP_ = 8; __VA_ARGS__; } break;           \
    }

// lanes-per-row for a padded hidden width Hp (multiple of 4, <= 256): next pow2 >= Hp/4
static inline int lt_lpr_for(int Hp) {
    int need = Hp / 4, l = 1;
    while (l < need) l <<= 1;
    return l;
}
static inline int lt_cp_for(int C) { return C <= 2 ? 2 : (C <= 4 ? 4 : 8); }
