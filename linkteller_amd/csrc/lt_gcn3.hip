// The probe primitive for the 3-layer model (GCN3, gcn/models.py:28-46; reached with --n-layer 3,
// gcn_trainer.py:81-86):   out = A (relu(A (relu(A (X W1) + b1) W2) + b2) W3) + b3
//
// Same idea as the 2-layer SPARSE mode, one hop deeper.  X + pert_v differs from X in row v only, so
//   S1 = X W1              changes in row v,
//   H1 = relu(A S1 + b1)   changes on R1 = {r : A[r,v] != 0},          S2 = H1 W2 on the same rows,
//   H2 = relu(A S2 + b2)   changes on R2 = {r2 : A[r2,r] != 0, r in R1}, S3 = H2 W3 on the same rows,
//   out = A S3 + b3        changes on the 3-hop set; only the observed rows are formed.
// Per probe chunk (no read-back: the GEMM's row count stays on the device, lt_launch_gemm_mdev):  level-1 items (b, r in R1) -> H1x rows (k3_rows_relu with row v substituted) -> one MFMA GEMM
// S2x = H1x W2 -> R2 by marking (k3_mark2: the thread that flips a bit appends the item) -> level-2 items (b, r2):
// k3_stageB runs the layer-2 chain of row r2 with the rows of R1 looked up in S2x (bitmap with positions), relu,
// . W3 -> S3x[b][r2] -> k3_stageC: layer-3 row of each observed node with the rows of R2 looked up in S3x, minus the
// baseline logits, / delta, L2 norm.
//
// Arithmetic = the fp32 finite difference of the reference (attacker.py:105-106), evaluated only where it can be
// non-zero: every recomputed row goes through the SAME device functions as the baseline forward below (row_dot chains
// started from the bias, relu_w2_partial + group_sum, row2_dot), so rows a probe cannot reach difference to exactly 0
// and rows it can reach carry only the perturbation and fp32 rounding -- the reference's noise class.
#include <new>

#include "lt_items.hip.h"

#define LT_BLOCK 256
#define LT3_GRID 2048

struct lt_baseline3 {
    const lt_graph *g = nullptr;
    int32_t n = 0, F = 0, H1 = 0, H2 = 0, C = 0, Hp1 = 0, Hp2 = 0;
    const float *X = nullptr;
    int64_t ldx = 0;
    const float *W1 = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr, *W3 = nullptr, *b3 = nullptr;
    // owned
    float *S1 = nullptr;    // [n, Hp1]  X W1
    float *Act1 = nullptr;  // [n, Hp1]  H1 = relu(A S1 + b1)
    float *S2 = nullptr;    // [n, Hp2]  H1 W2
    float *Z2 = nullptr;    // [n, Hp2]  A S2 + b2
    float *S3 = nullptr;    // [n, C]    relu(Z2) W3
    float *OUT = nullptr;   // [n, C]    A S3 + b3
    float *b1p = nullptr, *b2p = nullptr, *W3p = nullptr;   // zero-padded to Hp1 / Hp2 / [Hp2, C]
    float *slabs = nullptr;     // split-K partials of X W1
    float *seg_part = nullptr;  // [g->p_n_seg, Hp2] segment sums of the hub rows (layer 2)
    // fp64 baseline for the exact (delta) propagation, lt_baseline3_enable_fp64: the layer-1 pre-activation Z1d lives in an
    // inner 2-layer baseline handle (its fp64 routes: feature rows / matrix cores), the layer-2 one here
    lt_baseline *l1 = nullptr;
    double *S2d = nullptr;      // [n, Hp2]  relu(Z1d) W2
    double *Z2d = nullptr;      // [n, Hp2]  A S2d + b2
    double *seg2d = nullptr;    // [lt_f64_seg_rows(g), Hp2]
    bool fp64_fresh = false;
    // lt_baseline3_refresh launches nothing (round 6, as lt_baseline_refresh since ABI 2): the padded parameters and the fp32 forward
    // are recomputed on the stream of the first call that reads them -- a LT_MODE_DELTA build reads no fp32 layer at all
    bool pad_fresh = false, fp32_fresh = false;
};

// h[dst] = relu(A[r,:] S + bias): all rows (items == NULL), or the level-1 items of a probe chunk, whose row reads
// Sp[b] in place of S[v] (the perturbed row of probe b)
template <int LPR>
__global__ __launch_bounds__(LT_BLOCK) void k3_rows_relu(
    int n_rows, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S, int Hp, const float *__restrict__ biasp, float *__restrict__ out,
    const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow, const int32_t *__restrict__ probes, int nb,
    const int32_t *__restrict__ off, const float *__restrict__ Sp) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    const bool items = off != nullptr;
    const int total = items ? off[nb] : n_rows;
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    for (int base = wave0 * RPW; base < total; base += nwaves * RPW) {
        const int it = base + lane / LPR;
        if (it >= total || !active) continue;
        int r = it, v = -1;
        const float *sub = nullptr;
        if (items) {
            const int b = find_probe(off, nb, it);
            v = probes[b];
            r = trow[tptr[v] + (it - off[b])];
            sub = Sp + (size_t)b * Hp;
        }
        const f32x4 z = row_dot<8>(col, val, rowptr[r], rowptr[r + 1], S, Hp, coff, true, v, sub, ld4(biasp + coff));
        f32x4 h;
        h.x = fmaxf(z.x, 0.f); h.y = fmaxf(z.y, 0.f); h.z = fmaxf(z.z, 0.f); h.w = fmaxf(z.w, 0.f);
        *reinterpret_cast<f32x4 *>(out + (size_t)it * Hp + coff) = h;
    }
}

// R2 of every probe: for each level-1 item (b, r) the rows r2 that read row r (CSC column r) are marked in the probe's bitmap
// (k3_mark2); k3_list2 then turns every probe's bitmap row into its level-2 items (b, r2) (order irrelevant: every item is
// computed on its own and lands in S3x[b][r2]).
// (Up to round 5 the thread that flipped a bit appended the item itself: one add per wave and trip on ONE word -- 9 K of them at
// twitch size, which that word's L2 channel serves one after the other: 105 us of a 0.35 ms GCN3 build.  Now the marks are
// fire-and-forget and the list costs one add per 256 bitmap words that hold anything.)
__global__ __launch_bounds__(LT_BLOCK) void k3_mark2(
    const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow, const int32_t *__restrict__ probes, int nb,
    const int32_t *__restrict__ off, int words, uint32_t *__restrict__ bits2, int32_t *__restrict__ n_items2) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_items2 = 0;      // (k3_list2's cursor: cleared here instead of by a memset of its own)
    const int total = off[nb];
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    for (int it = wave0; it < total; it += nwaves) {
        const int b = find_probe(off, nb, it);
        const int v = probes[b];
        const int r = trow[tptr[v] + (it - off[b])];
        for (int t = tptr[r] + lane; t < tptr[r + 1]; t += 64) {
            const int r2 = trow[t];
            atomicOr(&bits2[(size_t)b * words + (r2 >> 5)], 1u << (r2 & 31));
        }
    }
}
// one block per probe: its bitmap row, 256 words at a time -> items2 (a place per set bit behind one add per stretch)
__global__ __launch_bounds__(LT_BLOCK) void k3_list2(int words, const uint32_t *__restrict__ bits2, int2 *__restrict__ items2,
                                                     int32_t *__restrict__ n_items2) {
    __shared__ int s_wave[LT_BLOCK / 64];
    __shared__ int s_base;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t *row = bits2 + (size_t)b * words;
    for (int w0 = 0; w0 < words; w0 += LT_BLOCK) {          // (block-uniform)
        const int w = w0 + threadIdx.x;
        uint32_t m = w < words ? row[w] : 0u;
        const int cnt = __popc(m);
        int incl = cnt;                                      // inclusive prefix over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wid] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < LT_BLOCK / 64; ++k) {
            before += k < wid ? s_wave[k] : 0;
            all += s_wave[k];
        }
        if (all > 0 && threadIdx.x == 0) s_base = atomicAdd(n_items2, all);
        __syncthreads();
        if (cnt > 0) {
            int pos = s_base + before + incl - cnt;
            while (m) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1u;
                items2[pos++] = make_int2(b, w * 32 + bit);
            }
        }
        __syncthreads();                                     // (s_wave / s_base are rewritten by the next stretch)
    }
}

// One CSR row against S2 with the rows of R1(b) replaced by their perturbed versions (S2x, found through the probe's
// bitmap with positions): row_dot's canonical order -- 128-entry fmaf chains, the first started from `init`, their
// sums added in segment order -- so an unperturbed row gives the bits of the baseline kernel.
__device__ __forceinline__ f32x4 row_dot_lookup(const int32_t *__restrict__ col, const float *__restrict__ val, int e0,
                                                int e1, const float *__restrict__ S, int ld, int coff,
                                                const uint2 *__restrict__ mb, const float *__restrict__ items, f32x4 init) {
    f32x4 total = init;
    for (int s0 = e0; s0 < e1 || s0 == e0; s0 += LT_ROW_SEG) {
        const int s1 = min(e1, s0 + LT_ROW_SEG);
        f32x4 acc = s0 == e0 ? init : f32x4{0.f, 0.f, 0.f, 0.f};
        int e = s0;
        for (; e + 4 <= s1; e += 4) {
            float a[4];
            f32x4 s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = col[e + k];
                a[k] = val[e + k];
                const int p = bits_pos(mb, c);
                s[k] = ld4((p >= 0 ? items + (size_t)p * ld : S + (size_t)c * ld) + coff);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma4(a[k], s[k], acc);
        }
        for (; e < s1; ++e) {
            const int c = col[e];
            const int p = bits_pos(mb, c);
            acc = fma4(val[e], ld4((p >= 0 ? items + (size_t)p * ld : S + (size_t)c * ld) + coff), acc);
        }
        if (s0 == e0) total = acc;
        else { total.x += acc.x; total.y += acc.y; total.z += acc.z; total.w += acc.w; }
        if (s1 >= e1) break;
    }
    return total;
}

// level-2 items: S3x[b][r2] = relu(A[r2,:] S2' + b2) . W3
template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k3_stageB(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S2, int Hp2, const float *__restrict__ b2p, const float *__restrict__ W3p, int C,
    const int32_t *__restrict__ off, const float *__restrict__ S2x, const uint2 *__restrict__ bits1, int words,
    const int2 *__restrict__ items2, const int32_t *__restrict__ n_items2, int n, float *__restrict__ S3x) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const int coff = 4 * gl;
    const bool active = coff < Hp2;
    const int total = *n_items2;
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    for (int base = wave0 * RPW; base < total; base += nwaves * RPW) {
        const int it = base + lane / LPR;
        const bool live = it < total;   // group-uniform; dead groups still join the shuffles
        float part[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = 0.f;
        int b = 0, r2 = 0;
        if (live) {
            const int2 w = items2[it];
            b = w.x; r2 = w.y;
            if (active) {
                const f32x4 z = row_dot_lookup(col, val, rowptr[r2], rowptr[r2 + 1], S2, Hp2, coff,
                                               bits1 + (size_t)b * words, S2x + (size_t)off[b] * Hp2, ld4(b2p + coff));
                relu_w2_partial<CP>(z, W3p + (size_t)coff * C, C, part);
            }
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
        if (live && gl == 0) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) S3x[((size_t)b * n + r2) * C + c] = part[c];
        }
    }
}

// observed rows: out[b][j] = || (A[u,:] S3' + b3 - OUT[u]) / delta ||, 8 lanes per (probe, observed node)
template <int CP>
__global__ __launch_bounds__(LT_BLOCK) void k3_stageC(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S3, int C, const float *__restrict__ b3, const float *__restrict__ OUT, int nb, int n,
    const float *__restrict__ S3x, const uint32_t *__restrict__ bits2, int words, const int32_t *__restrict__ observe,
    int n_obs, float delta, float *__restrict__ out, long ldo) {
    const long gid = ((long)blockIdx.x * LT_BLOCK + threadIdx.x) / LT_L2_LANES;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    if (gid >= (long)nb * n_obs) return;
    const int b = (int)(gid / n_obs), j = (int)(gid % n_obs);
    const int u = observe[j];
    const uint32_t *mb = bits2 + (size_t)b * words;
    const float *mine = S3x + (size_t)b * n * C;
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    auto member = [&](int c) { return ((mb[c >> 5] >> (c & 31)) & 1u) != 0u; };
    bool touch = false;
    for (int e = e0 + q; e < e1; e += LT_L2_LANES) touch |= member(col[e]);
    int t = touch ? 1 : 0;
#pragma unroll
    for (int m = LT_L2_LANES / 2; m >= 1; m >>= 1) t |= __shfl_xor(t, m, 64);
    float res = 0.f;
    if (t) {
        float acc[CP];
        row2_dot<CP>(col, val, e0, e1, q, C,
                     [&](int c, int) { return member(c) ? mine + (size_t)c * C : S3 + (size_t)c * C; }, acc);
        res = diff_norm<CP>(acc, b3, OUT + (size_t)u * C, C, delta);
    }
    if (q == 0) out[(long)b * ldo + j] = res;
}

// ---- exact propagation (LT_MODE_DELTA) for three layers ---------------------------------------------------------------
// The perturbation itself is pushed through the layers: dS1[v] = d * S1[v]; on the level-1 items dZ1[r] = A[r,v] dS1[v] and
// dH1 = relu(Z1 + dZ1) - relu(Z1) evaluated piecewise on the fp64 pre-activation (the kink test of the 2-layer delta mode);
// dS2 = dH1 W2 (the MFMA GEMM over the items); on the level-2 items dZ2[q] = sum over the members r of R1 in row q of
// A[q,r] dS2[r], dH2 likewise on the fp64 layer-2 pre-activation Z2d, dS3 = dH2 W3; the observed rows sum A[u,q] dS3[q] over
// the members of R2.  No subtraction of nearly equal numbers anywhere.
__device__ __forceinline__ float relu_diff(double z, float dz) {
    const double z1 = z + (double)dz;
    return z > 0.0 ? (z1 > 0.0 ? dz : (float)(-z)) : (z1 > 0.0 ? (float)z1 : 0.f);
}
template <int LPR>
__global__ __launch_bounds__(LT_BLOCK) void k3d_items1(
    const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow, const float *__restrict__ tval,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off, const double *__restrict__ Spd,
    const double *__restrict__ Z1d, int Hp, float delta, float *__restrict__ dH1x) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int coff = 4 * (lane & (LPR - 1));
    const int total = off[nb];
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    for (int base = wave0 * RPW; base < total; base += nwaves * RPW) {
        const int it = base + lane / LPR;
        if (it >= total || coff >= Hp) continue;
        const int b = find_probe(off, nb, it);
        const int v = probes[b];
        const int t = tptr[v] + (it - off[b]);
        const int r = trow[t];
        const float arv = tval[t];
        const double *sp = Spd + (size_t)b * Hp + coff, *zp = Z1d + (size_t)r * Hp + coff;
        f32x4 dh;
#pragma unroll
        for (int k = 0; k < 4; ++k) dh[k] = relu_diff(zp[k], arv * (delta * (float)sp[k]));
        *reinterpret_cast<f32x4 *>(dH1x + (size_t)it * Hp + coff) = dh;
    }
}

template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k3d_stageB(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const double *__restrict__ Z2d, int Hp2, const float *__restrict__ W3p, int C, const int32_t *__restrict__ off,
    const float *__restrict__ dS2x, const uint2 *__restrict__ bits1, int words, const int2 *__restrict__ items2,
    const int32_t *__restrict__ n_items2, int n, float *__restrict__ dS3x) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const int coff = 4 * gl;
    const bool active = coff < Hp2;
    const int total = *n_items2;
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    // (the lane's rows of W3 do not depend on the item: read once)
    float w3r[4][CP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CP; ++c) w3r[k][c] = (active && c < C) ? W3p[(size_t)(coff + k) * C + c] : 0.f;
    for (int base = wave0 * RPW; base < total; base += nwaves * RPW) {
        const int it = base + lane / LPR;
        const bool live = it < total;   // group-uniform; dead groups still join the shuffles
        float part[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = 0.f;
        int b = 0, q = 0;
        if (live) {
            const int2 w = items2[it];
            b = w.x; q = w.y;
            const uint2 *mb = bits1 + (size_t)b * words;
            const float *items = dS2x + (size_t)off[b] * Hp2;
            f32x4 dz = {0.f, 0.f, 0.f, 0.f};
            // (the row's fp64 pre-activation goes out in front of the look-ups: nothing in them depends on it)
            double zq[4] = {0.0, 0.0, 0.0, 0.0};
            if (active) {
                const double2 z01 = *reinterpret_cast<const double2 *>(Z2d + (size_t)q * Hp2 + coff);
                const double2 z23 = *reinterpret_cast<const double2 *>(Z2d + (size_t)q * Hp2 + coff + 2);
                zq[0] = z01.x; zq[1] = z01.y; zq[2] = z23.x; zq[3] = z23.y;
            }
            // (round 6) the group's lanes look up LPR entries of the row side by side (column, membership word: two trips for the
            // stretch instead of two per entry -- the 18 dependent pairs of an average row were this launch's 94 us), then the members'
            // items are added in entry order: the same chain
            const int e_end = rowptr[q + 1], gbase = lane & ~(LPR - 1);
            for (int e0 = rowptr[q]; e0 < e_end; e0 += LPR) {            // (group-uniform)
                const int e = e0 + gl;
                int p = -1;
                float a = 0.f;
                if (e < e_end) {
                    p = bits_pos(mb, col[e]);
                    a = val[e];
                }
                unsigned long long hits = __ballot(p >= 0) >> gbase;
                if (LPR < 64) hits &= (1ull << LPR) - 1ull;
                while (hits) {                                           // members of R1 only, in entry order (group-uniform)
                    const int k = __ffsll((long long)hits) - 1;
                    hits &= hits - 1ull;
                    const int pk = __shfl(p, gbase + k, 64);
                    const float ak = __shfl(a, gbase + k, 64);
                    if (active) dz = fma4(ak, ld4(items + (size_t)pk * Hp2 + coff), dz);
                }
            }
            if (active) {
                float dh[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) dh[k] = relu_diff(zq[k], dz[k]);
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) {
                        float pr = dh[0] * w3r[0][c];
                        pr = fmaf(dh[1], w3r[1][c], pr);
                        pr = fmaf(dh[2], w3r[2][c], pr);
                        pr = fmaf(dh[3], w3r[3][c], pr);
                        part[c] = pr;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
        if (live && gl == 0) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) dS3x[((size_t)b * n + q) * C + c] = part[c];
        }
    }
}

template <int CP>
__global__ __launch_bounds__(LT_BLOCK) void k3d_stageC(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val, int C, int nb, int n,
    const float *__restrict__ dS3x, const uint32_t *__restrict__ bits2, int words, const int32_t *__restrict__ observe,
    int n_obs, float delta, float *__restrict__ out, long ldo) {
    const long gid = ((long)blockIdx.x * LT_BLOCK + threadIdx.x) / LT_L2_LANES;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    if (gid >= (long)nb * n_obs) return;
    const int b = (int)(gid / n_obs), j = (int)(gid % n_obs);
    const int u = observe[j];
    const uint32_t *mb = bits2 + (size_t)b * words;
    const float *mine = dS3x + (size_t)b * n * C;
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    auto member = [&](int c) { return ((mb[c >> 5] >> (c & 31)) & 1u) != 0u; };
    // (round 6: no pass over the row in front that only asks whether any entry is a member -- three hops reach most pairs at twitch
    // size, and a pair nothing reaches comes out of the sums as +0 all the same: 0 / delta, fma, sqrt)
    float acc[CP];
    row2_dot<CP>(col, val, e0, e1, q, C,
                 [&](int c, int) { return member(c) ? mine + (size_t)c * C : (const float *)nullptr; }, acc);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) {
            const float d = acc[c] / delta;
            ss = fmaf(d, d, ss);
        }
    if (q == 0) out[(long)b * ldo + j] = sqrtf(ss);
}

__global__ void k3_pad(const float *__restrict__ b1, int H1, int Hp1, const float *__restrict__ b2, int H2, int Hp2,
                       const float *__restrict__ W3, int C, float *__restrict__ b1p, float *__restrict__ b2p,
                       float *__restrict__ W3p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Hp1) b1p[i] = i < H1 ? b1[i] : 0.f;
    if (i < Hp2) b2p[i] = i < H2 ? b2[i] : 0.f;
    if (i < Hp2 * C) W3p[i] = (i / C) < H2 ? W3[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
static void free_baseline3(lt_baseline3 *b) {
    if (!b) return;
    (void)hipFree(b->S1); (void)hipFree(b->Act1); (void)hipFree(b->S2); (void)hipFree(b->Z2); (void)hipFree(b->S3);
    (void)hipFree(b->OUT); (void)hipFree(b->b1p); (void)hipFree(b->b2p); (void)hipFree(b->W3p); (void)hipFree(b->slabs);
    (void)hipFree(b->seg_part);
    (void)hipFree(b->S2d); (void)hipFree(b->Z2d); (void)hipFree(b->seg2d);
    if (b->l1) (void)lt_baseline_destroy(b->l1);
    delete b;
}

static inline unsigned blocks_for(long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

extern "C" int lt_baseline3_refresh(lt_baseline3 *b, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline3_refresh: baseline is NULL");
    b->fp64_fresh = b->pad_fresh = b->fp32_fresh = false;
    if (b->l1) (void)lt_baseline_refresh(b->l1, stream);
    return LT_OK;
}

// b1 / b2 / W3 zero-padded to the padded widths (read by the fp32 forward, the fp64 layer-2 pre-activation and the probes' kernels)
static int ensure3_pad(const lt_baseline3 *cb, hipStream_t st) {
    lt_baseline3 *b = const_cast<lt_baseline3 *>(cb);   // cache state only
    if (b->pad_fresh || b->n == 0) return LT_OK;
    const int mx = (b->Hp1 > b->Hp2 * b->C ? b->Hp1 : b->Hp2 * b->C);
    hipLaunchKernelGGL(k3_pad, dim3((mx + 255) / 256), dim3(256), 0, st, b->b1, b->H1, b->Hp1, b->b2, b->H2, b->Hp2, b->W3,
                       b->C, b->b1p, b->b2p, b->W3p);
    LT_CHECK_LAUNCH();
    b->pad_fresh = true;
    return LT_OK;
}

// the unperturbed fp32 forward (the fp32 finite difference's baseline, lt_baseline3_logits)
static int ensure3_fp32(const lt_baseline3 *cb, hipStream_t st) {
    lt_baseline3 *b = const_cast<lt_baseline3 *>(cb);   // cache state only
    if (b->fp32_fresh || b->n == 0) return LT_OK;
    int rc = ensure3_pad(b, st);
    if (rc) return rc;
    const lt_graph *g = b->g;
    const int n = b->n;
    // S1 = X W1 (pad columns zero)
    if (b->Hp1 != b->H1) LT_HIP(hipMemsetAsync(b->S1, 0, (size_t)n * b->Hp1 * sizeof(float), st));
    rc = b->slabs ? lt_launch_gemm_splitk(b->X, b->ldx, b->W1, b->H1, b->S1, b->Hp1, n, b->H1, b->F,
                                          lt_gemm_pick_kslice(n, b->H1, b->F), b->slabs, st)
                  : lt_launch_gemm(b->X, b->ldx, b->W1, b->H1, b->S1, b->Hp1, n, b->H1, b->F, st);
    if (rc) return rc;
    // H1 = relu(A S1 + b1)
    const int lpr1 = lt_lpr_for(b->Hp1);
    LT_DISPATCH_LPR(lpr1, hipLaunchKernelGGL((k3_rows_relu<LPR_>), dim3(blocks_for(n, (LT_BLOCK / 64) * (64 / lpr1))),
                                             dim3(LT_BLOCK), 0, st, n, g->rowptr, g->col, g->val, b->S1, b->Hp1, b->b1p, b->Act1,
                                             (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0,
                                             (const int32_t *)nullptr, (const float *)nullptr));
    LT_CHECK_LAUNCH();
    // S2 = H1 W2
    if (b->Hp2 != b->H2) LT_HIP(hipMemsetAsync(b->S2, 0, (size_t)n * b->Hp2 * sizeof(float), st));
    rc = lt_launch_gemm(b->Act1, b->Hp1, b->W2, b->H2, b->S2, b->Hp2, n, b->H2, b->H1, st);
    if (rc) return rc;
    // S3 = relu(A S2 + b2) W3, OUT = A S3 + b3: the fused layer kernels of the 2-layer path
    rc = lt_launch_layer1(g, b->S2, b->Hp2, b->b2p, b->W3p, b->C, b->Z2, b->S3, st, b->seg_part);
    if (rc) return rc;
    rc = lt_launch_layer2(g, b->S3, b->C, b->b3, b->OUT, st);
    if (rc) return rc;
    b->fp32_fresh = true;
    return LT_OK;
}

extern "C" int lt_baseline3_create(const lt_graph *g, const float *X, int64_t ldx, int32_t F, const float *W1,
                                   const float *b1, int32_t H1, const float *W2, const float *b2, int32_t H2,
                                   const float *W3, const float *b3, int32_t C, void *stream, lt_baseline3 **out) {
    LT_REQUIRE(out != nullptr, "lt_baseline3_create: out is NULL");
    *out = nullptr;
    LT_REQUIRE(g != nullptr, "lt_baseline3_create: graph is NULL");
    LT_REQUIRE(F > 0 && H1 > 0 && H2 > 0 && C > 0, "lt_baseline3_create: F=%d H1=%d H2=%d C=%d must be positive", F, H1, H2, C);
    if (H1 > LT_MAX_H || H2 > LT_MAX_H || C > LT_MAX_C)
        return lt_set_error(LT_ERR_UNSUPPORTED, "lt_baseline3_create: H1=%d H2=%d C=%d (supported: H <= %d, C <= %d)", H1, H2, C,
                            LT_MAX_H, LT_MAX_C);
    LT_REQUIRE(X && W1 && b1 && W2 && b2 && W3 && b3, "lt_baseline3_create: NULL tensor pointer");
    LT_REQUIRE(ldx >= F, "lt_baseline3_create: ldx=%lld < F=%d", (long long)ldx, F);
    (void)lt_node_err_dev();      // (allocated outside any stream capture)
    lt_baseline3 *b = new (std::nothrow) lt_baseline3();
    if (!b) return lt_set_error(LT_ERR_NOMEM, "lt_baseline3_create: out of host memory");
    b->g = g; b->n = g->n; b->F = F; b->H1 = H1; b->H2 = H2; b->C = C;
    b->Hp1 = lt_round_up(H1, 4); b->Hp2 = lt_round_up(H2, 4);
    b->X = X; b->ldx = ldx; b->W1 = W1; b->b1 = b1; b->W2 = W2; b->b2 = b2; b->W3 = W3; b->b3 = b3;
    const size_t n1 = (size_t)(b->n > 0 ? b->n : 1);
#define B3_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            free_baseline3(b);                                                              \
            return lt_set_error(LT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)
    B3_HIP(hipMalloc((void **)&b->S1, n1 * b->Hp1 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->Act1, n1 * b->Hp1 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->S2, n1 * b->Hp2 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->Z2, n1 * b->Hp2 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->S3, n1 * C * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->OUT, n1 * C * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->b1p, (size_t)b->Hp1 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->b2p, (size_t)b->Hp2 * sizeof(float)));
    B3_HIP(hipMalloc((void **)&b->W3p, (size_t)b->Hp2 * C * sizeof(float)));
    if (g->p_n_seg > 0) B3_HIP(hipMalloc((void **)&b->seg_part, (size_t)g->p_n_seg * b->Hp2 * sizeof(float)));
    const size_t sb = lt_gemm_splitk_slab_bytes(b->n, H1, F, lt_gemm_pick_kslice(b->n, H1, F));
    if (sb) B3_HIP(hipMalloc((void **)&b->slabs, sb));
#undef B3_HIP
    const int rc = lt_baseline3_refresh(b, stream);
    if (rc) {
        free_baseline3(b);
        return rc;
    }
    *out = b;
    return LT_OK;
}

extern "C" int lt_baseline3_destroy(lt_baseline3 *b) {
    free_baseline3(b);
    return LT_OK;
}

extern "C" int lt_baseline3_logits(const lt_baseline3 *b, float *dst, void *stream) {
    LT_REQUIRE(b != nullptr && dst != nullptr, "lt_baseline3_logits: NULL argument");
    if (b->n == 0) return LT_OK;
    { const int rc = ensure3_fp32(b, (hipStream_t)stream); if (rc) return rc; }
    LT_HIP(hipMemcpyAsync(dst, b->OUT, (size_t)b->n * b->C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return LT_OK;
}

// fp64 baseline of the first two layers (the kink tests of LT_MODE_DELTA): Z1d through an inner 2-layer handle over the
// same borrowed X / W1 / b1 (so that it takes the fp64 product routes of lt_fp64.hip), S2d = relu(Z1d) W2 and
// Z2d = A S2d + b2 here.  Costs one more copy of the layer-1 buffers; redone after every lt_baseline3_refresh when next needed.
extern "C" int lt_baseline3_enable_fp64(lt_baseline3 *b, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline3_enable_fp64: baseline is NULL");
    if (b->Z2d) return LT_OK;
    lt_baseline *l1 = nullptr;
    // (W2 / b2 of the inner handle are never used for arithmetic: only its fp64 pre-activation is read; C = 1 keeps its
    // padding kernel inside W2's first H1 floats)
    int rc = lt_baseline_create(b->g, b->X, b->ldx, b->F, b->W1, b->b1, b->H1, b->W2, b->b2, 1, stream, &l1);
    if (rc) return rc;
    l1->no_agg = true;
    rc = lt_baseline_enable_fp64(l1, stream);
    if (rc) { (void)lt_baseline_destroy(l1); return rc; }
    const size_t n1 = (size_t)(b->n > 0 ? b->n : 1);
    double *s2d = nullptr, *z2d = nullptr, *seg = nullptr;
    hipError_t e = hipMalloc((void **)&s2d, n1 * b->Hp2 * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(s2d, 0, n1 * b->Hp2 * sizeof(double), (hipStream_t)stream);   // pad columns stay zero
    if (e == hipSuccess) e = hipMalloc((void **)&z2d, n1 * b->Hp2 * sizeof(double));
    if (e == hipSuccess && lt_f64_seg_rows(b->g) > 0) e = hipMalloc((void **)&seg, (size_t)lt_f64_seg_rows(b->g) * b->Hp2 * sizeof(double));
    if (e != hipSuccess) {
        (void)hipFree(s2d); (void)hipFree(z2d); (void)hipFree(seg); (void)lt_baseline_destroy(l1);
        return lt_set_error(LT_ERR_HIP, "lt_baseline3_enable_fp64: hipMalloc failed: %s", hipGetErrorString(e));
    }
    b->l1 = l1; b->S2d = s2d; b->Z2d = z2d; b->seg2d = seg;
    b->fp64_fresh = false;
    return LT_OK;
}

static int ensure3_fp64(const lt_baseline3 *cb, hipStream_t st) {
    lt_baseline3 *b = const_cast<lt_baseline3 *>(cb);   // cache state only
    if (b->fp64_fresh || b->n == 0) return LT_OK;
    int rc = ensure3_pad(b, st);                        // (b2p below)
    if (rc) return rc;
    rc = lt_fp64_form_all(b->l1, st);                   // Z1d, every row
    if (rc) return rc;
    rc = lt_launch_gemm_f64_dense(b->l1->Z1d, (long)b->Hp1, b->n, b->W2, (long)b->H2, b->H2, b->H1, nullptr, b->S2d, (long)b->Hp2, 1, st);
    if (rc) return rc;
    rc = lt_launch_spmm_f64(b->g, b->S2d, b->Hp2, b->b2p, b->Z2d, b->seg2d, st);
    if (rc) return rc;
    b->fp64_fresh = true;
    return LT_OK;
}

// ---- workspace ------------------------------------------------------------------------------
struct infl3_ws {
    float *Sp, *slabs, *H1x, *S2x, *S3x;
    double *Spd;            // delta: fp64 product rows of the chunk's probes [chunk, Hp1]
    int32_t *off, *n_items2;
    uint2 *bits1;
    uint32_t *bits2;
    int2 *items2;
    int32_t *probes_s, *obs_s;    // the call's lists, every id checked against [0, n) (lt_items.hip.h k_check_nodes)
    size_t bytes;
    int chunk;
};

static int probe_kslice3(const lt_baseline3 *b) { return lt_gemm_pick_kslice(b->n, b->H1, b->F); }

static infl3_ws carve3(void *base, const lt_baseline3 *b, int n_probe, int n_obs) {
    infl3_ws w = {};
    const size_t n = (size_t)b->n, C = (size_t)b->C, Hp1 = (size_t)b->Hp1, Hp2 = (size_t)b->Hp2;
    const size_t maxc = (size_t)(b->g->max_col_nnz > 0 ? b->g->max_col_nnz : 1);
    const size_t words = (n + 31) / 32;
    const size_t splitk = (((size_t)b->F + probe_kslice3(b) - 1) / probe_kslice3(b)) * (size_t)b->H1;
    const size_t per_probe = (Hp1 + splitk + maxc * (Hp1 + Hp2) + n * C) * sizeof(float) + n * sizeof(int2) +
                             words * (sizeof(uint2) + sizeof(uint32_t)) + sizeof(int32_t);
    size_t chunk = (size_t)lt_tune().chunk_budget / per_probe;
    if (chunk < 1) chunk = 1;
    if (chunk > 65534) chunk = 65534;
    if (chunk > (size_t)(n_probe > 0 ? n_probe : 1)) chunk = (size_t)(n_probe > 0 ? n_probe : 1);
    w.chunk = (int)chunk;
    size_t offb = 0;
    char *p = (char *)base;
    auto take = [&](size_t bytes) {
        void *q = p ? (void *)(p + offb) : nullptr;
        offb += lt_align_up(bytes ? bytes : 1, 256);
        return q;
    };
    w.Sp = (float *)take(chunk * Hp1 * sizeof(float));
    w.Spd = (double *)take(chunk * Hp1 * sizeof(double));
    w.slabs = (float *)take(lt_gemm_splitk_slab_bytes((int)chunk, b->H1, b->F, probe_kslice3(b)));
    w.off = (int32_t *)take((chunk + 1) * sizeof(int32_t));
    w.n_items2 = (int32_t *)take(sizeof(int32_t));
    w.H1x = (float *)take(chunk * maxc * Hp1 * sizeof(float));
    w.S2x = (float *)take(chunk * maxc * Hp2 * sizeof(float));
    w.bits1 = (uint2 *)take(chunk * words * sizeof(uint2));
    w.bits2 = (uint32_t *)take(chunk * words * sizeof(uint32_t));
    w.items2 = (int2 *)take(chunk * n * sizeof(int2));
    w.S3x = (float *)take(chunk * n * C * sizeof(float));
    w.probes_s = (int32_t *)take((size_t)(n_probe > 0 ? n_probe : 1) * sizeof(int32_t));
    w.obs_s = (int32_t *)take((size_t)(n_obs > 0 ? n_obs : 1) * sizeof(int32_t));
    w.bytes = offb;
    return w;
}

extern "C" size_t lt_influence3_workspace_bytes(const lt_baseline3 *b, int32_t n_probe, int32_t n_obs) {
    if (!b || n_probe < 0 || n_obs < 0) return 0;
    return carve3(nullptr, b, n_probe, n_obs).bytes;
}

extern "C" int lt_influence3_rows(const lt_baseline3 *b, const int32_t *probe_nodes, int32_t n_probe,
                                  const int32_t *observe_nodes, int32_t n_obs, float delta, float *out, int64_t ldo,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    return lt_influence3_rows_mode(b, probe_nodes, n_probe, observe_nodes, n_obs, delta, LT_MODE_SPARSE, out, ldo, workspace,
                                   workspace_bytes, stream);
}

extern "C" int lt_influence3_rows_mode(const lt_baseline3 *b, const int32_t *probe_nodes, int32_t n_probe,
                                       const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode, float *out,
                                       int64_t ldo, void *workspace, size_t workspace_bytes, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_influence3_rows: baseline is NULL");
    lt_prof_call prof_call_;
    LT_REQUIRE(mode == LT_MODE_SPARSE || mode == LT_MODE_FULL || mode == LT_MODE_DELTA, "lt_influence3_rows_mode: unknown mode %d", mode);
    const bool exact = mode == LT_MODE_DELTA;
    LT_REQUIRE(!exact || b->Z2d != nullptr, "lt_influence3_rows_mode: LT_MODE_DELTA needs lt_baseline3_enable_fp64");
    LT_REQUIRE(n_probe >= 0 && n_obs >= 0, "lt_influence3_rows: negative count");
    LT_REQUIRE(delta != 0.f && delta == delta, "lt_influence3_rows: delta must be a non-zero number");
    if (n_probe == 0 || n_obs == 0) return LT_OK;
    LT_REQUIRE(probe_nodes && observe_nodes && out, "lt_influence3_rows: NULL pointer");
    LT_REQUIRE(ldo >= n_obs, "lt_influence3_rows: ldo=%lld < n_obs=%d", (long long)ldo, n_obs);
    LT_REQUIRE(b->n > 0, "lt_influence3_rows: empty graph");
    { const int rc = lt_node_err_pending(); if (rc) return rc; }     // an earlier call's list held an id out of range
    const size_t need = lt_influence3_workspace_bytes(b, n_probe, n_obs);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 256))
        return lt_set_error(LT_ERR_WORKSPACE, "lt_influence3_rows: workspace needs %zu bytes, 256-byte aligned", need);
    hipStream_t st = (hipStream_t)stream;
    const lt_graph *g = b->g;
    const infl3_ws w = carve3(workspace, b, n_probe, n_obs);
    const int n = b->n, C = b->C, Hp1 = b->Hp1, Hp2 = b->Hp2, cp = lt_cp_for(C);
    {   // node ids: both lists checked into the workspace (the first reader is a GEMM that gathers X[probes])
        const long tot = (long)n_probe + n_obs;
        hipLaunchKernelGGL(k_check_nodes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, probe_nodes, n_probe, observe_nodes,
                           n_obs, n, w.probes_s, w.obs_s, lt_node_err_dev());
        LT_CHECK_LAUNCH();
        probe_nodes = w.probes_s;
        observe_nodes = w.obs_s;
    }
    const int lpr1 = lt_lpr_for(Hp1), lpr2 = lt_lpr_for(Hp2);
    const int words = (n + 31) / 32;
    const long maxc = g->max_col_nnz > 0 ? g->max_col_nnz : 1;
    {   // the baseline the mode reads: the fp64 pre-activations of the first two layers, or the fp32 forward
        const int rc = exact ? ensure3_fp64(b, st) : ensure3_fp32(b, st);
        if (rc) return rc;
    }
    for (int p0 = 0; p0 < n_probe; p0 += w.chunk) {
        const int nb = (n_probe - p0) < w.chunk ? (n_probe - p0) : w.chunk;
        const int32_t *probes = probe_nodes + p0;
        float *orow = out + (int64_t)p0 * ldo;
        LT_REQUIRE(((long)nb * n_obs * LT_L2_LANES + LT_BLOCK - 1) / LT_BLOCK < 2147483647L,
                   "lt_influence3_rows: %d probes x %d observed nodes per chunk exceed the grid limit", nb, n_obs);
        const long m_bound = (long)nb * maxc;
        LT_REQUIRE(m_bound < 2147483647L, "lt_influence3_rows: %d probes x %ld rows per chunk exceed the grid limit", nb, maxc);
        int rc;
        if (!exact) {
            // perturbed rows: Sp = (X[v] + X[v] d) W1, the slicing of the baseline product          attacker.py:101-105
            if (Hp1 != b->H1) LT_HIP(hipMemsetAsync(w.Sp, 0, (size_t)nb * Hp1 * sizeof(float), st));
            rc = lt_launch_gemm_splitk(b->X, b->ldx, b->W1, b->H1, w.Sp, Hp1, nb, b->H1, b->F, probe_kslice3(b), w.slabs, st,
                                       probes, delta);
            if (rc) return rc;
        } else {
            // the probes' own product rows in fp64 (dS1[v] = d * S1[v]): read off the inner baseline's product where it holds one
            // (ensure3_fp64 formed it for every row: the 2-layer stage A reads a probe's row the same way) -- X[probes] W1 over again
            // on the f64 cores was 0.40 of the 0.87 ms this build took at twitch size (round 6); formed here on the aggregate-first route
            rc = lt_tune().gcn3_product_gather != 0 ? lt_fp64_product_rows_gather(b->l1, probes, nb, w.Spd, (long)Hp1, st) : 1;
            if (rc == 1) {
                if (Hp1 != b->H1) LT_HIP(hipMemsetAsync(w.Spd, 0, (size_t)nb * Hp1 * sizeof(double), st));
                rc = lt_launch_gemm_f64_gather(b->X, (long)b->ldx, probes, nb, b->W1, (long)b->H1, b->H1, b->F, w.Spd, (long)Hp1, st);
            }
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_item_bits, dim3(nb), dim3(256), 0, st, g->tptr, g->trow, probes, nb, words, w.bits1, w.off, (int2 *)nullptr, (uint2 *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
        LT_CHECK_LAUNCH();
        // level 1: H1x rows (sparse: recomputed with row v substituted; delta: the change dH1), then S2x = H1x W2 (M = number of
        // items, only known on the device: the GEMM runs over the chunk's upper bound nb * max column length, the tiles
        // past the count exit -- no read-back, no host synchronisation)
        if (!exact) {
            LT_DISPATCH_LPR(lpr1, hipLaunchKernelGGL((k3_rows_relu<LPR_>), dim3(LT3_GRID), dim3(LT_BLOCK), 0, st, 0, g->rowptr,
                                                     g->col, g->val, b->S1, Hp1, b->b1p, w.H1x, g->tptr, g->trow, probes, nb, w.off,
                                                     w.Sp));
        } else {
            LT_DISPATCH_LPR(lpr1, hipLaunchKernelGGL((k3d_items1<LPR_>), dim3(LT3_GRID), dim3(LT_BLOCK), 0, st, g->tptr, g->trow,
                                                     g->tval, probes, nb, w.off, w.Spd, b->l1->Z1d, Hp1, delta, w.H1x));
        }
        LT_CHECK_LAUNCH();
        if (Hp2 != b->H2) LT_HIP(hipMemsetAsync(w.S2x, 0, (size_t)m_bound * Hp2 * sizeof(float), st));
        rc = lt_launch_gemm_mdev(w.H1x, Hp1, b->W2, b->H2, w.S2x, Hp2, (int)m_bound, w.off + nb, b->H2, b->H1, st);
        if (rc) return rc;
        // level 2: R2 and its items
        LT_HIP(hipMemsetAsync(w.bits2, 0, (size_t)nb * words * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k3_mark2, dim3(LT3_GRID), dim3(LT_BLOCK), 0, st, g->tptr, g->trow, probes, nb, w.off, words, w.bits2,
                           w.n_items2);
        LT_CHECK_LAUNCH();
        hipLaunchKernelGGL(k3_list2, dim3((unsigned)nb), dim3(LT_BLOCK), 0, st, words, w.bits2, w.items2, w.n_items2);
        LT_CHECK_LAUNCH();
        const unsigned gridC = (unsigned)(((long)nb * n_obs * LT_L2_LANES + LT_BLOCK - 1) / LT_BLOCK);
        if (!exact) {
            LT_DISPATCH_LPR(lpr2, LT_DISPATCH_CP(cp,
                hipLaunchKernelGGL((k3_stageB<LPR_, CP_>), dim3(LT3_GRID), dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, b->S2,
                                   Hp2, b->b2p, b->W3p, C, w.off, w.S2x, w.bits1, words, w.items2, w.n_items2, n, w.S3x)));
            LT_CHECK_LAUNCH();
            // level 3: observed rows
            LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k3_stageC<CP_>), dim3(gridC), dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val,
                                                  b->S3, C, b->b3, b->OUT, nb, n, w.S3x, w.bits2, words, observe_nodes, n_obs,
                                                  delta, orow, (long)ldo));
        } else {
            LT_DISPATCH_LPR(lpr2, LT_DISPATCH_CP(cp,
                hipLaunchKernelGGL((k3d_stageB<LPR_, CP_>), dim3(LT3_GRID), dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, b->Z2d,
                                   Hp2, b->W3p, C, w.off, w.S2x, w.bits1, words, w.items2, w.n_items2, n, w.S3x)));
            LT_CHECK_LAUNCH();
            LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k3d_stageC<CP_>), dim3(gridC), dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val,
                                                  C, nb, n, w.S3x, w.bits2, words, observe_nodes, n_obs, delta, orow, (long)ldo));
        }
        LT_CHECK_LAUNCH();
    }
    return LT_OK;
}
