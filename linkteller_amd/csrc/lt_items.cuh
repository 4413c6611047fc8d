// Item-list helpers shared by the SPARSE / DELTA probe kernels (lt_influence.hip) and the 3-layer path (lt_gcn3.hip):
// per-probe offsets into item lists, the membership bitmap with positions, the finite-difference tail.
#pragma once
#include "lt_rows.cuh"

__device__ __forceinline__ int find_probe(const int32_t *__restrict__ off, int nb, int item) {
    int lo = 0, hi = nb;  // off[lo] <= item < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}

// position of `c` in the ascending list rows[0..cnt), or -1
__device__ __forceinline__ int find_row(const int32_t *__restrict__ rows, int cnt, int c) {
    int lo = 0, hi = cnt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = rows[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid;
    }
    return -1;
}


// bits[b][r >> 5] = { mask, base }: bit (r & 31) of mask = 1  <=>  r in R_v of probe b, and base = the position in
// R_v (the ascending CSC list of column v) of the lowest member of this word, so that ONE 8-byte load tells stage B
// both whether an entry of an observed row is affected by the probe and which item replaces it:
//     position(r) = base + popcount(mask & ((1 << (r & 31)) - 1))
// (without the bitmap -- huge graphs -- both questions are a binary search in R_v).  One block per probe.
static __global__ __launch_bounds__(256) void k_item_bits(const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow,
                                                   const int32_t *__restrict__ probes, int nb, int words,
                                                   uint2 *__restrict__ bits, int32_t *__restrict__ off,
                                                   int2 *__restrict__ item_pr) {
    // One block per probe.  It also forms the probe's item offset off[b] = sum of |R_v| over the probes before it
    // (every block sums its own prefix: nb^2 / 2 four-byte loads in all, no scan kernel in front), the last block
    // writes the total off[nb].  bits == NULL: no bitmap (huge graphs); item_pr == NULL: no (probe, row) table.
    __shared__ int32_t red[4];
    const int b = blockIdx.x;
    int part = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
        const int vi = probes[i];
        part += tptr[vi + 1] - tptr[vi];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    uint2 *mine = bits ? bits + (size_t)b * words : nullptr;
    if (mine)
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = make_uint2(0u, 0xffffffffu);
    __syncthreads();
    const int my_off = red[0] + red[1] + red[2] + red[3];
    const int v = probes[b];
    const int t0 = tptr[v], t1 = tptr[v + 1];
    if (threadIdx.x == 0) {
        off[b] = my_off;
        if (b == nb - 1) off[nb] = my_off + (t1 - t0);
    }
    int2 *items = item_pr ? item_pr + my_off : nullptr;
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int r = trow[t];
        if (mine) {
            atomicOr(&mine[r >> 5].x, 1u << (r & 31));
            atomicMin(&mine[r >> 5].y, (unsigned)(t - t0));
        }
        // item (off[b] + position in R_v) = (probe index, row): stage A reads it instead of searching `off`
        if (items) items[t - t0] = make_int2(b, r);
    }
}
// position of column c in R_v from the probe's bitmap row, or -1
__device__ __forceinline__ int bits_pos(const uint2 *__restrict__ mb, int c) {
    const uint2 w = mb[c >> 5];
    const unsigned bit = 1u << (c & 31);
    return (w.x & bit) ? (int)(w.y + __popc(w.x & (bit - 1u))) : -1;
}

// ------------------------------------------------------------------------------------------------
// shared tail: finite difference + L2 norm of one observed row          attacker.py:105-106,227-229
// ------------------------------------------------------------------------------------------------
template <int CP>
__device__ __forceinline__ float diff_norm(const float (&acc)[CP], const float *__restrict__ b2,
                                           const float *__restrict__ base, int C, float delta) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) {
            const float o = acc[c] + b2[c];           // layers.py:34
            const float d = (o - base[c]) / delta;    // attacker.py:105-106
            ss = fmaf(d, d, ss);
        }
    return sqrtf(ss);
}

