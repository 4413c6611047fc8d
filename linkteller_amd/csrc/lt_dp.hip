// LapGraph cell selection on the device (reference worker.py:302-335).
//
// The reference adds an N x N Laplace noise matrix (strict lower triangle) to the adjacency and keeps the n_keep largest
// cells, with a 50-way np.argpartition over the flattened float64 matrix.  The noise itself has to stay numpy's (a given
// --noise-seed must give the reference's graph), so the host draws it and uploads it; everything after the draw runs here:
//   k_lap_add_edges   cell(i, j) += 1.0 for the edges j < i            (the fp64 add of worker.py:299, same rounding)
//   k_lap_hist        one radix pass of a top-k SELECT over the order-preserving 64-bit keys of the cells j < i: histogram
//                     of the next 8 bits among the cells whose higher bits match the prefix found so far
//   k_lap_pick        walks the 256 bins from the top and fixes the next 8 bits of the threshold (no host round trip)
//   k_lap_collect     cells above the threshold, then as many cells EQUAL to it as are still missing
// The selected SET is what np.argpartition returns whenever the n_keep-th largest value is not tied (continuous noise:
// ties occur with probability zero; cells of the upper triangle are exact zeros in the reference and are never reached
// because n_keep is far below the number of positive cells -- a threshold <= 0 is refused, the reference asserts there).
#include <string.h>

#include "lt_internal.h"

// order-preserving map double -> uint64 (larger double <=> larger key)
__device__ __forceinline__ unsigned long long lap_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void k_lap_add_edges(int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                double *__restrict__ cells) {
    const int i = blockIdx.x;
    for (int e = rowptr[i] + threadIdx.x; e < rowptr[i + 1]; e += blockDim.x) {
        const int j = col[e];
        if (j < i) cells[(size_t)i * n + j] += 1.0;
    }
}

// state[0] = prefix (the bits of the threshold key fixed so far, in place), state[1] = cells still to take inside the prefix
// bucket, state[2] = output cursor.  `shift` = position of the digit this pass histograms (56, 48, ..., 0).
__global__ __launch_bounds__(256) void k_lap_hist(int n, const double *__restrict__ cells, int shift,
                                                  const unsigned long long *__restrict__ state,
                                                  unsigned *__restrict__ hist) {
    __shared__ unsigned sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = state[0];
    const unsigned long long himask = shift == 56 ? 0ull : ~0ull << (shift + 8);
    for (int i = blockIdx.x + 1; i < n; i += gridDim.x) {          // row i holds the cells j < i
        const double *row = cells + (size_t)i * n;
        for (int j = threadIdx.x; j < i; j += 256) {
            const unsigned long long k = lap_key(row[j]);
            if ((k & himask) == (prefix & himask)) atomicAdd(&sh[(unsigned)(k >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void k_lap_pick(int shift, unsigned long long *__restrict__ state, unsigned *__restrict__ hist) {
    // single thread: 256 bins
    unsigned long long need = state[1];
    int d = 255;
    for (; d > 0; --d) {
        const unsigned c = hist[d];
        if (c >= need) break;
        need -= c;
    }
    state[0] |= (unsigned long long)d << shift;
    state[1] = need;                       // cells to take among those whose key matches the prefix so far
    for (int i = 0; i < 256; ++i) hist[i] = 0;
}

__global__ __launch_bounds__(256) void k_lap_collect(int n, const double *__restrict__ cells,
                                                     unsigned long long *__restrict__ state, long long k_total,
                                                     long long *__restrict__ out, int pass) {
    const unsigned long long thr = state[0];
    for (int i = blockIdx.x + 1; i < n; i += gridDim.x) {
        const double *row = cells + (size_t)i * n;
        for (int j = threadIdx.x; j < i; j += 256) {
            const unsigned long long k = lap_key(row[j]);
            if (pass == 0 ? k > thr : k == thr) {
                const unsigned long long p = atomicAdd(&state[2], 1ull);
                if ((long long)p < k_total) out[p] = (long long)i * n + j;
            }
        }
    }
}

// cells: [n, n] float64 on the device, holding the noise (only j < i is read); overwritten with adjacency + noise.
// lower_rowptr / lower_col: device CSR of the adjacency (any entries with j >= i are ignored).  out_idx: [n_keep] int64
// on the device, the flat indices i * n + j of the selected cells in no particular order.  work: >= 4096 bytes of
// device scratch.  threshold_out (host, optional): the n_keep-th largest cell value.  Synchronises (it returns a value).
extern "C" int lt_lapgraph_select(int32_t n, const int32_t *lower_rowptr, const int32_t *lower_col, double *cells,
                                  int64_t n_keep, int64_t *out_idx, void *work, size_t work_bytes,
                                  double *threshold_out, void *stream) {
    LT_REQUIRE(n > 1 && lower_rowptr && lower_col && cells && out_idx && work, "lt_lapgraph_select: NULL argument or n < 2");
    LT_REQUIRE(work_bytes >= 4096 && ((uintptr_t)work % 8) == 0, "lt_lapgraph_select: work needs 4096 bytes, 8-byte aligned");
    const long long total = (long long)n * (n - 1) / 2;
    LT_REQUIRE(n_keep > 0 && n_keep <= total, "lt_lapgraph_select: n_keep=%lld outside [1, %lld]", (long long)n_keep, total);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *state = (unsigned long long *)work;          // [0] prefix, [1] remaining, [2] cursor
    unsigned *hist = (unsigned *)((char *)work + 64);                // 256 bins
    LT_HIP(hipMemsetAsync(work, 0, 4096, st));
    const unsigned long long init[3] = {0ull, (unsigned long long)n_keep, 0ull};
    LT_HIP(hipMemcpyAsync(state, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_lap_add_edges, dim3((unsigned)n), dim3(64), 0, st, n, lower_rowptr, lower_col, cells);
    LT_CHECK_LAUNCH();
    const unsigned grid = (unsigned)(n - 1 < 4096 ? n - 1 : 4096);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_lap_hist, dim3(grid), dim3(256), 0, st, n, cells, shift, state, hist);
        hipLaunchKernelGGL(k_lap_pick, dim3(1), dim3(1), 0, st, shift, state, hist);
        LT_CHECK_LAUNCH();
    }
    // state[0] = key of the n_keep-th largest cell, state[1] = how many cells equal to it belong to the selection
    hipLaunchKernelGGL(k_lap_collect, dim3(grid), dim3(256), 0, st, n, cells, state, (long long)n_keep, (long long *)out_idx, 0);
    hipLaunchKernelGGL(k_lap_collect, dim3(grid), dim3(256), 0, st, n, cells, state, (long long)n_keep, (long long *)out_idx, 1);
    LT_CHECK_LAUNCH();
    unsigned long long fin[3];
    LT_HIP(hipMemcpyAsync(fin, state, sizeof(fin), hipMemcpyDeviceToHost, st));
    LT_HIP(hipStreamSynchronize(st));
    unsigned long long u = fin[0];
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    double thr;
    memcpy(&thr, &u, sizeof(thr));
    if (threshold_out) *threshold_out = thr;
    if (!(thr > 0.0))
        return lt_set_error(LT_ERR_UNSUPPORTED, "lt_lapgraph_select: the %lld-th largest cell is %g <= 0: the selection would reach "
                                                "the zero cells of the upper triangle (the reference asserts there, worker.py:326)",
                            (long long)n_keep, thr);
    if ((long long)fin[2] < n_keep)
        return lt_set_error(LT_ERR_INVALID, "lt_lapgraph_select: collected %llu of %lld cells", fin[2], (long long)n_keep);
    return LT_OK;
}
