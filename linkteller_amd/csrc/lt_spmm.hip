// SpMM  out = A_hat * S (+ bias)(ReLU)                       torch.spmm + bias, gcn/layers.py:32-36
//
// Two routes, same arithmetic (a row is a k-ordered fmaf chain per output column; a row of more than
// LT_ROW_SEG entries is the ordered sum of its 128-entry segment chains -- lt_rows.hip.h):
//
//  * small graphs (S fits the caches: twitch): k_spmm_rows, one lane group per row, the whole row of S per gather;
//    launch-bound, nothing to tile.
//  * large graphs (S beyond the L2s): k_rows_tiled.  What bounds a row-gather SpMM there is not HBM streaming but
//    the L2 capacity misses of the gathers (R-MAT scale 21: 67 GB of gathered rows against 5 GB of algorithmic
//    traffic), served by the fabric at ~5 TB/s.  Three things cut those misses (profiles/r02_spmm_lab_*.txt):
//      - column slices: a 16-lane group owns one work item x 64 columns (256 B of a gathered row) and the slice is
//        a function of the XCD a block lands on (blocks are dealt round-robin over the 8 XCDs), so one XCD's
//        4 MiB L2 only ever sees a quarter of S: four times the rows per byte of cache;
//      - work items instead of rows: rows of up to 128 entries and the 128-entry segments of the long rows are
//        the same kind of item, sorted so that a wave's four items are equally long (no idle lane groups) and the
//        hub rows are spread over the chip instead of serialising their waves;
//      - the segment items are ordered by the column their first entry reads.  Entries of a row are sorted by
//        column, so waves that run at the same time gather from one sliding window of S and the segments of
//        different hub rows that cross the same columns meet in L2.
//    (col, val) stream in with non-temporal loads, one coalesced 16-entry block per lane group, and are handed
//    round by DPP row broadcasts; results leave with non-temporal stores.
#include <stdlib.h>
#include <type_traits>

#include "lt_rows.hip.h"

#define LT_BLOCK 256
#define LT_TILE_GL 16    // lanes per item: 16 lanes x float4 = 64 columns
#define LT_TILE_U 8      // gathers in flight per lane

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
}
// lane K of this lane's 16-lane row (DPP row_newbcast)
template <int K>
__device__ __forceinline__ int row_bcast(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + K, 0xf, 0xf, false);
}

// BIG: S spans 4 GiB or more (byte offsets of a gathered row no longer fit 32 bits)
// (col, val) of entry e: from the interleaved stream lt_graph::cv when the graph has one -- ONE 8-byte request per entry instead
// of two of 4: the value stream beside the column stream cost k_rows_tiled 0.5 of 8.8 ms on the R-MAT graph of BASELINE configs[4],
// as instructions and requests, not as bytes -- else from the two arrays
__device__ __forceinline__ void tiled_entry(const int2 *__restrict__ cv, const int32_t *__restrict__ col, const float *__restrict__ val,
                                            int e, int &c, float &a) {
    if (cv) {
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        const i32x2 p_ = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(cv) + e);
        c = p_.x;
        a = __int_as_float(p_.y);
    } else {
        c = __builtin_nontemporal_load(col + e);
        a = __builtin_nontemporal_load(val + e);
    }
}

template <bool BIG>
__global__ __launch_bounds__(LT_BLOCK) void k_rows_tiled(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt,
    const int32_t *__restrict__ w_dst, int n, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S, long lds, int ncols, const float *__restrict__ init,
    const float *__restrict__ bias_after, int relu, float *__restrict__ out, long ldo,
    float *__restrict__ seg_out, long ld_seg, const int32_t *__restrict__ seg_begin,
    const int32_t *__restrict__ seg_long, const int32_t *__restrict__ long_row,
    const int32_t *__restrict__ rowptr, int ns, const int2 *__restrict__ cv) {
    constexpr int GL = LT_TILE_GL, U = LT_TILE_U;
    constexpr int GPW = 64 / GL, IPB = (LT_BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int j = lane & (GL - 1);
    // slice from the XCD: blocks b and b + 8 share an XCD (observed dispatch order; only speed depends on it)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int xps = 8 / ns;
    const int slice = xcd % ns;
    const int it = (q * xps + xcd / ns) * IPB + (threadIdx.x >> 6) * GPW + lane / GL;
    if (it >= n_items) return;
    const int e0 = __builtin_nontemporal_load(w_e0 + it);
    const int cnt = __builtin_nontemporal_load(w_cnt + it);
    const int dst = __builtin_nontemporal_load(w_dst + it);
    const int coff = slice * 4 * GL + 4 * j;
    const bool active = coff < ncols;
    // a chain starts from `init` (the layer-1 bias) on a short row and on the FIRST segment of a long row
    bool first = true;
    if (init != nullptr && dst >= n) {      // (only a chain that starts from `init` asks: four dependent loads per segment otherwise for nothing)
        const int sg = dst - n;
        first = seg_begin[sg] == rowptr[long_row[seg_long[sg]]];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (init != nullptr && first && active) acc = ld4(init + coff);
    const char *Sb = reinterpret_cast<const char *>(S);
    const size_t rowbytes = (size_t)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const int e1 = e0 + cnt;
    // (col, val) of the NEXT 16-entry block are requested before the gathers of the current one go out
    int nxc = 0;
    float nxa = 0.f;
    if (e0 + j < e1) tiled_entry(cv, col, val, e0 + j, nxc, nxa);
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        const float mya = nxa;
        nxc = 0;
        nxa = 0.f;
        if (me + GL < e1) tiled_entry(cv, col, val, me + GL, nxc, nxa);
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                f32x4 s[U];
                float a[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = row_bcast<k>(myc);
                    a[u] = __builtin_bit_cast(float, row_bcast<k>(__builtin_bit_cast(int, mya)));
                    s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (k < left && active) {
                        if (BIG) s[u] = *reinterpret_cast<const f32x4 *>(Sb + ((size_t)c * rowbytes + loff));
                        else s[u] = *reinterpret_cast<const f32x4 *>(Sb + (size_t)((unsigned)c * (unsigned)rowbytes + loff));
                    }
                });
                // entries past the end of the item are skipped, not multiplied by zero: the chain is exactly row_dot's
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    if (kb + u < left) acc = fma4(a[u], s[u], acc);
                });
            }
        });
    }
    if (!active) return;
    float *d;
    if (dst < n) {
        if (bias_after) {
            const f32x4 b = ld4(bias_after + coff);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        if (relu) {
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
        d = out + (size_t)dst * ldo + coff;
    } else {
        d = seg_out + (size_t)(dst - n) * ld_seg + coff;
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(d));
}

// fp64 twin of k_rows_tiled for the pre-activation of LT_MODE_DELTA (lt_fp64.hip): S and the result are double, the
// values stay the graph's floats.  A lane holds 2 columns (16 bytes, as in the f32 kernel), so a 16-lane group covers a
// 32-column slice -- 256 bytes of a gathered row again -- and 256 columns make 8 slices: one per XCD.  Chains start
// from zero and the bias is added after (k_spmm_f64's order); segment sums go raw to seg_out for k_spmm_f64_long.
typedef double f64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(LT_BLOCK) void k_rows_tiled_f64(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt,
    const int32_t *__restrict__ w_dst, int n, const int32_t *__restrict__ col, const float *__restrict__ val,
    const double *__restrict__ S, long lds, int ncols, const float *__restrict__ bias_after,
    double *__restrict__ out, long ldo, double *__restrict__ seg_out, long ld_seg, int ns, const int2 *__restrict__ cv) {
    constexpr int GL = LT_TILE_GL, U = LT_TILE_U;
    constexpr int GPW = 64 / GL, IPB = (LT_BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int j = lane & (GL - 1);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int xps = 8 / ns;
    const int slice = xcd % ns;
    const int it = (q * xps + xcd / ns) * IPB + (threadIdx.x >> 6) * GPW + lane / GL;
    if (it >= n_items) return;
    const int e0 = __builtin_nontemporal_load(w_e0 + it);
    const int cnt = __builtin_nontemporal_load(w_cnt + it);
    const int dst = __builtin_nontemporal_load(w_dst + it);
    const int coff = slice * 2 * GL + 2 * j;
    const bool active = coff < ncols;
    f64x2 acc = {0.0, 0.0};
    const char *Sb = reinterpret_cast<const char *>(S);
    const size_t rowbytes = (size_t)lds * 8u;
    const size_t loff = (size_t)coff * 8u;
    const int e1 = e0 + cnt;
    int nxc = 0;
    float nxa = 0.f;
    if (e0 + j < e1) tiled_entry(cv, col, val, e0 + j, nxc, nxa);
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        const float mya = nxa;
        nxc = 0;
        nxa = 0.f;
        if (me + GL < e1) tiled_entry(cv, col, val, me + GL, nxc, nxa);
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                f64x2 s[U];
                double a[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = row_bcast<k>(myc);
                    a[u] = (double)__builtin_bit_cast(float, row_bcast<k>(__builtin_bit_cast(int, mya)));
                    s[u] = f64x2{0.0, 0.0};
                    if (k < left && active) s[u] = *reinterpret_cast<const f64x2 *>(Sb + ((size_t)c * rowbytes + loff));
                });
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    if (kb + u < left) {
                        acc.x = fma(a[u], s[u].x, acc.x);
                        acc.y = fma(a[u], s[u].y, acc.y);
                    }
                });
            }
        });
    }
    if (!active) return;
    double *d;
    if (dst < n) {
        acc.x += (double)bias_after[coff];
        acc.y += (double)bias_after[coff + 1];
        d = out + (size_t)dst * ldo + coff;
    } else {
        d = seg_out + (size_t)(dst - n) * ld_seg + coff;
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<f64x2 *>(d));
}

// The same work-item kernel for the AGGREGATE-FIRST route of the fp64 pre-activation (lt_fp64.hip): the gathered matrix
// is the fp32 FEATURE matrix X (4 columns = 16 bytes per lane, 64-column slices), the chains accumulate in fp64 and only
// the items of rows marked `state[row] == 2` (rows a probe of this call reaches and that hold no valid pre-activation
// yet) run.  Y[row, :] = sum_e val[e] * X[col[e], :], chains from zero, no bias.
// Round 5: the marked items are COMPACTED first (k_z_flags / k_z_scan / k_z_scatter: their indices into the work-item list, in list order block by
// block, the count on the device) and this kernel walks that list with a fixed grid.  Before, every item of the graph got a
// lane group that looked its row up and returned -- at BASELINE configs[4] 2.5 M items x 4 slices for 190 K marked ones, and the
// marked ones sat one or two to a wave with the other lane groups idle: the gathers of a 512-probe call ran at 11 TB/s of a
// possible ~17 (L2-resident hub columns).
typedef double f64x4s __attribute__((ext_vector_type(4)));
// (the list keeps the ORDER of the work-item list -- segments by first column: waves that run together gather from one sliding
// window of X -- so it is built in three small steps, flags + block counts / scan of the counts / scatter, and not with one atomic
// cursor: blocks claim a cursor in completion order, which shuffled the list across ~ 500 K items and cost the gathers a fifth)
__global__ __launch_bounds__(256) void k_z_flags(int n_items, const int32_t *__restrict__ w_dst, int n,
                                                 const int32_t *__restrict__ seg_long, const int32_t *__restrict__ long_row,
                                                 const int32_t *__restrict__ state, unsigned long long *__restrict__ masks,
                                                 int32_t *__restrict__ bcnt) {
    __shared__ int32_t s_cnt[4];
    const int it = blockIdx.x * 256 + threadIdx.x;
    bool on = false;
    if (it < n_items) {
        const int dst = __builtin_nontemporal_load(w_dst + it);
        const int row = dst < n ? dst : long_row[seg_long[dst - n]];
        on = state[row] == 2;
    }
    const unsigned long long mk = __ballot(on);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { masks[(size_t)blockIdx.x * 4 + wid] = mk; s_cnt[wid] = __popcll(mk); }
    __syncthreads();
    if (threadIdx.x == 0) bcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
// exclusive scan of the block counts (one block walks them in tiles of 1024), the total -> *zicount
__global__ __launch_bounds__(1024) void k_z_scan(int nb, int32_t *__restrict__ bcnt, int32_t *__restrict__ zicount) {
    __shared__ int32_t s_w[16];
    __shared__ int32_t s_carry;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nb; t0 += 1024) {
        const int i = t0 + threadIdx.x;
        const int v = i < nb ? bcnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int t = __shfl_up(incl, m, 64);
            if (lane >= m) incl += t;
        }
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wid; ++w) before += s_w[w];
        if (i < nb) bcnt[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *zicount = s_carry;
}
__global__ __launch_bounds__(256) void k_z_scatter(int n_items, const unsigned long long *__restrict__ masks,
                                                   const int32_t *__restrict__ boff, int32_t *__restrict__ zitems) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long *mb = masks + (size_t)blockIdx.x * 4;
    const unsigned long long mk = mb[wid];
    if (!((mk >> lane) & 1ull) || it >= n_items) return;
    int before = __popcll(mk & ((1ull << lane) - 1ull));
    for (int w = 0; w < wid; ++w) before += __popcll(mb[w]);
    zitems[boff[blockIdx.x] + before] = it;
}
__global__ __launch_bounds__(LT_BLOCK) void k_rows_tiled_xf64(
    const int32_t *__restrict__ zitems, const int32_t *__restrict__ zicount, const int32_t *__restrict__ w_e0,
    const int32_t *__restrict__ w_cnt, const int32_t *__restrict__ w_dst, int n, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ X, long ldx, int ncols, double *__restrict__ out, long ldo,
    double *__restrict__ seg_out, long ld_seg, int ns, const int2 *__restrict__ cv) {
    constexpr int GL = LT_TILE_GL, U = LT_TILE_U;
    constexpr int GPW = 64 / GL, IPB = (LT_BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int j = lane & (GL - 1);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;
    const int xps = 8 / ns;
    const int slice = xcd % ns;
    const int total = *zicount;
    const int coff = slice * 4 * GL + 4 * j;
    const bool active = coff < ncols;
    const bool vec = coff + 3 < ncols && (ldx & 3) == 0;      // 16-byte gathers when the rows of X allow them
    for (long chunk = (long)q * xps + xcd / ns; chunk * IPB < total; chunk += (long)nq * xps) {
        const int k = (int)chunk * IPB + (threadIdx.x >> 6) * GPW + lane / GL;
        if (k >= total) continue;
        const int it = zitems[k];
        const int dst = __builtin_nontemporal_load(w_dst + it);
        const int e0 = __builtin_nontemporal_load(w_e0 + it);
        const int cnt = __builtin_nontemporal_load(w_cnt + it);
        f64x4s acc = {0.0, 0.0, 0.0, 0.0};
        const int e1 = e0 + cnt;
        int nxc = 0;
        float nxa = 0.f;
        if (e0 + j < e1) tiled_entry(cv, col, val, e0 + j, nxc, nxa);
        for (int eb = e0; eb < e1; eb += GL) {
            const int me = eb + j;
            const int myc = nxc;
            const float mya = nxa;
            nxc = 0;
            nxa = 0.f;
            if (me + GL < e1) tiled_entry(cv, col, val, me + GL, nxc, nxa);
            const int left = e1 - eb;
            static_for<GL / U>([&](auto kbt) {
                constexpr int kb = decltype(kbt)::value * U;
                if (kb < left) {
                    f32x4 s[U];
                    double a[U];
                    static_for<U>([&](auto ut) {
                        constexpr int u = decltype(ut)::value;
                        constexpr int kk = kb + u;
                        const int c = row_bcast<kk>(myc);
                        a[u] = (double)__builtin_bit_cast(float, row_bcast<kk>(__builtin_bit_cast(int, mya)));
                        s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (kk < left && active) {
                            const float *p = X + ((size_t)c * (size_t)ldx + coff);
                            if (vec) s[u] = *reinterpret_cast<const f32x4 *>(p);
                            else
#pragma unroll
                                for (int t = 0; t < 4; ++t) if (coff + t < ncols) s[u][t] = p[t];
                        }
                    });
                    static_for<U>([&](auto ut) {
                        constexpr int u = decltype(ut)::value;
                        if (kb + u < left) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[t] = fma(a[u], (double)s[u][t], acc[t]);
                        }
                    });
                }
            });
        }
        if (!active) continue;
        double *d = dst < n ? out + (size_t)dst * ldo + coff : seg_out + (size_t)(dst - n) * ld_seg + coff;
        __builtin_nontemporal_store(acc, reinterpret_cast<f64x4s *>(d));     // (ldo / ld_seg are multiples of 4: 32-byte aligned)
    }
}

// zitems: scratch of lt_xf64_scratch_words(g) int32 words for the compacted list and its construction, zicount: its device counter
size_t lt_xf64_scratch_words(const lt_graph *g) {
    const size_t nbz = ((size_t)(g->w_n > 0 ? g->w_n : 1) + 255) / 256;
    return (size_t)lt_round_up(g->w_n > 0 ? g->w_n : 1, 4) + nbz * 8 + nbz + 16;      // list | 4 x 8-byte masks per block | block counts
}
int lt_launch_rows_tiled_xf64(const lt_graph *g, const float *X, int64_t ldx, int ncols, double *out, int64_t ldo,
                              double *seg_out, int64_t ld_seg, const int32_t *state, int32_t *zitems, int32_t *zicount,
                              hipStream_t st) {
    if (g->w_n == 0) return LT_OK;
    int ns = (ncols + 4 * LT_TILE_GL - 1) / (4 * LT_TILE_GL);   // 64-column slices: 1, 2, 4, 8
    ns = ns <= 1 ? 1 : (ns == 2 ? 2 : (ns <= 4 ? 4 : 8));
    LT_REQUIRE(ncols <= 4 * LT_TILE_GL * 8 && ldo % 4 == 0 && ld_seg % 4 == 0, "tiled aggregate-first SpMM: ncols=%d", ncols);
    // zitems doubles as the scratch of the compaction: [0, w_n) the list, behind it the flag masks and the block counts
    const int nbz = (g->w_n + 255) / 256;
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(zitems + lt_round_up(g->w_n, 4));
    int32_t *bcnt = reinterpret_cast<int32_t *>(masks + (size_t)nbz * 4);
    hipLaunchKernelGGL(k_z_flags, dim3((unsigned)nbz), dim3(256), 0, st, g->w_n, g->w_dst, g->n, g->p_seg_long, g->p_long_row, state,
                       masks, bcnt);
    hipLaunchKernelGGL(k_z_scan, dim3(1), dim3(1024), 0, st, nbz, bcnt, zicount);
    hipLaunchKernelGGL(k_z_scatter, dim3((unsigned)nbz), dim3(256), 0, st, g->w_n, masks, bcnt, zitems);
    LT_CHECK_LAUNCH();
    // a fixed grid walks the list (its length stays on the device): as many blocks as the chip holds at once, or the graph needs
    const int xps = 8 / ns;
    constexpr int IPB = (LT_BLOCK / 64) * (64 / LT_TILE_GL);
    const long chunks = ((long)g->w_n + IPB - 1) / IPB;
    long grid = 8 * ((chunks + xps - 1) / xps);
    // (how many blocks walk the list at once matters -- more gathers in flight than the L2s hold windows for cost more than idle
    // CUs do.  BASELINE configs[4], 190 K marked items, per call incl. the rest of the baseline and the 0.76 ms of the probes:
    // 32 blocks per XCD 4.43 ms, 48: 3.66, 64: 3.07, 96: 3.00, 128: 3.18, 512: 3.33; profiles/r05_xf64_sweep.txt)
    const long cap = lt_tune().xf64_blocks;           // ("xf64_blocks", LT_XF64_BLOCKS: 96 per XCD)
    if (grid > 8 * cap) grid = 8 * cap;
    hipLaunchKernelGGL(k_rows_tiled_xf64, dim3((unsigned)grid), dim3(LT_BLOCK), 0, st, zitems, zicount, g->w_e0, g->w_cnt, g->w_dst,
                       g->n, g->col, g->val, X, (long)ldx, ncols, out, (long)ldo, seg_out, (long)ld_seg, ns, g->cv);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

int lt_launch_rows_tiled_f64(const lt_graph *g, const double *S, int64_t lds, int ncols, const float *bias_after,
                             double *out, int64_t ldo, double *seg_out, int64_t ld_seg, hipStream_t st) {
    if (g->w_n == 0) return LT_OK;
    int ns = (ncols + 2 * LT_TILE_GL - 1) / (2 * LT_TILE_GL);   // 32-column slices: 1, 2, 4, 8
    ns = ns <= 1 ? 1 : (ns == 2 ? 2 : (ns <= 4 ? 4 : 8));
    LT_REQUIRE(ncols % 2 == 0 && ncols <= 2 * LT_TILE_GL * 8, "tiled fp64 SpMM: ncols=%d", ncols);
    const int xps = 8 / ns;
    constexpr int IPB = (LT_BLOCK / 64) * (64 / LT_TILE_GL);
    const long chunks = ((long)g->w_n + IPB - 1) / IPB;
    const long grid = 8 * ((chunks + xps - 1) / xps);
    LT_REQUIRE(grid < 2147483647L, "tiled fp64 SpMM: grid limit");
    hipLaunchKernelGGL(k_rows_tiled_f64, dim3((unsigned)grid), dim3(LT_BLOCK), 0, st, g->w_n, g->w_e0, g->w_cnt, g->w_dst,
                       g->n, g->col, g->val, S, (long)lds, ncols, bias_after, out, (long)ldo, seg_out, (long)ld_seg, ns, g->cv);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// The tiled route pays when the gathers MISS and the misses are SKEWED: S beyond the L2s, a natural order without
// locality, hot columns that a quarter-of-S-per-XCD cache can keep (R-MAT, power laws).  Measured on 2 M-node graphs of
// average degree 33 (profiles/r02_spmm_lab_banded.txt): neighbours within +-4096 of the row index -- row kernel 3.5 ms
// (whole-row 1 KiB gathers hit L2 and move at the L2 gather rate, 20 TB/s), tiled 5.5; uniformly random neighbours,
// no hubs -- row kernel 11.2 ms, tiled 12.4 (nothing to keep in L2, and 256-byte pieces cost more per byte);
// R-MAT -- row kernel 13.1 ms, tiled 7.6.  tiled_min_bytes <= 0 forces the tiled route (tests).
bool lt_tiled_wanted(const lt_graph *g, int ncols) {
    if (g->w_n == 0 || ncols % 4 != 0) return false;
    const long long thr = lt_tune().tiled_min_bytes;
    if (thr <= 0) return true;
    return (long long)g->n * ncols * (long long)sizeof(float) >= thr && g->local_frac < 0.5f && g->hot_frac >= 0.2f;
}

int lt_launch_rows_tiled(const lt_graph *g, const float *S, int64_t lds, int ncols, const float *init,
                         const float *bias_after, int relu, float *out, int64_t ldo, float *seg_out,
                         int64_t ld_seg, hipStream_t st) {
    if (g->w_n == 0) return LT_OK;
    int ns = (ncols + 4 * LT_TILE_GL - 1) / (4 * LT_TILE_GL);   // 64-column slices: 1, 2, 4 (3 -> 4, one idle)
    ns = ns <= 1 ? 1 : (ns == 2 ? 2 : 4);
    LT_REQUIRE(ncols <= 4 * LT_TILE_GL * 4, "tiled SpMM: ncols=%d > %d", ncols, 16 * LT_TILE_GL);
    const int xps = 8 / ns;
    constexpr int IPB = (LT_BLOCK / 64) * (64 / LT_TILE_GL);
    const long chunks = ((long)g->w_n + IPB - 1) / IPB;
    const long grid = 8 * ((chunks + xps - 1) / xps);
    LT_REQUIRE(grid < 2147483647L, "tiled SpMM: grid limit");
    const bool big = lt_tune().tiled_big != 0 || (unsigned long long)g->n * (unsigned long long)lds * 4ull >= (1ull << 32);
    // (round 5 capped the blocks a CU holds with unused dynamic LDS -- "do fewer gathers in flight help here as they do the
    // aggregate-first gathers?" -- 6 / 5 / 4 / 3 / 2 blocks per CU: 8.41 / 8.59 / 8.59 / 8.66 / 9.66 ms against 8.36: no, and the
    // hook left the production launch in round 6; profiles/r05_tiled_occupancy.txt)
    const unsigned occ_lds = 0u;
    if (big)
        hipLaunchKernelGGL(k_rows_tiled<true>, dim3((unsigned)grid), dim3(LT_BLOCK), occ_lds, st, g->w_n, g->w_e0, g->w_cnt, g->w_dst,
                           g->n, g->col, g->val, S, (long)lds, ncols, init, bias_after, relu, out, (long)ldo, seg_out,
                           (long)ld_seg, g->p_seg_begin, g->p_seg_long, g->p_long_row, g->rowptr, ns, g->cv);
    else
        hipLaunchKernelGGL(k_rows_tiled<false>, dim3((unsigned)grid), dim3(LT_BLOCK), occ_lds, st, g->w_n, g->w_e0, g->w_cnt, g->w_dst,
                           g->n, g->col, g->val, S, (long)lds, ncols, init, bias_after, relu, out, (long)ldo, seg_out,
                           (long)ld_seg, g->p_seg_begin, g->p_seg_long, g->p_long_row, g->rowptr, ns, g->cv);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// --------------------------------------------------------------------------------------------
// Measurement support (bench.py `roofline_spmm.gather_ceiling`): k_rows_tiled with everything but its GATHERS removed --
// the same work items in the same order, the same column stream (col only), the same slice = f(XCD) placement and the
// same 16-lane x 16-byte pieces, U gathers in flight per lane; no val stream, no fmaf chains (the loaded words are
// XOR-folded, one VALU op each, so that the loads stay live), no result rows (one word per item and slice).  Its duration
// is what the memory system needs to deliver THIS index stream at THIS hit distribution: a lower bound for any
// row-gather SpMM that issues these gathers, whatever its arithmetic (tools/spmm_lab holds the variants that were tried:
// 8 / 16 / 32 / 64 lanes per item, 8 or 16 gathers in flight).
// --------------------------------------------------------------------------------------------
template <int U, bool BIG>
__global__ __launch_bounds__(LT_BLOCK) void k_rows_tiled_gathers_only(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt, const int32_t *__restrict__ col,
    const float *__restrict__ S, long lds, int ncols, unsigned *__restrict__ sink, int ns,
    const int32_t *__restrict__ w_dst, int n, float *__restrict__ out, long ldo) {
    constexpr int GL = LT_TILE_GL;
    constexpr int GPW = 64 / GL, IPB = (LT_BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int j = lane & (GL - 1);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int xps = 8 / ns;
    const int slice = xcd % ns;
    const int it = (q * xps + xcd / ns) * IPB + (threadIdx.x >> 6) * GPW + lane / GL;
    if (it >= n_items) return;
    const int e0 = __builtin_nontemporal_load(w_e0 + it);
    const int cnt = __builtin_nontemporal_load(w_cnt + it);
    const int coff = slice * 4 * GL + 4 * j;
    const bool active = coff < ncols;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 x = {0u, 0u, 0u, 0u};
    const char *Sb = reinterpret_cast<const char *>(S);
    const size_t rowbytes = (size_t)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const int e1 = e0 + cnt;
    int nxc = 0;
    if (e0 + j < e1) nxc = __builtin_nontemporal_load(col + e0 + j);
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        nxc = 0;
        if (me + GL < e1) nxc = __builtin_nontemporal_load(col + me + GL);
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                u32x4 s[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = row_bcast<k>(myc);
                    s[u] = u32x4{0u, 0u, 0u, 0u};
                    if (k < left && active) {
                        if (BIG) s[u] = *reinterpret_cast<const u32x4 *>(Sb + ((size_t)c * rowbytes + loff));
                        else s[u] = *reinterpret_cast<const u32x4 *>(Sb + (size_t)((unsigned)c * (unsigned)rowbytes + loff));
                    }
                });
                static_for<U>([&](auto ut) { x ^= s[decltype(ut)::value]; });
            }
        });
    }
    if (active && j == 0) sink[(size_t)it * ns + slice] = x.x ^ x.y ^ x.z ^ x.w;
    // out != NULL: the result rows leave as in the real kernel (16 bytes per lane, non-temporal) -- "gathers + result stores"
    if (out != nullptr && active) {
        const int dst = __builtin_nontemporal_load(w_dst + it);
        if (dst < n) __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(out + (size_t)dst * ldo + coff));
    }
}

extern "C" size_t lt_spmm_gather_ceiling_bytes(const lt_graph *g) {
    return g ? ((size_t)g->w_n * 4 + 64) * sizeof(unsigned) : 0;
}

extern "C" int lt_spmm_gather_ceiling(const lt_graph *g, const float *S, int64_t lds, int32_t ncols, int32_t in_flight,
                                      void *sink, size_t sink_bytes, float *out, int64_t ldo, void *stream) {
    LT_REQUIRE(g != nullptr && S != nullptr && sink != nullptr, "lt_spmm_gather_ceiling: NULL pointer");
    LT_REQUIRE(ncols > 0 && ncols % 4 == 0 && ncols <= 4 * LT_TILE_GL * 4 && lds >= ncols,
               "lt_spmm_gather_ceiling: ncols=%d (a multiple of 4 up to %d, one pass of the tiled kernel)", ncols, 16 * LT_TILE_GL);
    LT_REQUIRE(in_flight == 8 || in_flight == 16, "lt_spmm_gather_ceiling: in_flight = 8 (the kernel's own) or 16");
    LT_REQUIRE(out == nullptr || (ldo >= ncols && ldo % 4 == 0 && (uintptr_t)out % 16 == 0), "lt_spmm_gather_ceiling: out / ldo");
    LT_REQUIRE(sink_bytes >= lt_spmm_gather_ceiling_bytes(g), "lt_spmm_gather_ceiling: sink needs %zu bytes", lt_spmm_gather_ceiling_bytes(g));
    if (g->w_n == 0) return LT_OK;
    int ns = (ncols + 4 * LT_TILE_GL - 1) / (4 * LT_TILE_GL);
    ns = ns <= 1 ? 1 : (ns == 2 ? 2 : 4);
    const int xps = 8 / ns;
    constexpr int IPB = (LT_BLOCK / 64) * (64 / LT_TILE_GL);
    const long chunks = ((long)g->w_n + IPB - 1) / IPB;
    const long grid = 8 * ((chunks + xps - 1) / xps);
    LT_REQUIRE(grid < 2147483647L, "lt_spmm_gather_ceiling: grid limit");
    const bool big = (unsigned long long)g->n * (unsigned long long)lds * 4ull >= (1ull << 32);
    hipStream_t st = (hipStream_t)stream;
#define LT_GC_LAUNCH(U_, BIG_)                                                                                          \
    hipLaunchKernelGGL((k_rows_tiled_gathers_only<U_, BIG_>), dim3((unsigned)grid), dim3(LT_BLOCK), 0, st, g->w_n, g->w_e0, \
                       g->w_cnt, g->col, S, (long)lds, ncols, (unsigned *)sink, ns, g->w_dst, g->n, out, (long)ldo)
    if (in_flight == 8) { if (big) LT_GC_LAUNCH(8, true); else LT_GC_LAUNCH(8, false); }
    else { if (big) LT_GC_LAUNCH(16, true); else LT_GC_LAUNCH(16, false); }
#undef LT_GC_LAUNCH
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// --------------------------------------------------------------------------------------------
// small graphs: one LPR-lane group per row (4 columns per lane), rows of up to LT_ROW_SEG entries
// --------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(LT_BLOCK) void k_spmm_rows(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S, int lds, int ncols,
    const float *__restrict__ bias, int relu, float *__restrict__ out, int ldo) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int gl = lane & (LPR - 1);
    int r = wave * RPW + lane / LPR;
    if (LPR == 64) r = __builtin_amdgcn_readfirstlane(r);
    if (r >= n) return;
    const int coff = 4 * gl;
    const bool active = coff < ncols;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    if (e1 - e0 > LT_ROW_SEG) return;  // long rows: k_spmm_segments + k_spmm_long_combine
    const f32x4 acc = seg_chain<8>(col, val, e0, e1, S, lds, coff, active, -1, nullptr, f32x4{0.f, 0.f, 0.f, 0.f});
    if (!active) return;
    f32x4 o = acc;
    if (bias) {
        const f32x4 b = ld4(bias + coff);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    if (relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<f32x4 *>(out + (size_t)r * ldo + coff) = o;
}

// long rows, small-graph route: one lane group per segment -> partial[seg, :]
template <int LPR>
__global__ __launch_bounds__(LT_BLOCK) void k_spmm_segments(
    int n_seg, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ long_row,
    const int32_t *__restrict__ seg_long, const int32_t *__restrict__ seg_begin,
    const int32_t *__restrict__ col, const float *__restrict__ val, const float *__restrict__ S,
    int lds, int ncols, float *__restrict__ partial, int ldp) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int gl = lane & (LPR - 1);
    int sg = wave * RPW + lane / LPR;
    if (LPR == 64) sg = __builtin_amdgcn_readfirstlane(sg);
    if (sg >= n_seg) return;
    const int coff = 4 * gl;
    if (coff >= ncols) return;
    const int r = long_row[seg_long[sg]];
    const int b = seg_begin[sg];
    const int e = min(b + LT_ROW_SEG, rowptr[r + 1]);
    const f32x4 acc = seg_chain<16>(col, val, b, e, S, lds, coff, true, -1, nullptr, f32x4{0.f, 0.f, 0.f, 0.f});
    *reinterpret_cast<f32x4 *>(partial + (size_t)sg * ldp + coff) = acc;
}

// ... then one thread per (long row, column): segment sums added in segment order, epilogue, store
__global__ void k_spmm_long_combine(int n_long, const int32_t *__restrict__ long_row,
                                    const int32_t *__restrict__ long_segptr,
                                    const float *__restrict__ partial, long ldp, int ncols,
                                    const float *__restrict__ bias, int relu, float *__restrict__ out,
                                    long ldo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_long * ncols) return;
    const int li = (int)(i / ncols), c = (int)(i % ncols);
    // the biggest hub has hundreds of segments: 16 loads in flight, adds in segment order
    const int s0 = long_segptr[li], s1 = long_segptr[li + 1];
    float acc = partial[(size_t)s0 * ldp + c];
    int sg = s0 + 1;
    for (; sg + 16 <= s1; sg += 16) {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = __builtin_nontemporal_load(partial + (size_t)(sg + k) * ldp + c);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += t[k];
    }
    for (; sg < s1; ++sg) acc += partial[(size_t)sg * ldp + c];
    if (bias) acc += bias[c];
    if (relu) acc = fmaxf(acc, 0.f);
    out[(size_t)long_row[li] * ldo + c] = acc;
}

// SpMM, narrow right-hand side (ncols <= 8), 8 lanes per row (the layer-2 shape).
template <int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_spmm_narrow(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ T, int ldt, int C,
    const float *__restrict__ bias, int relu, float *__restrict__ out, int ldo) {
    const int gid = (blockIdx.x * LT_BLOCK + threadIdx.x) / LT_L2_LANES;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    if (gid >= n) return;  // whole 8-lane groups exit together
    float acc[CP];
    row2_dot<CP>(col, val, rowptr[gid], rowptr[gid + 1], q, C,
                 [&](int c, int) { return T + (size_t)c * ldt; }, acc);
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) {
                float o = acc[c];
                if (bias) o += bias[c];
                if (relu) o = fmaxf(o, 0.f);
                out[(size_t)gid * ldo + c] = o;
            }
    }
}

static inline unsigned blocks_for(long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// one pass of the row kernels: ncols % 4 == 0, ncols <= LT_MAX_H, 16-byte aligned operands
static int spmm_wide_slice(const lt_graph *g, const float *S, int64_t lds, int ncols, const float *bias, int relu,
                           float *out, int64_t ldo, hipStream_t st) {
    const int lpr = lt_lpr_for(ncols);
    if (lt_tiled_wanted(g, ncols)) {
        const int rc = lt_launch_rows_tiled(g, S, lds, ncols, nullptr, bias, relu, out, ldo, g->p_seg_scratch, LT_MAX_H, st);
        if (rc) return rc;
    } else {
        const unsigned grid = blocks_for(g->n, (LT_BLOCK / 64) * (64 / lpr));
        LT_DISPATCH_LPR(lpr,
            hipLaunchKernelGGL((k_spmm_rows<LPR_>), dim3(grid), dim3(LT_BLOCK), 0, st, g->n, g->rowptr,
                               g->col, g->val, S, (int)lds, ncols, bias, relu, out, (int)ldo));
        LT_CHECK_LAUNCH();
        if (g->p_n_long > 0) {
            const unsigned gseg = blocks_for(g->p_n_seg, (LT_BLOCK / 64) * (64 / lpr));
            LT_DISPATCH_LPR(lpr,
                hipLaunchKernelGGL((k_spmm_segments<LPR_>), dim3(gseg), dim3(LT_BLOCK), 0, st, g->p_n_seg,
                                   g->rowptr, g->p_long_row, g->p_seg_long, g->p_seg_begin, g->col, g->val, S,
                                   (int)lds, ncols, g->p_seg_scratch, LT_MAX_H));
            LT_CHECK_LAUNCH();
        }
    }
    if (g->p_n_long > 0) {
        const long tot = (long)g->p_n_long * ncols;
        hipLaunchKernelGGL(k_spmm_long_combine, dim3(blocks_for(tot, 256)), dim3(256), 0, st, g->p_n_long,
                           g->p_long_row, g->p_long_segptr, g->p_seg_scratch, (long)LT_MAX_H, ncols, bias, relu, out,
                           (long)ldo);
        LT_CHECK_LAUNCH();
    }
    return LT_OK;
}

// 1 when lt_spmm_csr_f32 would take the tiled (column-sliced work-item) route for `ncols` columns on this graph, 0 for
// the row kernels: what a benchmark reports as the kernel it timed.
extern "C" int lt_spmm_route(const lt_graph *g, int32_t ncols) {
    if (!g || ncols <= 0) return 0;
    const int w = ncols > LT_MAX_H ? LT_MAX_H : ncols;
    return (w % 4 == 0 && lt_tiled_wanted(g, w)) ? 1 : 0;
}

extern "C" int lt_spmm_csr_f32(const lt_graph *g, const float *S, int64_t lds, int32_t ncols,
                               const float *bias, int32_t relu, float *out, int64_t ldo,
                               void *stream) {
    LT_REQUIRE(g != nullptr, "lt_spmm_csr_f32: graph is NULL");
    LT_REQUIRE(ncols > 0, "lt_spmm_csr_f32: ncols=%d", ncols);
    LT_REQUIRE(S != nullptr && out != nullptr, "lt_spmm_csr_f32: S/out is NULL");
    LT_REQUIRE(lds >= ncols && ldo >= ncols, "lt_spmm_csr_f32: leading dimension < ncols");
    LT_REQUIRE(lds < INT32_MAX && ldo < INT32_MAX, "lt_spmm_csr_f32: leading dimension too large");
    hipStream_t st = (hipStream_t)stream;
    if (g->n == 0) return LT_OK;
    lt_prof_scope prof_(LT_K_SPMM, st);
    // Every output column is its own chain, so a layer wider than one pass of the row kernels (LT_MAX_H = 64 lanes x 4
    // columns) is served slice by slice with the bits a single wide pass would give (gcn/layers.py:30-36 has no width
    // limit).  The 16-byte vector path needs aligned rows; what it cannot take -- a tail of ncols % 4 columns, operands
    // with odd leading dimensions -- goes through the 8-lane narrow kernel, LT_MAX_C columns per launch.
    const bool vec_ok = lds % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)S % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                        (!bias || (uintptr_t)bias % 16 == 0);
    int c0 = 0;
    if (vec_ok && ncols > LT_MAX_C) {
        const int wide = ncols / 4 * 4;
        for (; c0 < wide; c0 += LT_MAX_H) {
            const int wcols = wide - c0 < LT_MAX_H ? wide - c0 : LT_MAX_H;
            const int rc = spmm_wide_slice(g, S + c0, lds, wcols, bias ? bias + c0 : nullptr, relu, out + c0, ldo, st);
            if (rc) return rc;
        }
        c0 = wide;
    } else if (vec_ok && ncols % 4 == 0) {   // 4 or 8 columns, aligned: the vector path as before
        return spmm_wide_slice(g, S, lds, ncols, bias, relu, out, ldo, st);
    }
    const unsigned grid = blocks_for(g->n, LT_BLOCK / LT_L2_LANES);
    for (; c0 < ncols; c0 += LT_MAX_C) {
        const int wcols = ncols - c0 < LT_MAX_C ? ncols - c0 : LT_MAX_C;
        LT_DISPATCH_CP(lt_cp_for(wcols),
            hipLaunchKernelGGL((k_spmm_narrow<CP_>), dim3(grid), dim3(LT_BLOCK), 0, st, g->n,
                               g->rowptr, g->col, g->val, S + c0, (int)lds, wcols, bias ? bias + c0 : nullptr, relu, out + c0,
                               (int)ldo));
        LT_CHECK_LAUNCH();
    }
    return LT_OK;
}
