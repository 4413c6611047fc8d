// fp64-accumulated baseline pre-activation for LT_MODE_DELTA.
//
// The reference's quantity is a finite difference across ReLU kinks: a hidden unit whose
// pre-activation z lies within |dz| ~ 1e-4*|s| of zero contributes (relu(z + dz) - relu(z)) / d,
// which depends on z itself to an absolute 1e-6 -- below what an fp32-accumulated X*W1 (K = 3170)
// and SpMM can deliver (measured: 1e-4 relative error on exactly those matrix entries).  So the
// kink test of the delta kernels reads Z1 = A_hat (X W1) + b1 accumulated in fp64:
//   * S1d = X*W1 with v_mfma_f64_16x16x4_f64 (fp32 operands widened on the LDS read), split-K with
//     an ordered slab sum;
//   * Z1d = A_hat*S1d + b1, row-owned fp64 fma chains.
// One-off cost per baseline (not per probe); only built when delta mode is used.
#include <new>
#include <type_traits>

#include "lt_rows.hip.h"
#include "lt_items.hip.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u_ __attribute__((ext_vector_type(4), aligned(4)));

#define GD_BM 64
#define GD_BN 64
#define GD_BK 16
#define GD_LDA (GD_BK + 1)
#define GD_LDB (GD_BN + 16)  // +16 floats: the 4 k-rows a wave reads at once land in disjoint bank groups

__global__ __launch_bounds__(256) void k_gemm_f64acc(const float *__restrict__ A, long lda,
                                                     const float *__restrict__ B, long ldb,
                                                     double *__restrict__ C, long ldc, int M, int N, int K,
                                                     int kslice, long slab_stride) {
    __shared__ __attribute__((aligned(16))) float As[2][GD_BM * GD_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GD_BK * GD_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GD_BM, n0 = blockIdx.y * GD_BN;
    const int kb = blockIdx.z * kslice, ke = min(K, kb + kslice);
    C += (long)blockIdx.z * slab_stride;

    const int a_row = tid >> 2, a_col = (tid & 3) * 4;
    const int b_row = tid >> 4, b_col = (tid & 15) * 4;
    const bool a_row_ok = (m0 + a_row) < M;
    const float *a_ptr = A + (long)(m0 + a_row) * lda + a_col;
    const float *b_ptr = B + (long)b_row * ldb + n0 + b_col;
    const bool b_full = (n0 + b_col + 3) < N;
    f32x4 ra, rb;
    auto load_tiles = [&](int k0) {
        ra = f32x4{0.f, 0.f, 0.f, 0.f};
        rb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_row_ok) {
            if (k0 + a_col + 3 < ke) ra = *reinterpret_cast<const f32x4u_ *>(a_ptr + k0);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + a_col + j < ke) ra[j] = a_ptr[k0 + j];
        }
        if (k0 + b_row < ke) {
            const float *p = b_ptr + (long)k0 * ldb;
            if (b_full) rb = *reinterpret_cast<const f32x4u_ *>(p);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + b_col + j < N) rb[j] = p[j];
        }
    };
    auto store_tiles = [&](int buf) {
        float *as = &As[buf][a_row * GD_LDA + a_col];
        as[0] = ra.x; as[1] = ra.y; as[2] = ra.z; as[3] = ra.w;
        *reinterpret_cast<f32x4 *>(&Bs[buf][b_row * GD_LDB + b_col]) = rb;
    };

    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int nk = (ke - kb + GD_BK - 1) / GD_BK;
    load_tiles(kb);
    store_tiles(0);
    __syncthreads();
    // v_mfma_f64_16x16x4_f64 operands: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]
    const int a_frag = (wr * 32 + (lane & 15)) * GD_LDA + (lane >> 4);
    const int b_frag = (lane >> 4) * GD_LDB + wc * 32 + (lane & 15);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kb + (kt + 1) * GD_BK);
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
#pragma unroll
        for (int kk = 0; kk < GD_BK; kk += 4) {
            const double a0 = (double)as[kk], a1 = (double)as[16 * GD_LDA + kk];
            const double b0 = (double)bs[kk * GD_LDB], b1 = (double)bs[kk * GD_LDB + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg  (NOT the f32 map)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cn = n0 + wc * 32 + j * 16 + (lane & 15);
            if (cn >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cm = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * reg;
                if (cm < M) C[(long)cm * ldc + cn] = acc[i][j][reg];
            }
        }
}

// ---- 128x128 block tile for the big product (N a multiple of 128, M >= 1024): each of the 4 waves owns a 64x64
// quadrant = 4x4 MFMA tiles of 16x16, so one k-step (4 deep) is 16 v_mfma_f64_16x16x4_f64 (1024 cycles) fed by 4 + 4
// LDS fragment reads; global loads run two k-tiles ahead of the LDS stores in two register stages and the steady
// state is branch-free, exactly as in k_gemm_f32_mfma_128 (lt_gemm.hip).  fp32 operands are widened on the LDS read.
#define GE_BM 128
#define GE_BN 128
#define GE_BK 16
#define GE_LDA (GE_BK + 1)
#define GE_LDB (GE_BN + 16)   // +16 floats: the k-rows a 32-lane group reads land in disjoint banks
#define GE_PASS (GE_BK / 8)
__global__ __launch_bounds__(256, 2) void k_gemm_f64acc_128(const float *__restrict__ A, long lda,
                                                         const float *__restrict__ B, long ldb,
                                                         double *__restrict__ C, long ldc, int M, int N, int K,
                                                         int kslice, long slab_stride) {
    __shared__ __attribute__((aligned(16))) float As[2][GE_BM * GE_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GE_BK * GE_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GE_BM, n0 = blockIdx.y * GE_BN;
    const int kb = blockIdx.z * kslice, ke = min(K, kb + kslice);
    C += (long)blockIdx.z * slab_stride;
    constexpr int A_TPR = GE_BK / 4, A_RPP = 256 / A_TPR;
    const int a_row = tid / A_TPR, a_col = (tid % A_TPR) * 4;
    const int b_row = tid >> 5, b_col = (tid & 31) * 4;
    const float *a_ptr[GE_PASS], *b_ptr[GE_PASS];
#pragma unroll
    for (int p = 0; p < GE_PASS; ++p) {
        a_ptr[p] = A + (long)min(m0 + a_row + p * A_RPP, M - 1) * lda + a_col + kb;   // rows past M: clamped, never stored
        b_ptr[p] = B + (long)(kb + b_row + 8 * p) * ldb + n0 + b_col;
    }
    const long b_step = (long)GE_BK * ldb;
    f32x4 ra[2][GE_PASS], rb[2][GE_PASS];
    auto load_full = [&](auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GE_PASS; ++p) {
            ra[S][p] = *reinterpret_cast<const f32x4u_ *>(a_ptr[p]);
            rb[S][p] = *reinterpret_cast<const f32x4u_ *>(b_ptr[p]);
            a_ptr[p] += GE_BK;
            b_ptr[p] += b_step;
        }
    };
    auto load_tail = [&](int k0, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GE_PASS; ++p) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + a_col + j < ke) r[j] = a_ptr[p][j];
            ra[S][p] = r;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k0 + b_row + 8 * p < ke) t = *reinterpret_cast<const f32x4u_ *>(b_ptr[p]);
            rb[S][p] = t;
        }
    };
    auto store_tiles = [&](int buf, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GE_PASS; ++p) {
            float *as = &As[buf][(a_row + p * A_RPP) * GE_LDA + a_col];
            as[0] = ra[S][p].x; as[1] = ra[S][p].y; as[2] = ra[S][p].z; as[3] = ra[S][p].w;
            *reinterpret_cast<f32x4 *>(&Bs[buf][(b_row + 8 * p) * GE_LDB + b_col]) = rb[S][p];
        }
    };
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    // operands of v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]
    const int a_frag = (wr * 64 + (lane & 15)) * GE_LDA + (lane >> 4);
    const int b_frag = (lane >> 4) * GE_LDB + wc * 64 + (lane & 15);
    auto multiply = [&](int buf) {
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
#pragma unroll
        for (int kk = 0; kk < GE_BK; kk += 4) {
            double a[4], bq[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = (double)as[t * 16 * GE_LDA + kk];
                bq[t] = (double)bs[kk * GE_LDB + t * 16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nfull = (ke - kb) / GE_BK;
    const bool partial = (ke - kb) % GE_BK != 0;
    if (nfull > 0) {
        load_full(S0{});
        store_tiles(0, S0{});
        if (nfull > 1) load_full(S1{});
        __syncthreads();
        auto step = [&](int kt, auto even_tag) {
            constexpr int E = decltype(even_tag)::value;
            if (kt + 2 < nfull) load_full(std::integral_constant<int, E>{});
            multiply(E);
            if (kt + 1 < nfull) store_tiles(E ^ 1, std::integral_constant<int, E ^ 1>{});
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 3 < nfull; kt += 2) {     // steady state: two tiles per trip, everything unconditional
            load_full(S0{});
            multiply(0);
            store_tiles(1, S1{});
            __syncthreads();
            load_full(S1{});
            multiply(1);
            store_tiles(0, S0{});
            __syncthreads();
        }
        for (; kt + 1 < nfull; kt += 2) {
            step(kt, S0{});
            step(kt + 1, S1{});
        }
        if (kt < nfull) step(kt, S0{});
    }
    if (partial) {
        load_tail(kb + nfull * GE_BK, S0{});
        store_tiles(0, S0{});
        __syncthreads();
        multiply(0);
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cn = n0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cm = m0 + wr * 64 + i * 16 + (lane >> 4) + 4 * reg;
                if (cm < M) C[(long)cm * ldc + cn] = acc[i][j][reg];
            }
        }
}

__global__ void k_sum_slabs_f64(const double *__restrict__ slabs, long slab_stride, int splits, long total,
                                int N, double *__restrict__ C, long ldc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double acc = slabs[i];
    for (int z = 1; z < splits; ++z) acc += slabs[(long)z * slab_stride + i];
    C[(i / N) * ldc + (i % N)] = acc;
}

// The same sum, the rows leaving as 32-bit FIXED POINT with one scale per row (the storage of the feature-difference route,
// k_s1d_feature_rows: q = round(value / scale), scale = row max / 2^31): one wave per row, a lane per 4 columns (H <= 256).
// Round 5: the matrix-core route stores its product rows like that too -- half the bytes the fp64 SpMM gathers and stage A reads.
__global__ __launch_bounds__(256) void k_sum_slabs_f64_q(const double *__restrict__ slabs, long slab_stride, int splits, int M, int N,
                                                         int Hp, float *__restrict__ S1x, double *__restrict__ S1qs,
                                                         int32_t *__restrict__ zstate = nullptr, unsigned *__restrict__ zero_words = nullptr,
                                                         int n_zero = 0) {
    // zstate != NULL: every row's pre-activation is marked stale here (saves the refresh a memset launch, as k_s1d_feature_rows does);
    // zero_words: the int8 split's exponent words, cleared for the NEXT refresh's k_i8_w_max (the product kernel in front of this
    // launch was their last reader) -- two ~3 us memsets of a 0.12 ms dense-feature build
    if (zero_words && blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0u;
    const int row = (int)(((long)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    if (zstate && lane == 0) zstate[row] = 0;
    const int c0 = 4 * lane;
    double o[4] = {0.0, 0.0, 0.0, 0.0};
    if (c0 < N) {       // (N % 4 == 0: 32-byte loads, every slab's in flight before the first add; slab order as k_sum_slabs_f64)
        const double *p = slabs + (long)row * N + c0;
        f64x4 acc = *reinterpret_cast<const f64x4 *>(p);
        int z = 1;
        for (; z + 4 <= splits; z += 4) {
            f64x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f64x4 *>(p + (long)(z + t) * slab_stride);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += v[t][k];
        }
        for (; z < splits; ++z) {
            const f64x4 v = *reinterpret_cast<const f64x4 *>(p + (long)z * slab_stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acc[k];
    }
    double mx = fmax(fmax(fabs(o[0]), fabs(o[1])), fmax(fabs(o[2]), fabs(o[3])));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
    const double scale = mx > 0.0 ? mx * (1.0 / 2147483000.0) : 1.0;
    const double inv = 1.0 / scale;
    if (c0 < Hp) {
        int q[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = (int)rint(o[t] * inv);
        *reinterpret_cast<int4 *>(S1x + (size_t)row * Hp + c0) = make_int4(q[0], q[1], q[2], q[3]);
    }
    if (lane == 0) S1qs[row] = scale;
}

// The same storage from finished fp64 rows (S[M, ld], pad columns zero): the product of ONE K slice, and the rows the ranks'
// all-gather rebuilt (multi-GPU) -- the same value gives the same words, so sharded and single-GPU runs keep every bit.
__global__ __launch_bounds__(256) void k_quant_rows_f64(const double *__restrict__ S, long ld, int M, int N, int Hp,
                                                        float *__restrict__ S1x, double *__restrict__ S1qs) {
    const int row = (int)(((long)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int c0 = 4 * lane;
    double o[4] = {0.0, 0.0, 0.0, 0.0};
    if (c0 < N) {
        const f64x4 v = *reinterpret_cast<const f64x4 *>(S + (long)row * ld + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k];
    }
    double mx = fmax(fmax(fabs(o[0]), fabs(o[1])), fmax(fabs(o[2]), fabs(o[3])));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
    const double scale = mx > 0.0 ? mx * (1.0 / 2147483000.0) : 1.0;
    const double inv = 1.0 / scale;
    if (c0 < Hp) {
        int q[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = (int)rint(o[t] * inv);
        *reinterpret_cast<int4 *>(S1x + (size_t)row * Hp + c0) = make_int4(q[0], q[1], q[2], q[3]);
    }
    if (lane == 0) S1qs[row] = scale;
}

// Z1d[r, :] = sum_e val[e] * S1d[col[e], :] + b1     (fp64 fma chain in CSR order, 4 columns per lane)
// One lane group per row of up to LT_ROW_SEG entries; on a graph with hub rows the first seg_blocks blocks of the
// launch take one SEGMENT of a long row per lane group instead, raw sum into seg_out[segment] (k_spmm_f64_long adds
// them in segment order and the bias).  fp64: the cut only decides how a hub row's work is spread.
// ST = double, or "float": the feature route stores its fp64-ACCUMULATED product as 32-bit fixed point with a scale per row (S1x +
// S1qs: 31 bits against the row's largest value, half the bytes every gather moves; plain fp32 rows were not enough -- a unit within
// dz of its kink carries the rounding of the terms into the result as an absolute error, up to 7e-5 of the largest score);
// the chains accumulate in fp64.
template <int LPR, typename ST>
__global__ __launch_bounds__(256) void k_spmm_f64(int n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col,
                                                  const float *__restrict__ val,
                                                  const ST *__restrict__ S, int ld,
                                                  const float *__restrict__ b1p,
                                                  double *__restrict__ out, int seg_blocks, int n_seg,
                                                  const int32_t *__restrict__ seg_begin,
                                                  const int32_t *__restrict__ seg_long,
                                                  const int32_t *__restrict__ long_row,
                                                  double *__restrict__ seg_out, int32_t *__restrict__ state,
                                                  const double *__restrict__ rs, const double *__restrict__ crefv,
                                                  const lt_bits_job job = lt_bits_job{}, const int job_first = 0,
                                                  float *__restrict__ outf = nullptr, const double *__restrict__ Sq = nullptr,
                                                  const int seg_len = LT_F64_SEG, const int long_thr = LT_F64_LONG) {
    // (seg_len / long_thr: the cut of the segment tables the caller passes -- the graph's fp64 tables: rows of more than long_thr
    // entries in seg_len-entry segments)
    // ST = float: S holds int32 fixed point, row c scaled by Sq[c] (k_s1d_feature_rows): a term is A_hat[r, c] * Sq[c] * q
    // outf != NULL: the finished rows go there rounded once to fp32 instead of to `out` (the segment sums stay fp64).  The
    // kink test of LT_MODE_DELTA reads a pre-activation for its SIGN, the sign of z + dz and, where they differ, its value: a
    // relative rounding of z (6e-8) moves none of the three by more than 6e-8 of what the exact z gives -- unlike a rounding of
    // the terms z is summed from.  Half the bytes stage A gathers per item (2 KB -> 1 KB) and this kernel writes.
    // job.nblocks > 0: the blocks from job_first on build the item tables of a probe chunk (k_item_bits' blocks: nothing in
    // this launch depends on them, and the launch in front of this one that they used to be cost the step 4 us)
    // job_first < 0: the job's blocks are the FIRST of the grid instead (for a job whose blocks outlast a block of rows; the record
    // gather -- 4.5 us on its own -- is better off last: 11.1 against 11.5 us for the launch at twitch size)
    int bx = (int)blockIdx.x;
    // job.zero_blocks > 0: the LAST blocks of the grid zero-fill rows of lt_influence_rows_f64's matrix (lt_items.hip.h zero_rows_wave)
    if (job.zero_blocks > 0 && bx >= (int)gridDim.x - job.zero_blocks) {
        zero_rows_wave(job, bx - ((int)gridDim.x - job.zero_blocks));
        return;
    }
    if (job.nblocks > 0) {
        extern __shared__ __attribute__((aligned(16))) unsigned char spmm_job_smem[];      // (job.smem_bytes: the record gather's list)
        if (job_first < 0) {
            if (bx < job.nblocks) { item_bits_block(bx, job, spmm_job_smem); return; }
            bx -= job.nblocks;
        } else if (bx >= job_first) {
            item_bits_block(bx - job_first, job, spmm_job_smem);
            return;
        }
    }
    // crefv != NULL: S holds the feature rows' products WITHOUT the reference vector's (S1d - cref, lt_fp64 "deferred cref");
    // the row's share rs[r] * cref (rs = the row sum of A_hat) is added with the bias
    // state != NULL (on-demand, lt_fp64_prepare_rows): only the rows marked 2 are formed; a short row is marked 1 (valid) by its
    // own lane group, a hub row by k_spmm_f64_long once its segments are summed
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const bool SEG = bx < seg_blocks;      // block-uniform: the first blocks take the segments of the hub rows
    const int wave = ((SEG ? bx : bx - seg_blocks) * 256 + threadIdx.x) >> 6;
    int r = wave * RPW + lane / LPR;
    if (LPR == 64) r = __builtin_amdgcn_readfirstlane(r);
    if (r >= (SEG ? n_seg : n)) return;
    const int coff = 4 * gl;
    if (coff >= ld) return;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    int e, e1;
    if (SEG) {
        const int lr = long_row[seg_long[r]];
        if (state && state[lr] != 2) return;
        e = seg_begin[r];
        e1 = min(e + seg_len, rowptr[lr + 1]);
    } else {
        if (state && state[r] != 2) return;
        e = rowptr[r];
        e1 = rowptr[r + 1];
        if (seg_blocks > 0 && e1 - e > long_thr) return;
    }
    typedef ST sx4 __attribute__((ext_vector_type(4)));
    constexpr bool QNT = sizeof(ST) == 4;        // "float" rows are int32 fixed point with a scale per row (Sq)
    typedef int qx4 __attribute__((ext_vector_type(4)));
    auto term = [](const sx4 &sv, int k) -> double {
        if constexpr (QNT) return (double)__builtin_bit_cast(qx4, sv)[k];
        else return (double)sv[k];
    };
    if constexpr (QNT) {
        // sixteen in flight for the 16-byte fixed-point rows (round 5): a row or segment of 128 entries is 8 dependent trips instead
        // of 16 -- on a graph with rows of 64 .. 128 entries those lane groups are the launch (19 us against 11 for the same graph
        // size without them); the fma chain stays in entry order: same bits
        for (; e + 16 <= e1; e += 16) {
            double a[16];
            sx4 s[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = col[e + j];
                a[j] = (double)val[e + j] * Sq[c];
                s[j] = *reinterpret_cast<const sx4 *>(S + (size_t)c * ld + coff);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fma(a[j], term(s[j], k), acc[k]);
        }
    }
    for (; e + 8 <= e1; e += 8) {   // eight gathers in flight (a trip costs one L2 / Infinity-Cache latency); entry order kept
        double a[8];
        sx4 s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = col[e + j];
            a[j] = (double)val[e + j];
            if constexpr (QNT) a[j] *= Sq[c];
            s[j] = *reinterpret_cast<const sx4 *>(S + (size_t)c * ld + coff);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fma(a[j], term(s[j], k), acc[k]);
    }
    // (round 5 tried the tail as one masked batch of 8, and batches of 8 with the next (col, val) prefetched: both SLOWER -- 12.7 ->
    // 13.6 us for the launch at twitch size: the redundant gathers of the masked slots cost more than the trips they save)
    for (; e + 4 <= e1; e += 4) {
        double a[4];
        sx4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col[e + j];
            a[j] = (double)val[e + j];
            if constexpr (QNT) a[j] *= Sq[c];
            s[j] = *reinterpret_cast<const sx4 *>(S + (size_t)c * ld + coff);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fma(a[j], term(s[j], k), acc[k]);
    }
    for (; e < e1; ++e) {
        const int c = col[e];
        double a = (double)val[e];
        if constexpr (QNT) a *= Sq[c];
        const sx4 s = *reinterpret_cast<const sx4 *>(S + (size_t)c * ld + coff);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fma(a, term(s, k), acc[k]);
    }
    if (!SEG) {
        const f32x4 b = ld4(b1p + coff);
        if (crefv) {
            const double w = rs[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fma(w, crefv[coff + k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += (double)b[k];
    }
    if (outf && !SEG) *reinterpret_cast<f32x4 *>(outf + (size_t)r * ld + coff) = f32x4{(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
    else *reinterpret_cast<f64x4 *>((SEG ? seg_out : out) + (size_t)r * ld + coff) = acc;
    if (state && !SEG && coff == 0) state[r] = 1;
}
// the hub rows: segment sums added in segment order + the bias (`state`: as in k_spmm_f64; the marked rows are set valid by a
// second launch with `finish` = 1, once every column of the row has been written)
__global__ void k_spmm_f64_long(int n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_segptr,
                                const double *__restrict__ part, int ld, const float *__restrict__ b1p,
                                double *__restrict__ out, int32_t *__restrict__ state, int finish,
                                const double *__restrict__ rs, const double *__restrict__ crefv, float *__restrict__ outf = nullptr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (finish) {
        if (i < n_long && state[long_row[i]] == 2) state[long_row[i]] = 1;
        return;
    }
    if (i >= (long)n_long * ld) return;
    const int li = (int)(i / ld), c = (int)(i % ld);
    if (state && state[long_row[li]] != 2) return;
    // (segment order; eight loads in flight -- as k_y_long: one dependent load per segment was most of this launch's 5 us; sixteen while
    // there are that many: a 1 749-entry row is 55 segments)
    const int s0 = long_segptr[li], s1 = long_segptr[li + 1];
    double acc = part[(size_t)s0 * ld + c];
    int sg = s0 + 1;
    for (; sg + 16 <= s1; sg += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = part[(size_t)(sg + t) * ld + c];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc += v[t];
    }
    for (; sg + 8 <= s1; sg += 8) {
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = part[(size_t)(sg + t) * ld + c];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc += v[t];
    }
    if (sg < s1) {      // (the tail in one trip too: loads past the row's last segment read the last again and are not added)
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = part[(size_t)min(sg + t, s1 - 1) * ld + c];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (sg + t < s1) acc += v[t];
    }
    if (crefv) acc = fma(rs[long_row[li]], crefv[c], acc);
    if (outf) outf[(size_t)long_row[li] * ld + c] = (float)(acc + (double)b1p[c]);      // (as k_spmm_f64's outf)
    else out[(size_t)long_row[li] * ld + c] = acc + (double)b1p[c];
}

// ---- S1d = X*W1 from the DIFFERENCES of the feature rows to one reference row ---------------------------------------
//   S1d[i, :] = m W1 + sum_{j : X[i,j] != m[j]} (X[i,j] - m[j]) * W1[j, :]
// is an identity for ANY reference vector m (in fp64 the difference of two fp32 values is exact, so the terms are the
// exact products the dense sum holds; only the order of an fp64 summation changes).  It pays when the rows differ from m
// in few columns -- which is how the reference's own twitch features look: utils/load.py:53-59 builds 0/1 indicator
// rows (a few dozen of 3170 set) and worker.py standardises them per column, so every column holds TWO values and, with
// m[j] = the value most of the rows hold, a row differs from m exactly where its own features are set.  Then the fp64
// product is one pass over X (N*F*4 bytes) and ~ 20 W1 rows per node instead of 2*N*F*H flops on the f64 matrix cores
// (twitch-RU: 7.1 GFLOP -> 0.05).
//   k_ref_vector        m[j] = the more frequent of (min, max) of column j over the first <= 64 rows (once per baseline)
//   k_ref_product       cref = m W1 (every refresh: W1 may have changed)
//   k_s1d_feature_rows  one wave per row: ALL loads of the row go out first (one HBM round trip), the reference vector is
//                       staged in LDS meanwhile, the differing columns are compacted into a per-wave LDS list (ballot +
//                       prefix) and every lane walks the list for its 4 hidden columns.  A row with more differing columns
//                       than the list holds is read again piecewise (slow and correct); a row with more than `hint_cap`
//                       sets *dense_hint -- a word of mapped host memory the host looks at before the NEXT refresh to move
//                       the baseline to the matrix-core product for good.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
#ifdef LT_FD_TRACE      // tools/read_lab/feat_lab.hip: phase stamps of every row wave on the constant 100 MHz clock
__device__ unsigned long long *g_fd_trace = nullptr;
#define FD_STAMP(k_)                                                                                                   \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0 && g_fd_trace)                                                                     \
            g_fd_trace[((size_t)blockIdx.x * FD_WAVES + (threadIdx.x >> 6)) * 8 + (k_)] = wall_clock64();               \
    } while (0)
#else
#define FD_STAMP(k_)
#endif
#define FD_CAP 384
#define FD_WAVES 4
#ifndef FD_PU
#define FD_PU 6          // W1 rows in flight per lane while a list is walked (FD_CAP is a multiple; 8 costs the fifth wave per SIMD: 110 VGPRs
                         // against 90 -- ~10 per row in flight, address pair + data + index -- and 12 the fourth: 138; round 5 re-measured)
#endif
#ifndef FD_MIN_WAVES      // (tools/read_lab/feat_lab builds with 5: its stamps would otherwise cost the kernel its fifth wave per SIMD)
#define FD_MIN_WAVES 1
#endif
#define FD_REF_PAD (16 * 64 * FD_WAVES + 64)   // the staging loop of k_s1d_feature_rows reads the reference vector in whole passes
#define FD_UN 26         // loads in flight per lane in the row pass (x 128 floats: F <= 3328 is one trip)
// VEC: floats per lane and load, 2 when the rows of X are 8-byte aligned, else 1.  (Round 4 built VEC = 4 -- 13 loads of 16 bytes,
// ONE compare step per load with scalar lane masks: 9 VALU per step instead of ~60 -- and removed it: 26.0 against 24.3 us by
// events.  The compare steps are not where the time is: tools/read_lab/feat_lab.hip, profiles/r04_feat_lab_timeline.txt.)
// ONE: the reference vector is staged in one
// pass (F <= 16 * 64 * FD_WAVES = 4096) -- no loop then, which hipcc needs to keep the row's loads in flight across the
// staging (a path through a loop in front of the compare steps makes it wait for everything there).
// One K slice of cref = m W1 (deferred form) by a 256-thread block: block z sums slice [64 z, 64 z + 64) for every hidden column (thread
// (kq, cq): 16 k's x 4 columns, one trip; the four k-quarters added in order through LDS) into slabs[z]; with a gate the block that
// finishes last adds the slices.  Shared by k_s1d_feature_rows and k_s1d_feature_ring (their first nslab blocks): the same bits.
__device__ __forceinline__ void fd_slab_block(unsigned char *fd_smem, int nslab, int F, int H, int Hp, const float *__restrict__ ref,
                                              const float *__restrict__ W1, double *__restrict__ slabs, unsigned *__restrict__ gate,
                                              double *__restrict__ cref_out) {
        double (*s_p)[256] = reinterpret_cast<double (*)[256]>(fd_smem);           // [4][256] (H <= 256, H % 4 == 0)
        const int k0 = blockIdx.x * 64, k1 = min(F, k0 + 64);
        const int kq = threadIdx.x >> 6, cq = threadIdx.x & 63, cc = 4 * cq;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (cc < H) {
            // (two trips of 8 loads: the rows' path below must keep its 5 waves per SIMD, i.e. <= 102 VGPRs for the kernel)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                f32x4 w[8];
                float m[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 16 * kq + 8 * h + u;
                    m[u] = k < k1 ? ref[k] : 0.f;
                    w[u] = k < k1 ? ld4(W1 + (size_t)k * H + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[t] = fma((double)m[u], (double)w[u][t], a[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) s_p[kq][(cc + t) & 255] = a[t];
        __syncthreads();
        if (!gate) {
            for (int c = threadIdx.x; c < H; c += 256) slabs[(size_t)blockIdx.x * H + c] = ((s_p[0][c] + s_p[1][c]) + s_p[2][c]) + s_p[3][c];
            return;
        }
        // gate != NULL: the slab block that finishes last adds the slices (thread (zq, c): slices zq, zq + 4, ... in order, then the
        // four partial sums in order -- a fixed association whoever comes last), so cref is there when this launch ends, without a launch of its own.  The slices travel between the blocks
        // as device-scope atomic stores / loads (they bypass the XCD's L2): a __threadfence() here would write back the whole
        // L2 while the rows stream through it (measured: +8 us).
        for (int c = threadIdx.x; c < H; c += 256)
            __hip_atomic_store(slabs + (size_t)blockIdx.x * H + c, ((s_p[0][c] + s_p[1][c]) + s_p[2][c]) + s_p[3][c], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        unsigned &s_last = *reinterpret_cast<unsigned *>(fd_smem + 4 * 256 * sizeof(double));      // (behind s_p: fd_smem_bytes / fr_smem_bytes leave room)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "this hand-off is written against gfx950's memory system (sc1 write-through stores / sc1 loads, vmcnt counting stores): on another target give the ticket __ATOMIC_RELEASE and the slab loads __ATOMIC_ACQUIRE at agent scope"
#endif
        // (Under the HIP memory model relaxed atomics + s_waitcnt are a data race; what makes it correct is the ISA-level argument
        // below, which is why the translation unit refuses to build for any other target, and why tests/test_gpu_round4.py runs
        // 20 000 refreshes of it against alternating weights under uneven background traffic.)
        // Memory-order argument (MI355X_MICROARCH.md, "Valid forms", sc1 both sides, first table row):
        //  (1) every byte of a slice is stored `sc1` (write-through, above) and loaded `sc1` (below): no L1 / L2 copy to go stale;
        //  (2) each storing wave drains ITS stores (s_waitcnt vmcnt(0): they have reached memory, not just left the wave);
        //  (3) the workgroup barrier puts every wave's drain in front of lane 0's ticket;
        //  (4) the ticket is an agent-scope atomic whose RETURNED value names the last block, and that block's loads come after
        //      the barrier its lane 0 joins once the add has returned.  A workgroup-scope fence here emits nothing (the ticket
        //      could pass a slice still in flight); an agent release would write back the whole L2 under the rows' stream (+8 us).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nslab - 1u ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        const int zq = threadIdx.x >> 6;
        for (int cb = 0; cb < Hp; cb += 64) {
            const int c = cb + (threadIdx.x & 63);
            double acc = 0.0;
            if (c < H)
                for (int z0 = 0; z0 < nslab; z0 += 32) {
                    double t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int z = z0 + zq + 4 * u;
                        t[u] = z < nslab ? __hip_atomic_load(slabs + (size_t)z * H + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += t[u];
                }
            __syncthreads();
            s_p[zq][threadIdx.x & 63] = acc;
            __syncthreads();
            if (zq == 0 && c < Hp)
                cref_out[c] = c < H ? ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x] : 0.0;
        }
        if (threadIdx.x == 0) __hip_atomic_store(gate, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        return;
    }
// max over the wave's 64 lanes (DPP inside a row of 16, the four rows by v_readlane): ~12 VALU instead of six ds_bpermute round trips
__device__ __forceinline__ unsigned fr_wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));     // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));     // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
// the largest of 64 non-negative finite doubles, exactly (their bit patterns order as integers: high words first, then the low
// words of the lanes that hold the largest high word)
__device__ __forceinline__ double fr_wave_max_nonneg(double v) {
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const unsigned mh = fr_wave_max_u32(hi);
    const unsigned ml = fr_wave_max_u32(hi == mh ? lo : 0u);
    return __hiloint2double((int)mh, (int)ml);
}
template <int VEC, bool ONE>
__global__ __launch_bounds__(64 * FD_WAVES, FD_MIN_WAVES) void k_s1d_feature_rows(
    int n, int F, int H, int Hp, const float *__restrict__ X, long ldx, const float *__restrict__ ref,
    const float *__restrict__ W1, const double *__restrict__ cref, double *__restrict__ S1d, int hint_cap,
    int *__restrict__ dense_hint, int nslab, double *__restrict__ slabs, int32_t *__restrict__ zstate,
    float *__restrict__ S1x, unsigned *__restrict__ gate, double *__restrict__ cref_out, double *__restrict__ S1qs,
    const lt_bits_job job = lt_bits_job{}, const int job_first = 0, const int flag_bits = 0, const int stagger = 0) {
    // job.nblocks > 0 (round 5): the blocks from job_first on -- BEHIND the rows in dispatch order, into the CU slots the rows leave
    // free -- are a probe chunk's record blocks or item-table blocks (lt_items.hip.h: nothing in them reads a layer).  They used to ride
    // in the launch that forms the pre-activation and cost it 1.5 us; this launch is seven times longer and bound by the pass over X.
    // zstate != NULL: every row's pre-activation is marked stale here (saves the refresh its memset launch).
    // The first nslab blocks of the launch (deferred cref, nslab > 0) form the K slices of cref = m W1 instead of rows:
    // block z sums slice [64 z, 64 z + 64) for every hidden column (thread (kq, cq): 16 k's x 4 columns, one trip; the four
    // k-quarters added in order through LDS) into slabs[z]; the last of these blocks adds the slices (below), and the rows are
    // written WITHOUT cref (cref == NULL), which the fp64 SpMM and stage A add where they read them.
    extern __shared__ __attribute__((aligned(16))) unsigned char fd_smem[];
    if (job.nblocks > 0 && (int)blockIdx.x >= job_first) {
        item_bits_block((int)blockIdx.x - job_first, job, fd_smem);
        return;
    }
    if ((int)blockIdx.x < nslab) {
        fd_slab_block(fd_smem, nslab, F, H, Hp, ref, W1, slabs, gate, cref_out);
        return;
    }
    // job.zero_blocks > 0 (lt_influence_rows_f64): the next blocks of the launch -- in front of the rows in dispatch order -- fill
    // rows of the caller's float64 matrix with +0.0: np.zeros of attacker.py:216 crossing PCIe for as long as the rows take
    const int nzero = job.zero_blocks;
    if ((int)blockIdx.x < nslab + nzero) {
        zero_rows_wave(job, (int)blockIdx.x - nslab);
        return;
    }
    // stagger (round 6): the row blocks start in `stagger & 255` groups, group k (blocks in launch order) `stagger >> 8` ticks of the
    // 100 MHz clock after group k - 1.  Every row of the launch is one generation of waves: started together, their rows all land
    // at the end of the pass over X and the list walks of ALL rows (20 W1 rows per row of X: 1.6 x the bytes of X through the
    // L1s) queue behind it; started in groups, group k walks while group k + 1's rows are still arriving.
    if ((stagger & 255) > 1) {
        const int nrb = (int)gridDim.x - nslab - nzero - (job.nblocks > 0 ? job.nblocks : 0);
        const int grp = (int)(((long)((int)blockIdx.x - nslab - nzero) * (stagger & 255)) / (nrb > 0 ? nrb : 1));
        const unsigned long long until = wall_clock64() + (unsigned long long)grp * (unsigned)(stagger >> 8);
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
    FD_STAMP(0);
    float *sref = reinterpret_cast<float *>(fd_smem);                              // [Fp] the reference vector
    const int Fp = (F + 1) & ~1;
    double *ldv = reinterpret_cast<double *>(fd_smem + (((size_t)Fp * 4 + 15) & ~(size_t)15));   // [FD_WAVES][FD_CAP]
    int *lj = reinterpret_cast<int *>(ldv + FD_WAVES * FD_CAP);                    // [FD_WAVES][FD_CAP]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i = ((int)blockIdx.x - nslab - nzero) * FD_WAVES + wid;
    const bool live = i < n;                        // (waves past the last row still help staging and join the barrier)
    const float *xr = X + (long)(live ? i : 0) * ldx;
    constexpr int STEP = 64 * VEC;
    // the first trip's loads of the row go out BEFORE the reference vector is staged: one round trip covers both
    // Branch-free loads: hipcc turns a load guarded by a branch into "wait for everything, then load", i.e. it SERIALISES the
    // row's 26 loads (s_waitcnt vmcnt(0) in front of each: the pass took 24 us against 12 us for the same reads issued back
    // to back, tools/read_lab).  Lanes past the end of the row read the row's last pair instead (valid memory) and are
    // masked by `j < F` when the values are compared.
    constexpr int UN = FD_UN;
    float x[UN][VEC];
    // The loads are constant offsets from ONE base address (26 individually clamped addresses cost 52 VGPRs and the kernel
    // its fifth wave per SIMD), so a trip reads T = FD_UN * STEP floats whatever F is: past the end of the row into the
    // following rows (valid memory, masked by `j < F` later).  Near the END of X there is nothing behind: the window of
    // such a trip is shifted left so that it ends with the matrix (it then starts in earlier elements: masked by `j >= j0`).
    // The host takes this route only when X holds at least T floats.
    constexpr int T = UN * STEP;
    const long total_floats = (long)(n - 1) * ldx + F;
    const long off_i = (long)(live ? i : 0) * ldx;
    int shift = 0;                                  // wave-uniform, of the trip in flight (even when VEC == 2)
    auto load_trip = [&](int j0) {
        const long over = off_i + j0 + T - total_floats;
        shift = over > 0 ? (int)over : 0;
        const float *p = xr + j0 + lane * VEC - shift;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if constexpr (VEC == 2) {
                const f32x2_ t = *reinterpret_cast<const f32x2_ *>(p + u * STEP);
                x[u][0] = t.x; x[u][1] = t.y;
            } else {
                x[u][0] = p[u * STEP];
            }
        }
    };
    {
        float r[16];                                            // (16 loads in flight per thread, no branch between them)
#pragma unroll
        for (int u = 0; u < 16; ++u) r[u] = ref[u * 64 * FD_WAVES + tid];   // (ref is allocated FD_REF_PAD floats past F: constant offsets from one address)
        // The row's loads go out BEHIND the reference vector's (loads return in order): the staging below then waits for
        // its own 16 alone, the barrier for nothing, and the compare steps start while the row is still arriving.  Issued
        // in front of them, every wave sat at s_waitcnt vmcnt(0) until its whole row had landed.
        asm volatile("" ::: "memory");      // (keeps hipcc from hoisting the row's loads back in front)
        load_trip(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = u * 64 * FD_WAVES + tid;
            if (j < Fp) sref[j] = j < F ? r[u] : 0.f;
        }
    }
    if constexpr (!ONE) {
        for (int j0 = 16 * 64 * FD_WAVES; j0 < Fp; j0 += 16 * 64 * FD_WAVES) {
            float r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = ref[j0 + u * 64 * FD_WAVES + tid];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + u * 64 * FD_WAVES + tid;
                if (j < Fp) sref[j] = j < F ? r[u] : 0.f;
            }
        }
    }
    __syncthreads();
    FD_STAMP(1);
    if (!live) return;
    double *mv = ldv + wid * FD_CAP;
    int *mj = lj + wid * FD_CAP;
    const int c0 = 4 * lane;
    const bool own = c0 < Hp;                       // this lane holds 4 hidden columns
    const bool vec_ok = (H % 4 == 0) && c0 + 3 < H;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // walk the wave's list (entries 0 .. cnt, padded with zero terms to a multiple of FD_PU)
    auto walk = [&](int cnt) {
        const int padded = (cnt + FD_PU - 1) / FD_PU * FD_PU;
        if (lane < padded - cnt) { mj[cnt + lane] = 0; mv[cnt + lane] = 0.0; }      // (d = 0: the term adds exactly nothing)
        if (own && vec_ok) {
            for (int e = 0; e < padded; e += FD_PU) {
                f32x4 w[FD_PU];
                double d[FD_PU];
#pragma unroll
                for (int k = 0; k < FD_PU; ++k) {       // (no branch between the loads: they all go out before the first wait)
                    d[k] = mv[e + k];
                    w[k] = ld4(W1 + (size_t)mj[e + k] * H + c0);
                }
#pragma unroll
                for (int k = 0; k < FD_PU; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = fma(d[k], (double)w[k][t], acc[t]);
                // (round 6 re-measured the trip depth with __launch_bounds__(256, 5) holding the fifth wave: 6 / 8 / 12 rows in flight,
                // the differences read when used: 22.6 / 22.4 / 22.5 us -- the walk's trips are not what bounds the launch)
            }
        } else if (own) {
            for (int e = 0; e < padded; ++e) {
                const int j = mj[e];
                const double d = mv[e];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = fma(d, c0 + t < H ? (double)W1[(size_t)j * H + c0 + t] : 0.0, acc[t]);
            }
        }
    };
    const unsigned long long lt = (1ull << lane) - 1ull;
    // pass 1 (the common case is all there is); differing columns beyond the list's capacity are only counted
    int total = 0;                                  // wave-uniform: differing columns of the row
    bool listed = false;
    if constexpr (VEC == 2) {
        if (flag_bits && F <= T && shift == 0) {
            // (round 6) the 52 ballot steps as plain VALU: one bit per value (xor, min, shift-or into four accumulators), then the
            // flagged values appended level by level -- the lane's three lowest in one trip, read again from the row (an L2 hit; a
            // register array cannot be indexed by a lane's own bit number) -- order (level, lane).  A ballot + scalar branch per
            // value is ~10 dependent VALU -> SALU hops of 10-20 cycles each (tools/read_lab/ring_lab: 3.6 us of a wave's row).
            unsigned fl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const f32x2_ r2 = *reinterpret_cast<const f32x2_ *>(sref + u * STEP + 2 * lane);
                const float rr[2] = {r2.x, r2.y};
                const bool inside = (u + 1) * STEP <= F;          // (wave-uniform)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    unsigned t = min(__float_as_uint(x[u][v]) ^ __float_as_uint(rr[v]), 1u);
                    if (!inside) t = u * STEP + 2 * lane + v < F ? t : 0u;
                    const int bit = 2 * u + v;
                    fl[(bit >> 5) * 2 + v] |= t << (bit & 31);
                }
            }
            unsigned long long flags = ((unsigned long long)(fl[2] | fl[3]) << 32) | (unsigned long long)(fl[0] | fl[1]);
            FD_STAMP(2);
            FD_STAMP(3);
            {
                int jq[3];
                float xq[3], rq[3];
                bool has[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    has[t] = flags != 0ull;
                    const int bit = has[t] ? __ffsll((long long)flags) - 1 : 0;
                    flags &= flags - 1ull;
                    jq[t] = (bit >> 1) * STEP + 2 * lane + (bit & 1);
                    xq[t] = xr[min(jq[t], F - 1)];
                    rq[t] = sref[jq[t]];
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const unsigned long long m = __ballot(has[t]);
                    const int pos = total + __popcll(m & lt);
                    if (has[t] && pos < FD_CAP) { mj[pos] = jq[t]; mv[pos] = (double)xq[t] - (double)rq[t]; }
                    total += __popcll(m);
                }
            }
            while (__ballot(flags != 0ull)) {
                const bool has = flags != 0ull;
                const unsigned long long m = __ballot(has);
                if (has) {
                    const int bit = __ffsll((long long)flags) - 1;
                    flags &= flags - 1ull;
                    const int j = (bit >> 1) * STEP + 2 * lane + (bit & 1);
                    const float xq = xr[j], rq = sref[j];
                    const int pos = total + __popcll(m & lt);
                    if (pos < FD_CAP) { mj[pos] = j; mv[pos] = (double)xq - (double)rq; }
                }
                total += __popcll(m);
            }
            listed = true;
        }
    }
    for (int j0 = 0; j0 < F && !listed; j0 += STEP * UN) {
        if (j0 > 0) load_trip(j0);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int j = j0 + u * STEP + lane * VEC + v - shift;
                const float r = (j >= j0 && j < F) ? sref[j] : 0.f;
                const bool diff = j >= j0 && j < F && x[u][v] != r;
                const unsigned long long m = __ballot(diff);
                if (m) {
                    const int pos = total + __popcll(m & lt);
                    if (diff && pos < FD_CAP) { mj[pos] = j; mv[pos] = (double)x[u][v] - (double)r; }
                    total += __popcll(m);
                }
            }
            if (u == 0) FD_STAMP(2);
            if (u == UN / 2) FD_STAMP(3);
        }
    }
    FD_STAMP(4);
    if (total <= FD_CAP) {
        walk(total);
    } else {
        // a dense row: what pass 1 listed is incomplete, so the row is read again, one piece at a time, and every piece's
        // list is walked before the next is made (slow and correct; the hint below moves the baseline off this route)
        for (int j0 = 0; j0 < F; j0 += 64) {
            const int j = j0 + lane;
            const float xv = j < F ? xr[j] : 0.f;
            const float r = j < F ? sref[j] : 0.f;
            const bool diff = j < F && xv != r;
            const unsigned long long m = __ballot(diff);
            if (diff) { const int pos = __popcll(m & lt); mj[pos] = j; mv[pos] = (double)xv - (double)r; }
            walk(__popcll(m));
        }
    }
    FD_STAMP(5);
    if (total > hint_cap && lane == 0) *dense_hint = 1;
    if (zstate && lane == 0) zstate[i] = 0;
    if (!own && !S1x) return;
    f64x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = (own && c0 + t < H) ? (cref ? cref[c0 + t] + acc[t] : acc[t]) : 0.0;
    if (S1x) {      // (every lane of the wave is here: the row's largest value is a wave reduction)
        // 32-bit FIXED POINT with one scale per row: q = round(value / scale), scale = (largest |value| of the row) / 2^31 -- 31 bits
        // against the row's largest value instead of fp32's 24 against each value, in the same 4 bytes.  What the readers sum is
        // A_hat[r, c] * value: an ABSOLUTE error per term is what reaches a pre-activation, and plain fp32 rows reached the result
        // as up to 7e-5 of the largest score where a hidden unit sat within dz of its kink (tools/fuzz_gpu.py 60 31337).
        // (the wave's maximum by DPP on the two 32-bit halves, as the ring kernel takes it: six xor-shuffles of a double are twelve
        // ds_bpermute round trips in every wave's tail)
        const double mx = fr_wave_max_nonneg(fmax(fmax(fabs(o[0]), fabs(o[1])), fmax(fabs(o[2]), fabs(o[3]))));
        const double scale = mx > 0.0 ? mx * (1.0 / 2147483000.0) : 1.0;
        const double inv = 1.0 / scale;
        int q[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = (int)rint(o[t] * inv);
        if (own) *reinterpret_cast<int4 *>(S1x + (size_t)i * Hp + c0) = make_int4(q[0], q[1], q[2], q[3]);
        if (lane == 0) S1qs[i] = scale;
    }
    else if (own) *reinterpret_cast<f64x4 *>(S1d + (size_t)i * Hp + c0) = o;
    FD_STAMP(6);
}
static size_t fd_smem_bytes(int F) {
    const size_t Fp = (size_t)((F + 1) & ~1);
    const size_t need = ((Fp * 4 + 15) & ~(size_t)15) + (size_t)FD_WAVES * FD_CAP * (sizeof(double) + sizeof(int));
    return need < 8192 + 16 ? 8192 + 16 : need;      // (the slab blocks of the deferred-cref launch use 4 x 256 doubles + a word of it)
}
// rows with more differing columns than this are "dense" for the route decision: the list walk costs ~ cnt * H fp64 FMAs and
// cnt row gathers per node, the matrix cores F * H at ~10 x the rate
static int fd_hint_cap(int F) { const int c = F / 16; return c < 8 ? 8 : (c > FD_CAP ? FD_CAP : c); }

#include "lt_feature_ring.hip.h"
static size_t fd_slab_doubles(int F, int H) { return (size_t)((F + 63) / 64) * H; }      // the K slices of m W1
static int fd_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev] = c;
    }
    return cus[dev];
}
// Whether the persistent ring form serves this product (else the row-per-wave kernel): rows 8-byte aligned, four hidden columns per
// lane, 9 .. 13 chunks of 1 KiB per row (two workgroups' rings + reference vector + lists inside one CU's LDS), enough rows
static bool fd_ring_ok(const lt_baseline *b, int n) {
    const int knob = lt_tune().feature_ring;
    if (knob == 0) return false;
    const int nch = fr_chunks(b->F);
    return b->H % 4 == 0 && b->H <= 256 && b->Hp == b->H && b->ldx % 2 == 0 && ((uintptr_t)b->X % 8) == 0 && nch >= FR_NCH_MIN &&
           nch <= FR_NCH_MAX && fr_smem_bytes(nch) <= (size_t)FR_LDS_MAX && n >= (knob > 0 ? 2 : lt_tune().feature_ring_min_rows) &&
           b->ldx >= 260;
}
// every block resident from the start (two per CU); the first nsl of them form a slice of m W1 before they take rows
static unsigned fd_ring_grid(int n, int nsl) {
    const int cap = 2 * fd_cu_count(), want = nsl + (n + FR_WAVES - 1) / FR_WAVES;
    return (unsigned)(want < cap ? want : (cap > nsl ? cap : nsl + 1));
}
static int fd_ring_allow_lds() {
    static unsigned long long done = 0ull;
    int dev = 0;
    LT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
#define LT_FR_ATTR(N_) LT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_s1d_feature_ring<N_>), hipFuncAttributeMaxDynamicSharedMemorySize, FR_LDS_MAX))
        LT_FR_ATTR(9); LT_FR_ATTR(10); LT_FR_ATTR(11); LT_FR_ATTR(12); LT_FR_ATTR(13);
#undef LT_FR_ATTR
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
    return LT_OK;
}

// The reference vector m[k] = the more frequent of (min, max) of X[0 .. rows, k] over the first rows <= 64 rows: the
// majority value of a two-valued column.  Computed ONCE (lt_baseline_enable_fp64): any m is a correct reference, so a
// later change of X costs speed at worst (and the dense hint then retires the route).
__global__ __launch_bounds__(64) void k_ref_vector(int n, int F, const float *__restrict__ X, long ldx, float *__restrict__ ref) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= F) return;
    const int rows = n < 64 ? n : 64;
    float mn = 3.4e38f, mx = -3.4e38f;
    for (int r = 0; r < rows; ++r) { const float v = X[(long)r * ldx + k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    int c = 0;
    for (int r = 0; r < rows; ++r) c += X[(long)r * ldx + k] == mn ? 1 : 0;
    ref[k] = 2 * c >= rows ? mn : mx;
}

// cref = m W1 in fp64, one launch per refresh.  Block z owns the 64-deep K slice [64 z, 64 z + 64); thread (kq, cq) sums 16 of
// its k's for the 4 columns 4 cq .. 4 cq + 3 (one trip of 16 float4 loads), the four k-quarters are added in order through
// LDS into slabs[z]; the block that finishes last adds the slices in slice order the same way (a fixed order whoever comes
// last).  H % 4 != 0 takes the column-per-thread form.
__global__ __launch_bounds__(256) void k_ref_product(int F, int H, int Hp, const float *__restrict__ ref,
                                                     const float *__restrict__ W1, double *__restrict__ slabs,
                                                     double *__restrict__ cref, unsigned *__restrict__ counter) {
    __shared__ unsigned s_last;
    __shared__ double s_p[4][256];
    const int k0 = blockIdx.x * 64, k1 = min(F, k0 + 64);
    const unsigned nz = gridDim.x;
    if (H % 4 == 0 && H <= 256) {
        const int kq = threadIdx.x >> 6, cq = threadIdx.x & 63, c0 = 4 * cq;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (c0 < H) {
            f32x4 w[16];
            float m[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 16 * kq + u;
                m[u] = k < k1 ? ref[k] : 0.f;
                w[u] = k < k1 ? ld4(W1 + (size_t)k * H + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = fma((double)m[u], (double)w[u][t], a[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) s_p[kq][(c0 + t) & 255] = a[t];
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256) slabs[(size_t)blockIdx.x * H + c] = ((s_p[0][c] + s_p[1][c]) + s_p[2][c]) + s_p[3][c];
    } else {
        for (int c = threadIdx.x; c < H; c += 256) {
            double a = 0.0;
            for (int k = k0; k < k1; ++k) a = fma((double)ref[k], (double)W1[(size_t)k * H + c], a);
            slabs[(size_t)blockIdx.x * H + c] = a;
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(counter, 1u) == nz - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (H <= 256) {
        // thread (zq, c): the slices z = zq, zq + 4, ... of column c in one trip when there are <= 64 of them; the four partial
        // sums are added in order -- a fixed association of the slices whoever runs this
        const int zq = threadIdx.x >> 6;
        for (int cb = 0; cb < H; cb += 64) {
            const int c = cb + (threadIdx.x & 63);
            double a = 0.0;
            if (c < H)
                for (unsigned z0 = 0; z0 < nz; z0 += 64) {
                    double t[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const unsigned z = z0 + zq + 4u * u;
                        t[u] = z < nz ? __builtin_nontemporal_load(slabs + (size_t)z * H + c) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) a += t[u];
                }
            __syncthreads();
            s_p[zq][threadIdx.x & 63] = a;
            __syncthreads();
            if (zq == 0 && c < H) cref[c] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
        }
        for (int c = H + threadIdx.x; c < Hp; c += 256) cref[c] = 0.0;
    } else {
        for (int c = threadIdx.x; c < Hp; c += 256) {
            double a = 0.0;
            if (c < H)
                for (unsigned z = 0; z < nz; ++z) a += slabs[(size_t)z * H + c];
            cref[c] = a;
        }
    }
    if (threadIdx.x == 0) *counter = 0;              // ready for the next launch
}

// rs[r] = sum of row r of A_hat, fp64 (entry order)
__global__ void k_row_sums(int n, const int32_t *__restrict__ rowptr, const float *__restrict__ val, double *__restrict__ rs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double a = 0.0;
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) a += (double)val[e];
    rs[r] = a;
}

// K slice of the fp64 product (64x64 tiles, 4 waves, 8 workgroups per CU): the fewest slices (at least 400 deep) whose
// workgroups still occupy every CU, and never more workgroups than are resident at once (256 CUs x 8) -- a ninth per CU
// runs alone after the others (twitch-RU: 8 slices = 2208 workgroups; 7 slices = 1932, all resident; measured 134-137 us
// for 6 ... 8 slices alike, the 64x64 kernel is not sensitive to it).
static bool fp64_big(int n, int H) { return n >= 1024 && H % GE_BN == 0; }
static int fp64_kslice(int n, int H, int F) {
    if (fp64_big(n, H)) {
        // 128x128 tiles, two workgroups per CU: the slicing rule of the f32 product (whole rounds of 256 CUs)
        const long tiles = (long)((n + GE_BM - 1) / GE_BM) * (H / GE_BN);
        int best = (F + 15) / 16 * 16;
        double best_cost = 1e30;
        for (int s2 = 1; s2 <= 16; ++s2) {
            const int ks = ((F + s2 - 1) / s2 + 15) / 16 * 16;
            const int slices = (F + ks - 1) / ks;
            const long rounds = (tiles * slices + 255) / 256;
            const double cost = (double)rounds * (ks / 16) + 2.2 * (slices - 1);
            if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
        }
        return best;
    }
    const long tiles = (long)((n + GD_BM - 1) / GD_BM) * ((H + GD_BN - 1) / GD_BN);
    int best = (F + 15) / 16 * 16;
    for (int s = 2; s <= 64; ++s) {
        const int ks = ((F + s - 1) / s + 15) / 16 * 16;
        if (ks < 400 && tiles * (s - 1) >= 512) break;
        if (tiles * ((F + ks - 1) / ks) > 2048) break;
        best = ks;
    }
    return best > 0 ? best : 16;
}

// The dense product S1d = X*W1 on the f64 matrix cores: rows [r0, r1) into dst[(r1 - r0), ldd].
// quant: the whole product (r0 = 0, r1 = n) leaves as fixed-point rows in b->S1x / b->S1qs instead of dst (needs the split-K form)
// (a rule of the SHAPES and the "s1_f32" knob only: every rank of a sharded run and a single-GPU run decide alike)
static bool dense_quant_shapes(const lt_baseline *b) {
    return (long long)b->n * b->Hp * (long long)sizeof(double) < ((long long)32 << 20) && b->H <= 256 && b->H % 4 == 0;
}
static bool dense_quant_possible(const lt_baseline *b) {
    return dense_quant_shapes(b) && lt_tune().s1_f32 != 0 && b->S1x && b->S1qs;
}
static int quant_rows(lt_baseline *b, const double *S, hipStream_t st) {
    hipLaunchKernelGGL(k_quant_rows_f64, dim3((unsigned)((b->n + 3) / 4)), dim3(256), 0, st, S, (long)b->Hp, b->n, b->H, b->Hp, b->S1x, b->S1qs);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
#include "lt_i8_split.hip.h"
// "i8_split": the product on the int8 matrix cores as an error-free split (lt_i8_split.hip.h: X as five, W1 as four signed base-256
// digits, the fourteen digit pairs of order >= 3, exact integer sums per order, one rounding per K slice) when the baseline holds
// the digit buffers (shapes lt_i8_shapes_ok) -- rows within 5e-10 of the row's largest value of the fp64 product, the error of the
// 32-bit fixed-point rows they are stored as
static bool i8_route(const lt_baseline *b) {
    return lt_tune().i8_split != 0 && b->i8_wd != nullptr && b->i8_ew != nullptr && lt_i8_shapes_ok(b->n, b->H, b->F);
}
static int launch_dense_s1d(lt_baseline *b, int r0, int r1, double *dst, long ldd_out, hipStream_t st, bool quant = false,
                            int32_t *zstate_rows = nullptr) {      // zstate_rows: cleared by the slab sum (rows [r0, r1) = all rows)
    const int H = b->H, n = b->n, F = b->F, m = r1 - r0;
    if (m <= 0) return LT_OK;
    // the K slicing is that of the FULL product whatever the row range: a row has the same bits whichever rank computed it
    const int kslice = fp64_kslice(n, H, F);
    const bool i8 = i8_route(b) && b->slabs_d != nullptr;
    const int splits = i8 ? lt_i8_slices(n, H, F) : (F + kslice - 1) / kslice;
    const bool big = fp64_big(n, H);
    dim3 grid(big ? (m + GE_BM - 1) / GE_BM : (m + GD_BM - 1) / GD_BM, big ? H / GE_BN : (H + GD_BN - 1) / GD_BN, splits);
    double *out = (splits > 1 || i8) ? b->slabs_d : dst;
    const long ldd = splits > 1 ? (long)H : ldd_out;
    const long stride = splits > 1 ? (long)m * H : 0L;
    const float *A = b->X + (size_t)r0 * b->ldx;
    if (i8) {
        // W1's digits (it may have changed since the last refresh), then this row range's slabs: one fp64 partial per K slice
        // (the exponent words are zero already when the last product's slab sum cleared them)
        const bool clean = b->i8_ew_clean;
        b->i8_ew_clean = false;
        int rc = lt_launch_i8_w_digits(b->W1, n, F, H, b->i8_wd, b->i8_ew, st, !clean);
        if (rc) return rc;
        rc = lt_launch_gemm_i8split<3>(A, (long)b->ldx, m, n, F, H, b->i8_wd, b->i8_ew, b->slabs_d, st);
        if (rc) return rc;
        if (splits == 1) {      // (one slice: the slab is the product; the sums below expect at least two)
            const long tot = (long)m * H;
            hipLaunchKernelGGL(k_sum_slabs_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b->slabs_d, tot, 1, tot, H, dst, ldd_out);
            LT_CHECK_LAUNCH();
            return quant ? quant_rows(b, dst, st) : LT_OK;
        }
    } else if (big)
        hipLaunchKernelGGL(k_gemm_f64acc_128, grid, dim3(256), 0, st, A, (long)b->ldx, b->W1, (long)H, out, ldd, m, H, F,
                           splits > 1 ? kslice : (F > 0 ? F : 1), stride);
    else
        hipLaunchKernelGGL(k_gemm_f64acc, grid, dim3(256), 0, st, A, (long)b->ldx, b->W1, (long)H, out, ldd, m, H, F,
                           splits > 1 ? kslice : (F > 0 ? F : 1), stride);
    LT_CHECK_LAUNCH();
    if (splits > 1 && quant && b->slabs_d) {
        hipLaunchKernelGGL(k_sum_slabs_f64_q, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, b->slabs_d, (long)m * H, splits, m, H, b->Hp,
                           b->S1x, b->S1qs, zstate_rows, i8 ? b->i8_ew : (unsigned *)nullptr,
                           i8 ? (int)(lt_i8_ew_bytes(n, H, F) / sizeof(unsigned)) : 0);
        LT_CHECK_LAUNCH();
        if (i8) b->i8_ew_clean = true;
    } else if (splits > 1) {
        const long tot = (long)m * H;
        hipLaunchKernelGGL(k_sum_slabs_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b->slabs_d,
                           tot, splits, tot, H, dst, ldd_out);
        LT_CHECK_LAUNCH();
    } else if (quant) {       // one K slice: the rows are in dst already
        return quant_rows(b, dst, st);
    }
    return LT_OK;
}

// ---- aggregate-first route: Z1d[r] = (A_hat X)[r] * W1 + b1 on the rows a call's probes reach ------------------------
// LT_MODE_DELTA reads the fp64 pre-activation only on its items' rows r (the union of R_v over the probes) and the fp64
// product only on the probes' own rows.  For features no wider than ~2 H (BASELINE configs[4]: F = H = 256) it is cheaper
// to aggregate first -- Y[r] = sum_c A_hat[r,c] X[c], fp32 gathers of F columns, fp64 chains, ONLY for the rows needed --
// and to multiply those few rows by W1 afterwards, than to form S1d = X W1 for all n rows (2 n F H flops) and gather
// 8-byte S1d rows for every row of the graph: R-MAT scale 21, 512 probes: 15 K of 2 M rows are needed.
//   k_z_mark            items (b, r) -> rows with no valid pre-activation since the last refresh: zstate 0 -> 2, listed
//   k_rows_tiled_xf64   (lt_spmm.hip) the work items of the marked rows; k_y_long adds the segment sums of marked hub rows
//   k_gemm_f64_rows     Z1d[rows[i]] = Y[rows[i]] W1 + b1 (f64 matrix cores, rows through the list, count on the device),
//                       zstate -> 1; the same kernel with fp32 A forms the probes' product rows Spd = X[probes] W1
static __global__ __launch_bounds__(256) void k_z_mark(const int32_t *__restrict__ off, int nb, const int2 *__restrict__ item_pr,
                                                       int32_t *__restrict__ zstate, int32_t *__restrict__ zrows,
                                                       int32_t *__restrict__ zcount) {
    const int total = off[nb];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = item_pr[i].y;
        if (zstate[r] != 0) continue;                       // valid already, or claimed by another item of this chunk
        if (atomicCAS(&zstate[r], 0, 2) == 0) zrows[atomicAdd(zcount, 1)] = r;   // (list order is arbitrary: rows are independent)
    }
}
static __global__ void k_y_long(int n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_segptr,
                                const double *__restrict__ part, int ld, const int32_t *__restrict__ zstate,
                                double *__restrict__ Y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_long * ld) return;
    const int li = (int)(i / ld), c = (int)(i % ld);
    const int r = long_row[li];
    if (zstate[r] != 2) return;
    // (segment order, as everywhere; eight loads in flight -- one dependent load per segment made a 900-segment hub row the
    // whole launch: 0.26 ms at BASELINE configs[4])
    const int s0 = long_segptr[li], s1 = long_segptr[li + 1];
    double acc = part[(size_t)s0 * ld + c];
    int sg = s0 + 1;
    for (; sg + 8 <= s1; sg += 8) {
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = part[(size_t)(sg + t) * ld + c];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc += v[t];
    }
    for (; sg < s1; ++sg) acc += part[(size_t)sg * ld + c];
    Y[(size_t)r * ld + c] = acc;
}

// C[row(i), 0..N) = A[rows[i], 0..K) * B[K, N] (+ bias), i < M, on v_mfma_f64_16x16x4_f64: 64 x 64 tiles, a fixed grid
// walking the tiles (M may live on the device).  AT = double (the aggregated rows) or float (feature rows, widened).
// scatter: 1 = row(i) = rows[i] (results land at the node's row), 0 = row(i) = i.  state != NULL: state[rows[i]] = 1.
// PF: k-steps whose global loads go out together (1: a step's loads, then its MFMAs -- every step pays a round trip; 4 for launches of
// few tiles, where nothing else hides it: GCN3's S2d = relu(Z1d) W2 is 69 tiles on 256 CUs, 33.5 -> 27.7 us; a wave per row with W2 in
// LDS and v_readlane broadcasts measured 48 us: instruction-bound).  Same accumulation order.
template <typename AT, int PF = 1>
__global__ __launch_bounds__(256) void k_gemm_f64_rows(const AT *__restrict__ A, long lda, const int32_t *__restrict__ rows,
                                                       const int32_t *__restrict__ m_dev, int m_host,
                                                       const float *__restrict__ B, long ldb, int N, int K,
                                                       const float *__restrict__ bias, double *__restrict__ C, long ldc,
                                                       int scatter, int32_t *__restrict__ state, int relu_a) {
    // rows == NULL: row(i) = i.  relu_a: A is read through max(., 0) (the hidden activation of a pre-activation matrix)
    __shared__ __attribute__((aligned(16))) double As[GD_BM * GD_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[GD_BK * GD_LDB];
    const int M = m_dev ? *m_dev : m_host;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int ntn = (N + GD_BN - 1) / GD_BN;
    const long tiles = (long)((M + GD_BM - 1) / GD_BM) * ntn;
    const int a_row = tid >> 2, a_col = (tid & 3) * 4;
    const int b_row = tid >> 4, b_col = (tid & 15) * 4;
    const int a_frag = (wr * 32 + (lane & 15)) * GD_LDA + (lane >> 4);
    const int b_frag = (lane >> 4) * GD_LDB + wc * 32 + (lane & 15);
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = (int)(tile / ntn) * GD_BM, n0 = (int)(tile % ntn) * GD_BN;
        const bool a_ok = m0 + a_row < M;
        const AT *ap = A + (size_t)(a_ok ? (rows ? rows[m0 + a_row] : m0 + a_row) : 0) * lda;
        f64x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int k00 = 0; k00 < K; k00 += GD_BK * PF) {
            double ra[PF][4];
            float rb[PF][4];
#pragma unroll
            for (int s_ = 0; s_ < PF; ++s_) {
                const int k0 = k00 + s_ * GD_BK;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ra[s_][j] = 0.0; rb[s_][j] = 0.f; }
                if (a_ok)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + a_col + j < K) { const double v = (double)ap[k0 + a_col + j]; ra[s_][j] = (relu_a && v < 0.0) ? 0.0 : v; }
                if (k0 + b_row < K)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n0 + b_col + j < N) rb[s_][j] = B[(size_t)(k0 + b_row) * ldb + n0 + b_col + j];
            }
#pragma unroll
            for (int s_ = 0; s_ < PF; ++s_) {
                if (k00 + s_ * GD_BK >= K) break;      // (block-uniform)
                __syncthreads();   // the previous k-step's fragment reads are done
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    As[a_row * GD_LDA + a_col + j] = ra[s_][j];
                    Bs[b_row * GD_LDB + b_col + j] = rb[s_][j];
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < GD_BK; kk += 4) {
                    const double a0 = As[a_frag + kk], a1 = As[a_frag + 16 * GD_LDA + kk];
                    const double b0 = (double)Bs[b_frag + kk * GD_LDB], b1 = (double)Bs[b_frag + kk * GD_LDB + 16];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
        // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cn = n0 + wc * 32 + j * 16 + (lane & 15);
                if (cn >= N) continue;
                const double bv = bias ? (double)bias[cn] : 0.0;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int cm = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * reg;
                    if (cm < M) C[(size_t)((scatter && rows) ? rows[cm] : cm) * ldc + cn] = acc[i][j][reg] + bv;
                }
            }
        if (state && n0 == 0 && tid < GD_BM && m0 + tid < M) state[rows[m0 + tid]] = 1;
        __syncthreads();
    }
}

static bool agg_shapes_ok(const lt_baseline *b) {
    return !b->no_agg && b->n > 0 && b->F <= 2 * b->Hp && lt_round_up(b->F, 4) <= 512 && b->g->w_n > 0;
}
bool lt_fp64_agg_active(const lt_baseline *b) {
    if (!b->Z1d || !b->Yd) return false;
    const int knob = lt_tune().aggregate_first;
    if (knob == 0) return false;
    if (knob > 0) return true;
    return b->agg_default;
}

// the rows marked 2 (listed in zrows / zcount): (A_hat X)[r] by the tiled gathers, then Z1d[r] = Y[r] W1 + b1, zstate -> 1
static int agg_form_marked(const lt_baseline *b, hipStream_t st) {
    const lt_graph *g = b->g;
    const int Hp = b->Hp, H = b->H, F = b->F, Fp = b->Fp;
    { lt_prof_scope prof_(LT_K_FP64_SPMM, st);
    int rc = lt_launch_rows_tiled_xf64(g, b->X, b->ldx, F, b->Yd, Fp, b->seg_y, Fp, b->zstate, b->zitems, b->zicount, st);
    if (rc) return rc;
    if (g->p_n_long > 0) {
        const long tot = (long)g->p_n_long * Fp;
        hipLaunchKernelGGL(k_y_long, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g->p_n_long, g->p_long_row,
                           g->p_long_segptr, b->seg_y, Fp, b->zstate, b->Yd);
        LT_CHECK_LAUNCH();
    } }
    lt_prof_scope prof_(LT_K_FP64_PRODUCT, st);
    // (Yd rows hold F valid columns; K = F, so the pad columns of Yd are never read)
    hipLaunchKernelGGL((k_gemm_f64_rows<double>), dim3(2048), dim3(256), 0, st, b->Yd, (long)Fp, b->zrows, b->zcount, 0, b->W1,
                       (long)H, H, F, b->b1, b->Z1d, (long)Hp, 1, b->zstate, 0);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

int lt_fp64_prepare_items(const lt_baseline *b, const int32_t *off, int nb, const int2 *item_pr, const int32_t *probes,
                          double *Spd, hipStream_t st) {
    const int Hp = b->Hp, H = b->H, F = b->F;
    { lt_prof_scope prof_(LT_K_FP64_SPMM, st);
    LT_HIP(hipMemsetAsync(b->zcount, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_z_mark, dim3(1024), dim3(256), 0, st, off, nb, item_pr, b->zstate, b->zrows, b->zcount);
    LT_CHECK_LAUNCH(); }
    { const int rc = agg_form_marked(b, st);
      if (rc) return rc; }
    lt_prof_scope prof_(LT_K_FP64_PRODUCT, st);
    if (Hp != H) LT_HIP(hipMemsetAsync(Spd, 0, (size_t)nb * Hp * sizeof(double), st));
    const int tiles = ((nb + GD_BM - 1) / GD_BM) * ((H + GD_BN - 1) / GD_BN);
    if (tiles < 512)       // (few tiles: four k-steps' loads in flight, see k_gemm_f64_rows)
        hipLaunchKernelGGL((k_gemm_f64_rows<float, 4>), dim3((unsigned)tiles), dim3(256), 0, st, b->X, (long)b->ldx,
                           probes, (const int32_t *)nullptr, nb, b->W1, (long)H, H, F, (const float *)nullptr, Spd, (long)Hp, 0,
                           (int32_t *)nullptr, 0);
    else
        hipLaunchKernelGGL((k_gemm_f64_rows<float>), dim3((unsigned)(tiles < 2048 ? tiles : 2048)), dim3(256), 0, st, b->X, (long)b->ldx,
                           probes, (const int32_t *)nullptr, nb, b->W1, (long)H, H, F, (const float *)nullptr, Spd, (long)Hp, 0,
                           (int32_t *)nullptr, 0);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// Whether this refresh takes the feature-difference product (k_s1d_feature_rows): the "feature_delta" knob, else what the
// probe of lt_baseline_enable_fp64 found -- corrected by the hint word the kernel sets when it meets dense rows (the
// refresh that met them was still served correctly by the kernel, only slowly; from the next one on the matrix cores run).
static bool want_feature_rows(const lt_baseline *cb) {
    lt_baseline *b = const_cast<lt_baseline *>(cb);
    if (!b->fd_cref || !b->S1d || b->n < 2) return false;
    if (b->fd_hint_host && *(volatile int *)b->fd_hint_host != 0) b->feat_sparse = 0;
    const int knob = lt_tune().feature_delta;
    return knob == 0 ? false : (knob > 0 ? true : b->feat_sparse != 0);
}

// A probe chunk's record blocks may ride in the rows' launch (influence_rows_impl offers them before it asks for the baseline's
// fp64 parts; whether they went along is read back afterwards -- the launch only happens when the product is stale)
static thread_local const lt_bits_job *g_offered_job = nullptr;
static thread_local bool g_offered_rode = false;
void lt_fp64_offer_job(const lt_bits_job *job) { g_offered_job = job; g_offered_rode = false; }
bool lt_fp64_offer_taken() { const bool r = g_offered_rode; g_offered_job = nullptr; g_offered_rode = false; return r; }

static int launch_feature_s1d(lt_baseline *b, hipStream_t st, int n_rows = -1, bool defer = false, int32_t *zstate = nullptr) {
    // (fp32 storage of the fp64-accumulated rows goes with the deferred cref: the small-graph route whose readers know about both)
    const int Hp = b->Hp, H = b->H, n = n_rows < 0 ? b->n : n_rows, F = b->F;
    if (!b->fd_ref_valid) {       // once: any reference vector is correct, a good one makes the rows' lists short
        hipLaunchKernelGGL(k_ref_vector, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, st, n, F, b->X, (long)b->ldx, b->fd_ref);
        LT_CHECK_LAUNCH();
        b->fd_ref_valid = true;
    }
    const int nz = (F + 63) / 64;
    // deferred cref: the slices of m W1 ride in the rows' own launch, one block adds them afterwards, and the readers of
    // S1d add cref themselves (H a multiple of 4 and <= 256: the slab blocks' layout)
    defer = defer && H % 4 == 0 && H <= 256 && b->fd_rs != nullptr;
    if (!defer) {
        hipLaunchKernelGGL(k_ref_product, dim3((unsigned)nz), dim3(256), 0, st, F, H, Hp, b->fd_ref, b->W1, b->fd_slabs,
                           b->fd_cref, (unsigned *)b->fd_gate);
        LT_CHECK_LAUNCH();
    }
    const double *cref = defer ? (const double *)nullptr : b->fd_cref;
    float *s1x = (defer && b->S1x && b->S1qs && lt_tune().s1_f32 != 0) ? b->S1x : nullptr;
    if (fd_ring_ok(b, n)) {
        // the persistent ring form (lt_feature_ring.hip.h): the slab blocks in front, then two workgroups per CU; a probe chunk's
        // record blocks do not ride here (the workgroups take the CU's LDS between them) -- they go with the pre-activation's
        // launch as with "records_early" = 0
        int rc = fd_ring_allow_lds();
        if (rc) return rc;
        const int nch = fr_chunks(F), nsl = defer ? nz : 0;
        const unsigned grid = fd_ring_grid(n, nsl);
#define LT_FR_LAUNCH(N_)                                                                                                      \
    hipLaunchKernelGGL(k_s1d_feature_ring<N_>, dim3(grid), dim3(64 * FR_WAVES), fr_smem_bytes(nch), st, n, F, H, b->X, (long)b->ldx,   \
                       b->fd_ref, b->W1, cref, b->S1d, fd_hint_cap(F) < FR_CAP ? fd_hint_cap(F) : FR_CAP, b->fd_hint_dev, nsl, b->fd_slabs, \
                       zstate, s1x, (unsigned *)b->fd_gate, b->fd_cref, b->S1qs, parity)
        const int parity = b->fd_ring_parity;
        b->fd_ring_parity ^= 1;
        switch (nch) {
        case 9: LT_FR_LAUNCH(9); break;
        case 10: LT_FR_LAUNCH(10); break;
        case 11: LT_FR_LAUNCH(11); break;
        case 12: LT_FR_LAUNCH(12); break;
        default: LT_FR_LAUNCH(13); break;      // (F = 3170, utils/load.py:56: the twitch loader's width)
        }
#undef LT_FR_LAUNCH
        LT_CHECK_LAUNCH();
        b->cref_deferred = defer;
        b->s1_f32 = s1x != nullptr;
        return LT_OK;
    }
    const int nslab = defer ? nz : 0;
    unsigned blocks = (unsigned)((n + FD_WAVES - 1) / FD_WAVES + nslab);
    const size_t smem = fd_smem_bytes(F);
    lt_bits_job jb = lt_bits_job{};
    int job_first = (int)blocks;
    if (g_offered_job && n_rows < 0 && g_offered_job->nblocks > 0 && g_offered_job->smem_bytes <= smem && lt_tune().records_early != 0) {
        jb = *g_offered_job;
        job_first += jb.zero_blocks;          // (the zero-fill blocks sit between the slab blocks and the rows)
        blocks += (unsigned)(jb.nblocks + jb.zero_blocks);
        g_offered_rode = true;
    }
    const bool vec2 = b->ldx % 2 == 0 && F % 2 == 0 && ((uintptr_t)b->X % 8) == 0, one = F <= 16 * 64 * FD_WAVES;
#define LT_FD_LAUNCH(V_, O_)                                                                                                  \
    hipLaunchKernelGGL((k_s1d_feature_rows<V_, O_>), dim3(blocks), dim3(64 * FD_WAVES), smem, st, n, F, H, Hp, b->X, (long)b->ldx, \
                       b->fd_ref, b->W1, cref, b->S1d, fd_hint_cap(F), b->fd_hint_dev, nslab, b->fd_slabs, zstate, s1x,           \
                       defer ? (unsigned *)b->fd_gate : (unsigned *)nullptr, b->fd_cref, b->S1qs, jb, job_first, lt_tune().feature_flags, lt_tune().feature_stagger)
    if (vec2 && one) LT_FD_LAUNCH(2, true);
    else if (vec2) LT_FD_LAUNCH(2, false);
    else if (one) LT_FD_LAUNCH(1, true);
    else LT_FD_LAUNCH(1, false);
#undef LT_FD_LAUNCH
    LT_CHECK_LAUNCH();
    b->cref_deferred = defer;
    b->s1_f32 = s1x != nullptr;
    return LT_OK;
}

// The fp64 product S1d = X W1 by the baseline's route (nothing on the aggregate-first route), and every pre-activation row
// marked stale: Z1d is formed afterwards -- for all rows (form_z1d) or only for the rows a call's items read
// (lt_fp64_prepare_rows / lt_fp64_prepare_items).
static int compute_s1d(lt_baseline *b, hipStream_t st) {
    if (b->n == 0) return LT_OK;
    const int Hp = b->Hp, H = b->H, n = b->n;
    b->z_all_valid = false;
    b->z1x_valid = false;
    // every pre-activation row is stale from here on (the feature-rows kernel resets the words itself)
    const bool feat = !lt_fp64_agg_active(b) && b->S1d && !b->S1d_external && want_feature_rows(b);
    // (the matrix-core product whose rows leave through k_sum_slabs_f64_q: that launch clears the words)
    bool dense_resets = false;
    if (!feat && !lt_fp64_agg_active(b) && b->S1d && !b->S1d_external && dense_quant_possible(b) && b->slabs_d) {
        const int ks0 = fp64_kslice(n, H, b->F);
        dense_resets = (i8_route(b) ? lt_i8_slices(n, H, b->F) : (b->F + ks0 - 1) / ks0) > 1;
    }
    if (!feat && !dense_resets) LT_HIP(hipMemsetAsync(b->zstate, 0, (size_t)n * sizeof(int32_t), st));
    if (lt_fp64_agg_active(b)) return LT_OK;
    if (!b->S1d)
        return lt_set_error(LT_ERR_UNSUPPORTED, "fp64 pre-activation: the S1d route was not allocated (set \"aggregate_first\" "
                                                "before lt_baseline_enable_fp64)");
    lt_prof_scope prof_(LT_K_FP64_PRODUCT, st);
    b->cref_deferred = false;
    b->s1_f32 = false;
    if (b->S1d_external) {
        // multi-GPU: S1d arrives by the caller's all-gather of the ranks' row shards (lt_baseline_refresh_rows_fp64); the rows
        // take the storage form the single-GPU product would have (the same fp64 values -> the same fixed-point words)
        if (dense_quant_possible(b)) {
            int rc = quant_rows(b, b->S1d, st);
            if (rc) return rc;
            b->s1_f32 = true;
        }
    } else if (want_feature_rows(b)) {
        // (writes the pad columns of S1d as zeros itself.)  The deferred cref + fp32 storage are for products the caches hold
        // (a fixed size rule, not the "tiled_min_bytes" knob, so that knob keeps every bit); beyond it the rows carry cref
        // themselves in fp64 and the SpMM may take the tiled route
        const bool small = (long long)n * Hp * (long long)sizeof(double) < ((long long)32 << 20);
        int rc = launch_feature_s1d(b, st, -1, small && lt_tune().defer_cref != 0, b->zstate);
        if (rc) return rc;
    } else {
        // the matrix-core product; rows the caches hold leave as 32-bit fixed point (round 5: the storage form of the feature route:
        // "s1_f32" = 0 keeps them fp64)
        const bool quant = dense_quant_possible(b);
        const int ks1 = fp64_kslice(n, H, b->F);
        if ((!quant || (b->F + ks1 - 1) / ks1 <= 1) && Hp != H) LT_HIP(hipMemsetAsync(b->S1d, 0, (size_t)n * Hp * sizeof(double), st));
        int rc = launch_dense_s1d(b, 0, n, b->S1d, (long)Hp, st, quant, dense_resets ? b->zstate : (int32_t *)nullptr);
        if (rc) return rc;
        b->s1_f32 = quant;
    }
    return LT_OK;
}

// Z1d = A_hat S1d + b1: every row (state == NULL) or the rows marked 2 in `state`
static int form_z1d(lt_baseline *b, int32_t *state, hipStream_t st, const lt_bits_job *job = nullptr, bool *job_done = nullptr,
                    bool allow_f32 = false) {
    const int Hp = b->Hp, n = b->n;
    const int lpr = lt_lpr_for(Hp);
    const unsigned g2 = (unsigned)((n + (4 * (64 / lpr)) - 1) / (4 * (64 / lpr)));
    const lt_graph *g = b->g;
    const int have_long = (g->q_n_long > 0 && b->seg_d) ? 1 : 0;     // (the row kernel's own cut: lt_graph::q_*)
    b->z1x_valid = false;
    if (!state && !b->cref_deferred && !b->s1_f32 && lt_tiled_wanted(g, Hp) && (g->p_n_long == 0 || b->seg_d)) {      // (fixed-point rows: the row kernel reads them)
        // S1d beyond the caches (R-MAT scale 21: 4.3 GB): the column-sliced work-item route of lt_spmm.hip, same chains
        int rc = lt_launch_rows_tiled_f64(g, b->S1d, Hp, Hp, b->b1p, b->Z1d, Hp, b->seg_d, Hp, st);
        if (rc) return rc;
        if (g->p_n_long > 0 && b->seg_d) {      // (the work items carry the 128-entry segments of lt_graph::p_*)
            const long tot = (long)g->p_n_long * Hp;
            hipLaunchKernelGGL(k_spmm_f64_long, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g->p_n_long,
                               g->p_long_row, g->p_long_segptr, b->seg_d, Hp, b->b1p, b->Z1d, (int32_t *)nullptr, 0,
                               (const double *)nullptr, (const double *)nullptr);
            LT_CHECK_LAUNCH();
        }
        return LT_OK;
    }
    const unsigned gs = have_long ? (unsigned)((g->q_n_seg + (4 * (64 / lpr)) - 1) / (4 * (64 / lpr))) : 0u;
    const double *crefv = b->cref_deferred ? b->fd_cref : nullptr;      // (deferred cref: S1d holds S1d - cref)
    const lt_bits_job jb = (job && !state) ? *job : lt_bits_job{};
    // the feature route with fp32 row storage, for LT_MODE_DELTA's stage A alone: the result in fp32 too
    float *zf = (allow_f32 && b->s1_f32 && b->Z1x && lt_tune().s1_f32 != 0) ? b->Z1x : nullptr;     // (on demand too: the same bits)
    b->z1x_valid = zf != nullptr;
    const unsigned gz = jb.zero_blocks > 0 ? (unsigned)jb.zero_blocks : 0u;
    const unsigned gj = (jb.nblocks > 0 ? (unsigned)jb.nblocks : 0u);
    if (job_done) *job_done = gj + gz > 0;
    const size_t jsm = gj > 0 ? (size_t)jb.smem_bytes : 0;
    if (b->s1_f32) {
        LT_DISPATCH_LPR(lpr, hipLaunchKernelGGL((k_spmm_f64<LPR_, float>), dim3(g2 + gs + gj + gz), dim3(256), jsm, st, n, g->rowptr,
                                                g->col, g->val, b->S1x, Hp, b->b1p, b->Z1d, (int)gs, g->q_n_seg, g->q_seg_begin,
                                                g->q_seg_long, g->q_long_row, b->seg_d, state, b->fd_rs, crefv, jb, (int)(g2 + gs), zf, b->S1qs));
    } else {
        LT_DISPATCH_LPR(lpr, hipLaunchKernelGGL((k_spmm_f64<LPR_, double>), dim3(g2 + gs + gj + gz), dim3(256), jsm, st, n, g->rowptr,
                                                g->col, g->val, b->S1d, Hp, b->b1p, b->Z1d, (int)gs, g->q_n_seg, g->q_seg_begin,
                                                g->q_seg_long, g->q_long_row, b->seg_d, state, b->fd_rs, crefv, jb, (int)(g2 + gs), zf));
    }
    LT_CHECK_LAUNCH();
    if (have_long) {
        const long tot = (long)g->q_n_long * Hp;
        hipLaunchKernelGGL(k_spmm_f64_long, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g->q_n_long,
                           g->q_long_row, g->q_long_segptr, b->seg_d, Hp, b->b1p, b->Z1d, state, 0, b->fd_rs, crefv, zf);
        LT_CHECK_LAUNCH();
        if (state) {
            hipLaunchKernelGGL(k_spmm_f64_long, dim3((unsigned)((g->q_n_long + 255) / 256)), dim3(256), 0, st, g->q_n_long,
                               g->q_long_row, g->q_long_segptr, b->seg_d, Hp, b->b1p, b->Z1d, state, 1, b->fd_rs, crefv);
            LT_CHECK_LAUNCH();
        }
    }
    return LT_OK;
}

// S1d routes, per probe chunk of an LT_MODE_DELTA call: make the pre-activation rows the chunk's items read valid.  A call
// whose items cannot cover much of the graph (n_probe_call * average column length well below n: one rank of many) forms
// only those rows (k_z_mark + the row kernel restricted to them; rows stay valid for later chunks and calls); otherwise all
// rows once.  Same chains either way: the bits do not depend on which rows were asked for.
// Would lt_fp64_prepare_rows form the pre-activation only on the rows a call of `n_probe_call` probes reads (true), or on all rows?
bool lt_fp64_on_demand(const lt_baseline *b, int n_probe_call) {
    const lt_graph *g = b->g;
    const double avg = g->n > 0 ? (double)g->nnz / (double)g->n : 0.0;
    const int knob = lt_tune().z_on_demand;
    const bool tiled = !b->cref_deferred && !b->s1_f32 && lt_tiled_wanted(g, b->Hp) && (g->p_n_long == 0 || b->seg_d);
    // (a graph whose whole fp64 SpMM is a 10 us launch is formed whole: marking the rows costs a memset, k_z_mark and the item
    // tables' own launch in front of it -- 55 against 50 us for the step one rank of 8 runs at twitch size)
    return !tiled && (knob > 0 || (knob < 0 && (double)n_probe_call * avg * 2.0 < (double)g->n &&
                                   (double)g->nnz * (double)b->Hp >= 2.5e8));
}

int lt_fp64_prepare_rows(const lt_baseline *cb, const int32_t *off, int nb, const int2 *item_pr, int n_probe_call, hipStream_t st,
                         const lt_bits_job *job, bool *job_done) {
    lt_baseline *b = const_cast<lt_baseline *>(cb);   // cache state only
    if (job_done) *job_done = false;
    if (b->z_all_valid || b->n == 0) return LT_OK;
    lt_prof_scope prof_(LT_K_FP64_SPMM, st);
    const bool ondemand = lt_fp64_on_demand(b, n_probe_call);
    if (!ondemand) {
        const int rc = form_z1d(b, nullptr, st, job, job_done, true);
        if (rc) return rc;
        b->z_all_valid = true;
        return LT_OK;
    }
    if (job) return LT_OK;        // on demand: the marks below read the job's tables -- the caller makes them and calls again
    LT_HIP(hipMemsetAsync(b->zcount, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_z_mark, dim3(256), dim3(256), 0, st, off, nb, item_pr, b->zstate, b->zrows, b->zcount);
    LT_CHECK_LAUNCH();
    return form_z1d(b, b->zstate, st, nullptr, nullptr, true);
}

// ---- pieces the 3-layer model's `delta` (lt_gcn3.hip) builds its fp64 baseline from ------------------------------------
// every row of Z1d valid (S1d routes only: the caller set no_agg before lt_baseline_enable_fp64)
int lt_fp64_form_all(lt_baseline *b, hipStream_t st) {
    int rc = lt_baseline_ensure_layers(b, true, st, false);
    if (rc) return rc;
    if (b->z_all_valid || b->n == 0) return LT_OK;
    lt_prof_scope prof_(LT_K_FP64_SPMM, st);
    rc = form_z1d(b, nullptr, st);
    if (rc) return rc;
    b->z_all_valid = true;
    return LT_OK;
}
// C[i, 0..N) = act(A[i, 0..K)) * B + bias, i < M, fp64 accumulation on the matrix cores (A double; act = relu when relu_a)
int lt_launch_gemm_f64_dense(const double *A, long lda, int M, const float *B, long ldb, int N, int K, const float *bias,
                             double *C, long ldc, int relu_a, hipStream_t st) {
    if (M <= 0) return LT_OK;
    const long tiles = (long)((M + GD_BM - 1) / GD_BM) * ((N + GD_BN - 1) / GD_BN);
    if (tiles < 512)       // (less than two tiles per CU: nothing but the loads' own depth hides their round trips)
        hipLaunchKernelGGL((k_gemm_f64_rows<double, 4>), dim3((unsigned)tiles), dim3(256), 0, st, A, lda,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, M, B, ldb, N, K, bias, C, ldc, 0, (int32_t *)nullptr, relu_a);
    else
        hipLaunchKernelGGL((k_gemm_f64_rows<double>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), 0, st, A, lda,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, M, B, ldb, N, K, bias, C, ldc, 0, (int32_t *)nullptr, relu_a);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
// C[i, 0 .. Hp) = the fp64 product row S1[rows[i]] = X[rows[i]] W1 as the baseline holds it after lt_fp64_form_all -- fp64 rows, or the
// 32-bit fixed-point rows times their scale, plus the reference vector's product where that is deferred: the expression the
// 2-layer stage A reads a probe's row by (k_item_stageA_d2).  Returns 1 (nothing enqueued) when the baseline holds no product
// rows (aggregate-first route): the caller forms them itself (lt_launch_gemm_f64_gather).
static __global__ __launch_bounds__(256) void k_product_rows_gather(const int32_t *__restrict__ rows, int m, int Hp,
                                                                     const double *__restrict__ S1d, const float *__restrict__ S1x,
                                                                     const double *__restrict__ S1qs, const double *__restrict__ cref,
                                                                     double *__restrict__ C, long ldc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)m * Hp) return;
    const int r = (int)(i / Hp), c = (int)(i % Hp), v = rows[r];
    double s = S1x ? (double)__float_as_int(S1x[(size_t)v * Hp + c]) * S1qs[v] : S1d[(size_t)v * Hp + c];
    if (cref) s += cref[c];
    C[(size_t)r * ldc + c] = s;
}
int lt_fp64_product_rows_gather(const lt_baseline *b, const int32_t *rows, int m, double *C, long ldc, hipStream_t st) {
    if (m <= 0) return LT_OK;
    if (lt_fp64_agg_active(b) || !b->S1d || b->S1d_external) return 1;
    const float *sx = b->s1_f32 ? b->S1x : (const float *)nullptr;
    if (b->s1_f32 && (!b->S1x || !b->S1qs)) return 1;
    const long tot = (long)m * b->Hp;
    hipLaunchKernelGGL(k_product_rows_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, rows, m, b->Hp, b->S1d, sx, b->S1qs,
                       b->cref_deferred ? b->fd_cref : (const double *)nullptr, C, ldc);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
// C[i, 0..N) = A[rows[i], 0..K) * B, i < M, A fp32 (feature rows), fp64 accumulation
int lt_launch_gemm_f64_gather(const float *A, long lda, const int32_t *rows, int M, const float *B, long ldb, int N, int K,
                              double *C, long ldc, hipStream_t st) {
    if (M <= 0) return LT_OK;
    const long tiles = (long)((M + GD_BM - 1) / GD_BM) * ((N + GD_BN - 1) / GD_BN);
    if (tiles < 512)
        hipLaunchKernelGGL((k_gemm_f64_rows<float, 4>), dim3((unsigned)tiles), dim3(256), 0, st, A, lda, rows,
                           (const int32_t *)nullptr, M, B, ldb, N, K, (const float *)nullptr, C, ldc, 0, (int32_t *)nullptr, 0);
    else
        hipLaunchKernelGGL((k_gemm_f64_rows<float>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), 0, st, A, lda, rows,
                           (const int32_t *)nullptr, M, B, ldb, N, K, (const float *)nullptr, C, ldc, 0, (int32_t *)nullptr, 0);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
// out[n, ld] = A_hat * S + bias (all rows, fp64 chains; seg_d: [lt_f64_seg_rows(g), ld] scratch for the hub rows, may be NULL
// when the graph has none)
int lt_launch_spmm_f64(const lt_graph *g, const double *S, int ld, const float *biasp, double *out, double *seg_d, hipStream_t st) {
    const int n = g->n;
    if (n == 0) return LT_OK;
    const int lpr = lt_lpr_for(ld);
    const unsigned g2 = (unsigned)((n + (4 * (64 / lpr)) - 1) / (4 * (64 / lpr)));
    const int have_long = (g->q_n_long > 0 && seg_d) ? 1 : 0;
    const unsigned gs = have_long ? (unsigned)((g->q_n_seg + (4 * (64 / lpr)) - 1) / (4 * (64 / lpr))) : 0u;
    LT_DISPATCH_LPR(lpr, hipLaunchKernelGGL((k_spmm_f64<LPR_, double>), dim3(g2 + gs), dim3(256), 0, st, n, g->rowptr, g->col, g->val, S, ld,
                                            biasp, out, (int)gs, g->q_n_seg, g->q_seg_begin, g->q_seg_long, g->q_long_row, seg_d,
                                            (int32_t *)nullptr, (const double *)nullptr, (const double *)nullptr));
    LT_CHECK_LAUNCH();
    if (have_long) {
        const long tot = (long)g->q_n_long * ld;
        hipLaunchKernelGGL(k_spmm_f64_long, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g->q_n_long, g->q_long_row,
                           g->q_long_segptr, seg_d, ld, biasp, out, (int32_t *)nullptr, 0, (const double *)nullptr, (const double *)nullptr);
        LT_CHECK_LAUNCH();
    }
    return LT_OK;
}

int lt_baseline_refresh_fp64(lt_baseline *b, hipStream_t st) {
    return b->Z1d ? compute_s1d(b, st) : LT_OK;
}

extern "C" int lt_baseline_enable_fp64(lt_baseline *b, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline_enable_fp64: baseline is NULL");
    if (b->Z1d) return LT_OK;   // Z1d is set only once all the buffers exist (see below)
    hipStream_t st = (hipStream_t)stream;
    const size_t n1 = (size_t)(b->n > 0 ? b->n : 1);
    const size_t nh = n1 * b->Hp * sizeof(double);
    b->Fp = lt_round_up(b->F, 4);
    // Which route forms the pre-activation is decided here, once, from the features and the shapes (the knobs pin it):
    //   1. feature rows that are sparse differences to a reference row (standardised indicators)  -> k_s1d_feature_rows
    //   2. else features no wider than ~2 H                                                       -> aggregate-first
    //   3. else                                                                                   -> the f64 matrix cores
    // Routes 1 / 3 share S1d + the all-rows fp64 SpMM; route 2 has its own buffers.  Small problems get both sets (the
    // knobs may then switch at any time); large ones only the chosen set.
    // (the rows kernel reads X in whole trips of 26 x 128 floats: tiny matrices take the matrix cores)
    const bool fd_possible = b->n >= 2 && fd_smem_bytes(b->F) <= (size_t)60 * 1024 && lt_tune().feature_delta != 0 &&
                             (long)(b->n - 1) * b->ldx + b->F >= (long)FD_UN * 128;
    int feat = -1;
    int *hint_host = nullptr, *hint_dev = nullptr;
    if (fd_possible) {
        // the word k_s1d_feature_rows sets when it meets dense rows: mapped host memory, read by want_feature_rows()
        if (hipHostMalloc((void **)&hint_host, sizeof(int), hipHostMallocMapped) == hipSuccess) {
            *hint_host = 0;
            if (hipHostGetDevicePointer((void **)&hint_dev, hint_host, 0) != hipSuccess) { (void)hipHostFree(hint_host); hint_host = nullptr; }
        }
        (void)hipGetLastError();
    }
    if (fd_possible && hint_host) {
        // probe (this call allocates, so it may synchronise): do the first rows differ from the reference row in few columns?
        // Temporary buffers: the real ones are allocated once the route is known.
        const int n_probe = b->n < 4096 ? b->n : 4096;
        double *cref = nullptr, *fslabs = nullptr, *s1d = nullptr;
        float *fref = nullptr;
        int *gate = nullptr;
        hipError_t e = hipMalloc((void **)&cref, (size_t)b->Hp * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&fref, (size_t)(b->F + FD_REF_PAD) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&fslabs, fd_slab_doubles(b->F, b->H) * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&gate, FR_GATE_WORDS * sizeof(int));
        if (e == hipSuccess) e = hipMemsetAsync(gate, 0, FR_GATE_WORDS * sizeof(int), st);
        if (e == hipSuccess) e = hipMalloc((void **)&s1d, (size_t)n_probe * b->Hp * sizeof(double));
        if (e == hipSuccess) {
            b->fd_cref = cref; b->fd_slabs = fslabs; b->fd_gate = gate; b->S1d = s1d; b->fd_ref = fref;
            b->fd_hint_host = hint_host; b->fd_hint_dev = hint_dev; b->fd_ref_valid = false;
            const int rc = launch_feature_s1d(b, st, n_probe);
            if (rc == LT_OK && hipStreamSynchronize(st) == hipSuccess) feat = *(volatile int *)hint_host == 0 ? 1 : 0;
            b->fd_cref = b->fd_slabs = nullptr; b->fd_gate = nullptr; b->S1d = nullptr; b->fd_ref = nullptr;
            b->fd_hint_host = b->fd_hint_dev = nullptr; b->fd_ref_valid = false;
        }
        (void)hipStreamSynchronize(st);
        (void)hipFree(cref); (void)hipFree(fslabs); (void)hipFree(gate); (void)hipFree(s1d); (void)hipFree(fref);
        (void)hipGetLastError();
        *hint_host = 0;
    }
    const int aknob = lt_tune().aggregate_first;
    const bool agg_ok = agg_shapes_ok(b) && aknob != 0;
    const bool agg_chosen = agg_ok && (aknob > 0 || feat != 1);
    const bool small = nh <= ((size_t)256 << 20);
    const bool alloc_s1d = !agg_chosen || small;
    const bool alloc_agg = agg_chosen || (agg_ok && small);

    const int ks_ = fp64_kslice(b->n, b->H, b->F);
    const int splits = (b->F + ks_ - 1) / ks_;
    double *s1d = nullptr, *z1d = nullptr, *slabs = nullptr, *segd = nullptr, *cref = nullptr, *fslabs = nullptr;
    double *yd = nullptr, *segy = nullptr;
    float *fref = nullptr;
    double *frs = nullptr;
    float *fs1x = nullptr, *fz1x = nullptr;
    double *fs1q = nullptr;
    int *gate = nullptr;
    int8_t *i8wd = nullptr;
    unsigned *i8ew = nullptr;
    int32_t *zst = nullptr, *zrw = nullptr, *zct = nullptr, *zit = nullptr, *zic = nullptr;
    hipError_t e = hipMalloc((void **)&z1d, nh);
    if (e == hipSuccess) e = hipMemsetAsync(z1d, 0, nh, st);      // (pad columns stay zero on every route)
    if (alloc_s1d) {
        if (e == hipSuccess) e = hipMalloc((void **)&s1d, nh);
        if (e == hipSuccess && lt_f64_seg_rows(b->g) > 0) e = hipMalloc((void **)&segd, (size_t)lt_f64_seg_rows(b->g) * b->Hp * sizeof(double));
        const bool i8_ok = lt_i8_shapes_ok(b->n, b->H, b->F);
        const int slab_n = std::max(splits > 1 ? splits : 0, i8_ok ? lt_i8_slices(b->n, b->H, b->F) : 0);
        if (e == hipSuccess && slab_n > 0) e = hipMalloc((void **)&slabs, (size_t)slab_n * n1 * b->H * sizeof(double));
        if (e == hipSuccess && i8_ok) e = hipMalloc((void **)&i8wd, lt_i8_wd_bytes(b->H, b->F));
        if (e == hipSuccess && i8_ok) e = hipMalloc((void **)&i8ew, lt_i8_ew_bytes(b->n, b->H, b->F));
        if (e == hipSuccess && fd_possible) e = hipMalloc((void **)&cref, (size_t)b->Hp * sizeof(double));
        if (e == hipSuccess && fd_possible) e = hipMalloc((void **)&fslabs, fd_slab_doubles(b->F, b->H) * sizeof(double));
        if (e == hipSuccess && fd_possible) e = hipMalloc((void **)&gate, FR_GATE_WORDS * sizeof(int));
        if (e == hipSuccess && fd_possible) e = hipMemsetAsync(gate, 0, FR_GATE_WORDS * sizeof(int), st);     // (the slice counter of k_ref_row_product)
        if (e == hipSuccess && fd_possible) e = hipMalloc((void **)&fref, (size_t)(b->F + FD_REF_PAD) * sizeof(float));
        if (e == hipSuccess && fd_possible) e = hipMalloc((void **)&frs, n1 * sizeof(double));
        const bool fixed = fd_possible || dense_quant_shapes(b);     // (the fixed-point rows: feature route, or a small dense product)
        if (e == hipSuccess && fixed) e = hipMalloc((void **)&fs1x, n1 * b->Hp * sizeof(float));
        if (e == hipSuccess && fixed) e = hipMalloc((void **)&fz1x, n1 * b->Hp * sizeof(float));
        if (e == hipSuccess && fixed) e = hipMalloc((void **)&fs1q, n1 * sizeof(double));
    }
    if (alloc_agg) {
        if (e == hipSuccess) e = hipMalloc((void **)&yd, n1 * b->Fp * sizeof(double));
        if (e == hipSuccess && b->g->p_n_seg > 0) e = hipMalloc((void **)&segy, (size_t)b->g->p_n_seg * b->Fp * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&zit, lt_xf64_scratch_words(b->g) * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&zic, sizeof(int32_t));
    }
    // per-row validity of Z1d (every route): rows are formed for all, or on demand for the rows a call reads
    if (e == hipSuccess) e = hipMalloc((void **)&zst, n1 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&zrw, n1 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&zct, sizeof(int32_t));
    if (e != hipSuccess) {   // all or nothing: a retry starts from a clean state, nothing leaks
        (void)hipFree(s1d); (void)hipFree(z1d); (void)hipFree(slabs); (void)hipFree(segd);
        (void)hipFree(cref); (void)hipFree(fslabs); (void)hipFree(gate); (void)hipFree(fref); (void)hipFree(frs); (void)hipFree(fs1x); (void)hipFree(fz1x); (void)hipFree(fs1q);
        (void)hipFree(i8wd); (void)hipFree(i8ew);
        (void)hipFree(yd); (void)hipFree(segy); (void)hipFree(zst); (void)hipFree(zrw); (void)hipFree(zct); (void)hipFree(zit); (void)hipFree(zic);
        if (hint_host) (void)hipHostFree(hint_host);
        return lt_set_error(LT_ERR_HIP, "lt_baseline_enable_fp64: hipMalloc failed: %s", hipGetErrorString(e));
    }
    b->S1d = s1d; b->Z1d = z1d; b->slabs_d = slabs; b->seg_d = segd;
    b->i8_wd = i8wd; b->i8_ew = i8ew;
    b->fd_cref = cref; b->fd_slabs = fslabs; b->fd_gate = gate; b->fd_ref = fref; b->fd_rs = frs; b->S1x = fs1x; b->Z1x = fz1x; b->S1qs = fs1q;
    if (frs && b->n > 0) {
        hipLaunchKernelGGL(k_row_sums, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, st, b->n, b->g->rowptr, b->g->val, frs);
        LT_CHECK_LAUNCH();
    }
    b->fd_hint_host = hint_host; b->fd_hint_dev = hint_dev;
    if (!cref && hint_host) { (void)hipHostFree(hint_host); b->fd_hint_host = b->fd_hint_dev = nullptr; }
    b->Yd = yd; b->seg_y = segy; b->zstate = zst; b->zrows = zrw; b->zcount = zct; b->zitems = zit; b->zicount = zic;
    b->S1d_owned = true; b->S1d_external = false; b->feat_sparse = feat; b->agg_default = agg_chosen;
    int rc = lt_baseline_ensure_padding(b, st);   // (the padded bias the fp64 SpMM adds)
    if (rc) return rc;
    rc = compute_s1d(b, st);
    b->fp64_fresh = rc == LT_OK;
    return rc;
}

// ---- the aggregate-first pre-activation by ROW LIST: what lets several ranks share the rows all of them reach -------------------
// On the on-demand route (lt_baseline_fp64_route == 2) a call forms Z1d on the rows its own probes reach.  On a heavy-tailed
// graph most of that work is in hub rows EVERY rank's probes reach (R-MAT scale 21, 8 x 512 probes: 84 % of a rank's gathers
// are in ~3 800 rows of >= 1 024 entries common to all ranks).  With these three a caller splits such rows over the ranks:
//   lt_baseline_form_rows_fp64    Z1d[r] for the listed rows now (rows already valid since the last refresh are skipped)
//   lt_baseline_gather_rows_fp64  dst[i, 0 .. Hp) = Z1d[rows[i]]           (the rank's send buffer)
//   lt_baseline_scatter_rows_fp64 Z1d[rows[i]] = src[i, 0 .. Hp), row valid (the all-gather's output, every rank's rows)
// A row carries the same bits whichever rank formed it (its gathers and its product are a function of the row alone).
static __global__ __launch_bounds__(256) void k_z_mark_rows(const int32_t *__restrict__ rows, int n_rows, int n,
                                                            int32_t *__restrict__ zstate, int32_t *__restrict__ zrows,
                                                            int32_t *__restrict__ zcount) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (long)gridDim.x * 256) {
        const int r = rows[i];
        if ((unsigned)r >= (unsigned)n || zstate[r] != 0) continue;
        if (atomicCAS(&zstate[r], 0, 2) == 0) zrows[atomicAdd(zcount, 1)] = r;
    }
}
static __global__ __launch_bounds__(256) void k_z_rows_copy(const int32_t *__restrict__ rows, int n_rows, int n, int Hp,
                                                            double *__restrict__ Z1d, double *__restrict__ buf,
                                                            int32_t *__restrict__ zstate, int scatter) {
    // a lane group of Hp / 2 lanes (16-byte pieces) per row
    const int per = Hp / 2;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < (long)n_rows * per; k += (long)gridDim.x * 256) {
        const int i = (int)(k / per), c = (int)(k % per) * 2;
        const int r = rows[i];
        if ((unsigned)r >= (unsigned)n) continue;
        double2 *z = reinterpret_cast<double2 *>(Z1d + (size_t)r * Hp + c), *q = reinterpret_cast<double2 *>(buf + (size_t)i * Hp + c);
        if (scatter) { *z = *q; if (c == 0) zstate[r] = 1; }
        else *q = *z;
    }
}
static int agg_rows_check(const char *who, lt_baseline *b, const int32_t *rows, int32_t n_rows, hipStream_t st) {
    LT_REQUIRE(b != nullptr, "%s: baseline is NULL", who);
    LT_REQUIRE(n_rows >= 0 && (rows != nullptr || n_rows == 0), "%s: bad row list", who);
    if (!lt_fp64_agg_active(b))
        return lt_set_error(LT_ERR_UNSUPPORTED, "%s: the baseline is not on the aggregate-first route (lt_baseline_fp64_route != 2)", who);
    return lt_baseline_ensure_layers(b, true, st, false);      // (a refresh since the last call: every row stale again)
}
extern "C" int lt_baseline_form_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = agg_rows_check("lt_baseline_form_rows_fp64", b, rows, n_rows, st);
    if (rc || n_rows == 0) return rc;
    { lt_prof_scope prof_(LT_K_FP64_SPMM, st);
    LT_HIP(hipMemsetAsync(b->zcount, 0, sizeof(int32_t), st));
    const int blocks = (n_rows + 255) / 256;
    hipLaunchKernelGGL(k_z_mark_rows, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, rows, n_rows, b->n, b->zstate,
                       b->zrows, b->zcount);
    LT_CHECK_LAUNCH(); }
    return agg_form_marked(b, st);
}
static int agg_rows_copy(const char *who, lt_baseline *b, const int32_t *rows, int32_t n_rows, double *buf, hipStream_t st, int scatter) {
    int rc = agg_rows_check(who, b, rows, n_rows, st);
    if (rc || n_rows == 0) return rc;
    LT_REQUIRE(buf != nullptr && ((uintptr_t)buf & 15) == 0, "%s: buffer is NULL or not 16-byte aligned", who);
    const long total = (long)n_rows * (b->Hp / 2);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_z_rows_copy, dim3((unsigned)blocks), dim3(256), 0, st, rows, n_rows, b->n, b->Hp, b->Z1d, buf, b->zstate, scatter);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
extern "C" int lt_baseline_gather_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, double *dst, void *stream) {
    return agg_rows_copy("lt_baseline_gather_rows_fp64", b, rows, n_rows, dst, (hipStream_t)stream, 0);
}
extern "C" int lt_baseline_scatter_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, const double *src, void *stream) {
    return agg_rows_copy("lt_baseline_scatter_rows_fp64", b, rows, n_rows, const_cast<double *>(src), (hipStream_t)stream, 1);
}

extern "C" int lt_baseline_fp64_route(const lt_baseline *b, int32_t *route) {
    LT_REQUIRE(b != nullptr && route != nullptr, "lt_baseline_fp64_route: NULL argument");
    *route = !b->Z1d ? -1 : (lt_fp64_agg_active(b) ? 2 : (want_feature_rows(b) ? 1 : 0));
    return LT_OK;
}

// Multi-GPU, dense features: rows [row_begin, row_end) of the fp64 product into dst[(row_end - row_begin), Hp] (the rank's
// send buffer); the ranks' all-gather rebuilds S1d in the storage attached with lt_baseline_attach_s1d.  The K slicing
// is the one of the full product, so a row has the same bits whichever rank computed it.
extern "C" int lt_baseline_refresh_rows_fp64(lt_baseline *b, int32_t row_begin, int32_t row_end, double *dst, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline_refresh_rows_fp64: baseline is NULL");
    LT_REQUIRE(b->Z1d != nullptr, "lt_baseline_refresh_rows_fp64: call lt_baseline_enable_fp64 first");
    if (lt_fp64_agg_active(b) || !b->S1d)
        return lt_set_error(LT_ERR_UNSUPPORTED, "lt_baseline_refresh_rows_fp64: this baseline forms its pre-activation aggregate-first "
                                                "(no S1d to shard: lt_baseline_fp64_route == 2)");
    LT_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= b->n,
               "lt_baseline_refresh_rows_fp64: rows [%d, %d) outside [0, %d]", row_begin, row_end, b->n);
    LT_REQUIRE(dst != nullptr && ((uintptr_t)dst % 16) == 0, "lt_baseline_refresh_rows_fp64: dst is NULL or not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    b->fp64_fresh = false;
    const int m = row_end - row_begin;
    if (m == 0) return LT_OK;
    if (b->Hp != b->H) LT_HIP(hipMemsetAsync(dst, 0, (size_t)m * b->Hp * sizeof(double), st));
    lt_prof_scope prof_(LT_K_FP64_PRODUCT, st);
    return launch_dense_s1d(b, row_begin, row_end, dst, (long)b->Hp, st);
}

// S1d lives in caller-owned storage from now on and is filled from outside (the ranks' all-gather of the shards
// lt_baseline_refresh_rows_fp64 produced): a refresh then recomputes only Z1d = A_hat*S1d + b1.  S1d == NULL returns to
// the library's own storage and product.
extern "C" int lt_baseline_attach_s1d(lt_baseline *b, double *S1d, int64_t ld, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline_attach_s1d: baseline is NULL");
    LT_REQUIRE(b->Z1d != nullptr, "lt_baseline_attach_s1d: call lt_baseline_enable_fp64 first");
    if (S1d != nullptr && (lt_fp64_agg_active(b) || !b->S1d))
        return lt_set_error(LT_ERR_UNSUPPORTED, "lt_baseline_attach_s1d: this baseline forms its pre-activation aggregate-first "
                                                "(no S1d: lt_baseline_fp64_route == 2)");
    hipStream_t st = (hipStream_t)stream;
    if (S1d == nullptr) {
        if (!b->S1d_owned) {
            double *own = nullptr;
            LT_HIP(hipMalloc((void **)&own, (size_t)(b->n > 0 ? b->n : 1) * b->Hp * sizeof(double)));
            b->S1d = own;
            b->S1d_owned = true;
        }
        b->S1d_external = false;
        b->fp64_fresh = false;
        return LT_OK;
    }
    LT_REQUIRE(ld == b->Hp, "lt_baseline_attach_s1d: ld=%lld, must equal the padded hidden width %d", (long long)ld, b->Hp);
    LT_REQUIRE(((uintptr_t)S1d % 16) == 0, "lt_baseline_attach_s1d: storage must be 16-byte aligned");
    if (S1d != b->S1d) {
        if (b->S1d_owned) {
            LT_HIP(hipStreamSynchronize(st));   // kernels in flight may still read the buffer freed below
            (void)hipFree(b->S1d);
        }
        b->S1d = S1d;
        b->S1d_owned = false;
    }
    b->S1d_external = true;
    b->fp64_fresh = false;
    return LT_OK;
}

void lt_baseline_free_fp64(lt_baseline *b) {
    if (b->S1d_owned) (void)hipFree(b->S1d);
    (void)hipFree(b->Z1d);
    (void)hipFree(b->slabs_d);
    (void)hipFree(b->i8_wd);
    (void)hipFree(b->i8_ew);
    (void)hipFree(b->seg_d);
    (void)hipFree(b->fd_cref);
    (void)hipFree(b->fd_slabs);
    (void)hipFree(b->fd_gate);
    (void)hipFree(b->fd_ref);
    (void)hipFree(b->fd_rs);
    (void)hipFree(b->S1x);
    (void)hipFree(b->Z1x);
    (void)hipFree(b->S1qs);
    b->S1x = b->Z1x = nullptr;
    b->S1qs = nullptr;
    b->z1x_valid = false;
    b->s1_f32 = false;
    b->fd_ref = nullptr;
    b->fd_rs = nullptr;
    b->cref_deferred = false;
    b->fd_ref_valid = false;
    if (b->fd_hint_host) (void)hipHostFree(b->fd_hint_host);
    b->fd_hint_host = b->fd_hint_dev = nullptr;
    (void)hipFree(b->Yd);
    (void)hipFree(b->seg_y);
    (void)hipFree(b->zstate);
    (void)hipFree(b->zrows);
    (void)hipFree(b->zcount);
    (void)hipFree(b->zitems);
    (void)hipFree(b->zicount);
    b->zitems = b->zicount = nullptr;
    b->S1d = b->Z1d = b->slabs_d = b->seg_d = b->fd_cref = b->fd_slabs = b->Yd = b->seg_y = nullptr;
    b->i8_wd = nullptr; b->i8_ew = nullptr;
    b->fd_gate = nullptr;
    b->zstate = b->zrows = b->zcount = nullptr;
    b->fp64_fresh = false;
}
