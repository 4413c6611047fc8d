// Dense fp32 GEMM C[M,N] = A[M,K] * B[K,N] on the gfx950 matrix cores, exact fp32.
//   reference call site: torch.mm(input, self.weight), gcn/layers.py:31 -- the one
//   GEMM-shaped op on the path (X*W1: [N,3170]x[3170,256]); everything else is sparse.
//
// v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate; each output element is a k-ordered fmaf
// chain (no reduced-precision path exists on gfx950), 64 FLOP/clk/SIMD = the fp32 peak.
//
// Summation order (round 4): a K slice is summed as chains of LT_GEMM_FOLD = 128 terms, each started from +0, whose sums are
// added in order ( ((c0 + c1) + c2) + ... ), and the slices of a split-K product are added in slice order as before.  One
// 3170-term chain per output carries ~sqrt(3170) roundings; torch's CPU sgemm (blocked, vectorised partial sums) carries far
// fewer, and the finite difference of `full` / `sparse` amplifies that rounding by 1 / delta = 1e4: with plain chains the fp32
// modes sat at 1.5x (rms) / 2x (max) the REFERENCE's own fp32 error over the whole twitch-ES matrix (BASELINE.md section 3
// asks <= 1x); folded every 128 terms they sit at its level (tests/test_gpu_round4.py, whole-matrix gate; the CPU
// experiment behind the choice: NOTES.md).  The fold is 16 v_add per accumulator tile per 64 MFMAs.  Every kernel below
// folds at the same k (multiples of 128 from the slice start) and once more at the end, so they still give each other's bits.
// Block = 4 waves as 2x2, each wave owns one 32x32 accumulator tile of a 64x64 block tile;
// K is walked in 16-deep tiles staged through LDS with a register prefetch of the next tile
// (one barrier per tile).  A-tile rows are padded to 17 floats: the MFMA A operand is read
// "column-wise" (lane -> row) and 17 is odd, so the 32 lanes of a half hit 32 distinct banks.
#include "lt_internal.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LT_GEMM_FOLD 128   // terms per inner chain (a multiple of every kernel's k-tile depth)
#define GM_BM 64
#define GM_BN 64
#define GM_BK 16
#define GM_LDA (GM_BK + 1)

// GATHER: row m of the A operand is  x + x * delta  of row rows[m] of A, rounded as the reference rounds it
// (perturbed features of probe m: attacker.py:101-105) -- the perturbed rows never exist in memory.
template <bool GATHER>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(const float *__restrict__ A, long lda,
                                                       const float *__restrict__ B, long ldb,
                                                       float *__restrict__ C, long ldc, int M, int N,
                                                       int K, int kslice, long slab_stride,
                                                       const int32_t *__restrict__ rows, float delta,
                                                       const int32_t *__restrict__ m_dev) {
    __shared__ __attribute__((aligned(16))) float As[2][GM_BM * GM_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK * GM_BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GM_BM;  // M tiles on x: neighbours share the B (weight) tiles in L2
    const int n0 = blockIdx.y * GM_BN;
    // m_dev: the row count lives on the device (an item count formed by an earlier kernel); the grid covers the
    // caller's upper bound M and the tiles past the real count leave at once
    if (m_dev != nullptr) {
        M = min(M, *m_dev);
        if (m0 >= M) return;
    }

    // staging coordinates: A tile 64 x 16 (thread -> row tid/4, 4 floats), B tile 16 x 64
    const int a_row = tid >> 2, a_col = (tid & 3) * 4;
    const int b_row = tid >> 4, b_col = (tid & 15) * 4;
    const bool a_row_ok = (m0 + a_row) < M;
    const long a_src = GATHER ? (a_row_ok ? (long)rows[m0 + a_row] : 0L) : (long)(m0 + a_row);
    const float *a_ptr = A + a_src * lda + a_col;
    const float *b_ptr = B + (long)b_row * ldb + n0 + b_col;
    const bool b_full = (n0 + b_col + 3) < N;

    // split-K: blockIdx.z owns K range [kb, ke) and writes its partial tile to slab z
    const int kb = blockIdx.z * kslice;
    const int ke = min(K, kb + kslice);
    C += (long)blockIdx.z * slab_stride;

    f32x4 ra, rb;
    auto load_tiles = [&](int k0) {
        ra = f32x4{0.f, 0.f, 0.f, 0.f};
        rb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_row_ok) {
            if (k0 + a_col + 3 < ke) {
                ra = *reinterpret_cast<const f32x4u *>(a_ptr + k0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + a_col + j < ke) ra[j] = a_ptr[k0 + j];
            }
            if (GATHER) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[j] = __fadd_rn(ra[j], __fmul_rn(ra[j], delta));   // two roundings, no fma
            }
        }
        if (k0 + b_row < ke) {
            const float *p = b_ptr + (long)k0 * ldb;
            if (b_full) {
                rb = *reinterpret_cast<const f32x4u *>(p);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + b_col + j < N) rb[j] = p[j];
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float *as = &As[buf][a_row * GM_LDA + a_col];
        as[0] = ra.x; as[1] = ra.y; as[2] = ra.z; as[3] = ra.w;
        *reinterpret_cast<f32x4 *>(&Bs[buf][b_row * GM_BN + b_col]) = rb;
    };

    f32x16 acc, total;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; total[i] = 0.f; }

    const int nk = (ke - kb + GM_BK - 1) / GM_BK;
    load_tiles(kb);
    store_tiles(0);
    __syncthreads();

    const int a_frag = (wr * 32 + (lane & 31)) * GM_LDA + (lane >> 5);
    const int b_frag = (lane >> 5) * GM_BN + wc * 32 + (lane & 31);
    constexpr int FOLD_TILES = LT_GEMM_FOLD / GM_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kb + (kt + 1) * GM_BK);
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[kk], bs[kk * GM_BN], acc, 0, 0, 0);
        if ((kt + 1) % FOLD_TILES == 0) {     // LT_GEMM_FOLD terms: the chain's sum joins the total, the next chain starts from +0
            total += acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
    total += acc;
    acc = total;

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int cn = n0 + wc * 32 + (lane & 31);
    if (cn < N) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int cm = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (cm < M) C[(long)cm * ldc + cn] = acc[reg];
        }
    }
}


// ---- 64x64 block tile, 64-deep k-tiles: the probe-row product X'[probes] * W1 (M = probes of a chunk, a
// few hundred rows) split into 256-deep K slices.  With 16-deep tiles a slice was 16 load -> 8 MFMA -> barrier
// round trips, each bound by the latency of its own loads (0.2 us of MFMAs against > 1 us of latency: 16-20 us
// per call); a 64-deep tile carries 32 MFMAs per wave (0.85 us) behind one round of loads, the next tile's
// loads are in flight meanwhile, and a slice is 4 trips.  Requires N % 64 == 0 (full B tiles); rows past M
// are clamped (they feed output rows that are never stored).  Same k-ordered chains, same bits.
#ifdef LT_GEMM_TRACE
__device__ unsigned long long *g_lt_gemm_trace = nullptr;
#endif
#define GD_BK 64
#define GD_LDA (GD_BK + 1)
#define GD_PASS 4   // 256 threads x float4 cover 16 rows x 64 floats per pass
template <bool GATHER>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma_deep(const float *__restrict__ A, long lda,
                                                            const float *__restrict__ B, long ldb,
                                                            float *__restrict__ C, long ldc, int M, int N,
                                                            int K, int kslice, long slab_stride,
                                                            const int32_t *__restrict__ rows, float delta) {
#ifdef LT_GEMM_TRACE
    const unsigned lt_trace_id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && g_lt_gemm_trace) {
        g_lt_gemm_trace[3 * lt_trace_id] = wall_clock64();
        g_lt_gemm_trace[3 * lt_trace_id + 2] = 0;
    }
#endif
    __shared__ __attribute__((aligned(16))) float As[2][GM_BM * GD_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GD_BK * GM_BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GM_BM, n0 = blockIdx.y * GM_BN;
    const int kb = blockIdx.z * kslice, ke = min(K, kb + kslice);
    C += (long)blockIdx.z * slab_stride;

    // staging: thread -> row t_row (+16 per pass), 4 floats at t_col, for both the A tile (64 x 64) and the
    // B tile (64 x 64)
    const int t_row = tid >> 4, t_col = (tid & 15) * 4;
    const float *a_ptr[GD_PASS], *b_ptr[GD_PASS];
#pragma unroll
    for (int p = 0; p < GD_PASS; ++p) {
        const int m = min(m0 + t_row + 16 * p, M - 1);
        const long src = GATHER ? (long)rows[m] : (long)m;
        a_ptr[p] = A + src * lda + t_col + kb;
        b_ptr[p] = B + (long)(kb + t_row + 16 * p) * ldb + n0 + t_col;
    }
    const long b_step = (long)GD_BK * ldb;
    auto perturb = [&](f32x4 v) {
        if (GATHER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __fadd_rn(v[j], __fmul_rn(v[j], delta));   // two roundings, no fma
        }
        return v;
    };
    f32x4 ra[GD_PASS], rb[GD_PASS];
    auto load_full = [&]() {      // the next k-tile, all of it inside [kb, ke)
#pragma unroll
        for (int p = 0; p < GD_PASS; ++p) {
            ra[p] = *reinterpret_cast<const f32x4u *>(a_ptr[p]);
            rb[p] = *reinterpret_cast<const f32x4u *>(b_ptr[p]);
            a_ptr[p] += GD_BK;
            b_ptr[p] += b_step;
        }
    };
    auto load_tail = [&](int k0) {      // the slice's last, partial k-tile: zero-filled past ke
#pragma unroll
        for (int p = 0; p < GD_PASS; ++p) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + t_col + j < ke) r[j] = a_ptr[p][j];
            ra[p] = r;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k0 + t_row + 16 * p < ke) t = *reinterpret_cast<const f32x4u *>(b_ptr[p]);
            rb[p] = t;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < GD_PASS; ++p) {
            const f32x4 v = perturb(ra[p]);
            float *as = &As[buf][(t_row + 16 * p) * GD_LDA + t_col];
            as[0] = v.x; as[1] = v.y; as[2] = v.z; as[3] = v.w;
            *reinterpret_cast<f32x4 *>(&Bs[buf][(t_row + 16 * p) * GM_BN + t_col]) = rb[p];
        }
    };
    f32x16 acc, total;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; total[i] = 0.f; }
    constexpr int FOLD_TILES = LT_GEMM_FOLD / GD_BK;
    const int a_frag = (wr * 32 + (lane & 31)) * GD_LDA + (lane >> 5);
    const int b_frag = (lane >> 5) * GM_BN + wc * 32 + (lane & 31);
    auto multiply = [&](int buf) {
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
#pragma unroll
        for (int kk = 0; kk < GD_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[kk], bs[kk * GM_BN], acc, 0, 0, 0);
    };
    const int nfull = (ke - kb) / GD_BK;
    const bool partial = (ke - kb) % GD_BK != 0;
    const int nk = nfull + (partial ? 1 : 0);
    auto load_any = [&](int t) { if (t < nfull) load_full(); else load_tail(kb + t * GD_BK); };
    if (nk > 0) {
        load_any(0);
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_any(kt + 1);
            multiply(buf);
            if ((kt + 1) % FOLD_TILES == 0) {     // (the fold of the header comment: same k as the other kernels)
                total += acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            __syncthreads();
        }
    }
    total += acc;
    acc = total;
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int cn = n0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int cm = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (cm < M) C[(long)cm * ldc + cn] = acc[reg];
    }
#ifdef LT_GEMM_TRACE
    if (threadIdx.x == 0 && g_lt_gemm_trace) g_lt_gemm_trace[3 * lt_trace_id + 1] = wall_clock64();
#endif
}


// ---- 128x128 block tile: each of the 4 waves owns a 64x64 quadrant = 2x2 MFMA tiles, so one A and
// one B fragment feed two MFMAs each (half the LDS reads per flop of the 64x64 kernel) and a k-tile
// carries 32 MFMAs per wave (2048 cycles).  Global loads run two k-tiles ahead of the LDS stores in two
// register stages, LDS fragment reads one k-step ahead of the MFMAs; the steady-state loop is branch-free
// so that hipcc's s_waitcnt are counted (vmcnt(7..4), lgkmcnt(3)) instead of drains.  Used for the big
// X*W1 product (N a multiple of 128); same k-ordered fmaf chains per output.  Measured on twitch-RU
// (4385 x 3170 x 256, 7 K-slices): 71 us = 100 TFLOP/s, 2/3 of the 150 TFLOP/s a pure MFMA stream reaches
// on this part (tools/fold_test/mfma_peak.hip); the LDS store + barrier per k-tile is the largest rest.
#define GL_BM 128
#define GL_BN 128
#ifndef GL_BK
#define GL_BK 16        // k-depth of a tile (32 measured 8 % slower, also with the two-stage prefetch)
#endif
#define GL_LDA (GL_BK + 1)
#define GL_PASS (GL_BK / 8)   // staging passes: 256 threads x float4 cover 128 x 8 (A) or 8 x 128 (B) floats
__global__ __launch_bounds__(256) void k_gemm_f32_mfma_128(const float *__restrict__ A, long lda,
                                                           const float *__restrict__ B, long ldb,
                                                           float *__restrict__ C, long ldc, int M, int N,
                                                           int K, int kslice, long slab_stride) {
    // requires N % 128 == 0 (every B tile is full; the launcher falls back to the 64x64 kernel otherwise)
#ifdef LT_GEMM_TRACE   // tools/gemm_lab: start / end time of every workgroup (constant 100 MHz clock)
    const unsigned lt_trace_id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && g_lt_gemm_trace) {
        g_lt_gemm_trace[3 * lt_trace_id] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_lt_gemm_trace[3 * lt_trace_id + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
#endif
    __shared__ __attribute__((aligned(16))) float As[2][GL_BM * GL_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GL_BK * GL_BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GL_BM, n0 = blockIdx.y * GL_BN;
    const int kb = blockIdx.z * kslice, ke = min(K, kb + kslice);
    C += (long)blockIdx.z * slab_stride;

    // staging: A tile 128 x BK -> thread owns row a_row (+ 128/PASS per pass), 4 floats at a_col;
    //          B tile BK x 128 -> thread owns row b_row (+ 8 per pass), 4 floats at b_col.
    // Rows past M are clamped to the last row: they only feed output rows that are never stored, and the
    // loads of a full k-tile stay unconditional -- straight-line code whose outstanding loads the compiler
    // can COUNT (vmcnt(n)); any branch in there and it drains vmcnt(0), i.e. kills the prefetch.
    constexpr int A_TPR = GL_BK / 4;            // threads per A row
    constexpr int A_RPP = 256 / A_TPR;          // A rows per pass
    const int a_row = tid / A_TPR, a_col = (tid % A_TPR) * 4;
    const int b_row = tid >> 5, b_col = (tid & 31) * 4;
    // The tile pointers walk along K (tiles are loaded strictly in order): no per-load address temporaries,
    // which hipcc would alias with the load destinations and then guard with s_waitcnt vmcnt(0).
    const float *a_ptr[GL_PASS], *b_ptr[GL_PASS];
#pragma unroll
    for (int p = 0; p < GL_PASS; ++p) {
        a_ptr[p] = A + (long)min(m0 + a_row + p * A_RPP, M - 1) * lda + a_col + kb;
        b_ptr[p] = B + (long)(kb + b_row + 8 * p) * ldb + n0 + b_col;
    }
    const long b_step = (long)GL_BK * ldb;

    // two register stages: tile t is loaded into stage t & 1 TWO iterations before it is stored to LDS.  One
    // k-tile of MFMAs is 2048 cycles (0.85 us), less than the latency of a loaded HBM system.
    f32x4 ra[2][GL_PASS], rb[2][GL_PASS];
    auto load_full = [&](auto stage_tag) {      // the next k-tile, all of it inside [kb, ke)
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GL_PASS; ++p) {
            ra[S][p] = *reinterpret_cast<const f32x4u *>(a_ptr[p]);
            rb[S][p] = *reinterpret_cast<const f32x4u *>(b_ptr[p]);
            a_ptr[p] += GL_BK;
            b_ptr[p] += b_step;
        }
    };
    auto load_tail = [&](int k0, auto stage_tag) {      // the slice's last, partial k-tile: zero-filled past ke
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GL_PASS; ++p) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + a_col + j < ke) r[j] = a_ptr[p][j];
            ra[S][p] = r;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k0 + b_row + 8 * p < ke) t = *reinterpret_cast<const f32x4u *>(b_ptr[p]);
            rb[S][p] = t;
        }
    };
    auto store_tiles = [&](int buf, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
#pragma unroll
        for (int p = 0; p < GL_PASS; ++p) {
            float *as = &As[buf][(a_row + p * A_RPP) * GL_LDA + a_col];
            as[0] = ra[S][p].x; as[1] = ra[S][p].y; as[2] = ra[S][p].z; as[3] = ra[S][p].w;
            *reinterpret_cast<f32x4 *>(&Bs[buf][(b_row + 8 * p) * GL_BN + b_col]) = rb[S][p];
        }
    };

    f32x16 acc[2][2], total[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; total[i][j][e] = 0.f; }
    constexpr int FOLD_TILES = LT_GEMM_FOLD / GL_BK;
    static_assert((FOLD_TILES & 1) == 0, "the steady-state loop folds after the second tile of a trip");
    // the fold of the header comment: after tile t with (t + 1) % FOLD_TILES == 0
    auto fold = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                total[i][j] += acc[i][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            }
    };

    const int a_frag = (wr * 64 + (lane & 31)) * GL_LDA + (lane >> 5);
    const int b_frag = (lane >> 5) * GL_BN + wc * 64 + (lane & 31);
    // one k-tile out of LDS buffer `buf`; the fragments of step kk+2 are read while step kk multiplies
    auto multiply = [&](int buf) {
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
        float a0 = as[0], a1 = as[32 * GL_LDA], b0 = bs[0], b1 = bs[32];
#pragma unroll
        for (int kk = 0; kk < GL_BK; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < GL_BK) {
                na0 = as[kk + 2]; na1 = as[32 * GL_LDA + kk + 2];
                nb0 = bs[(kk + 2) * GL_BN]; nb1 = bs[(kk + 2) * GL_BN + 32];
            }
            // keep the reads above in front of the MFMAs below (hipcc sinks them to just before their use,
            // which puts an LDS round trip in front of every group of four MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nfull = (ke - kb) / GL_BK;               // full k-tiles; a partial one may follow
    const bool partial = (ke - kb) % GL_BK != 0;
    if (nfull > 0) {
        load_full(S0{});
        store_tiles(0, S0{});
        if (nfull > 1) load_full(S1{});
        __syncthreads();
        // iteration kt: tile kt+2 starts loading into stage kt & 1, tile kt is multiplied out of LDS buffer
        // kt & 1, tile kt+1 (stage (kt+1) & 1, loaded an iteration ago) goes to the other LDS buffer
        auto step = [&](int kt, auto even_tag) {
            constexpr int E = decltype(even_tag)::value;   // kt & 1
            if (kt + 2 < nfull) load_full(std::integral_constant<int, E>{});
            multiply(E);
            if ((kt + 1) % FOLD_TILES == 0) fold();
            if (kt + 1 < nfull) store_tiles(E ^ 1, std::integral_constant<int, E ^ 1>{});
            __syncthreads();
        };
        // steady state, two tiles per trip, everything unconditional: with a branch around the loads hipcc no
        // longer knows how many loads are younger than the stage it stores, and waits for all of them
        int kt = 0;
        for (; kt + 3 < nfull; kt += 2) {
            load_full(S0{});
            multiply(0);
            store_tiles(1, S1{});
            __syncthreads();
            load_full(S1{});
            multiply(1);
            if ((kt + 2) % FOLD_TILES == 0) fold();     // (register-only: the counted waits of the loads above are unaffected)
            store_tiles(0, S0{});
            __syncthreads();
        }
        // the last one to three full tiles
        for (; kt + 1 < nfull; kt += 2) {
            step(kt, S0{});
            step(kt + 1, S1{});
        }
        if (kt < nfull) step(kt, S0{});
    }
    if (partial) {
        load_tail(kb + nfull * GL_BK, S0{});
        store_tiles(0, S0{});
        __syncthreads();
        multiply(0);
    }
    // (a partial tile is the slice's last: its terms belong to the chain the final fold closes)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            total[i][j] += acc[i][j];
            acc[i][j] = total[i][j];
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cn = n0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int cm = m0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (cm < M) C[(long)cm * ldc + cn] = acc[i][j][reg];
            }
        }
#ifdef LT_GEMM_TRACE
    if (threadIdx.x == 0 && g_lt_gemm_trace) g_lt_gemm_trace[3 * lt_trace_id + 1] = wall_clock64();
#endif
}

int lt_launch_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                   int M, int N, int K, hipStream_t st) {
    if (M == 0 || N == 0) return LT_OK;
    lt_prof_scope prof_(LT_K_GEMM, st);
    if (M >= 1024 && N % GL_BN == 0 && K > 0) {
        // many rows, one slice (a 2 M-node graph with F = 256): the 128 x 128 tiles of the split-K product, same k order
        dim3 gridl((M + GL_BM - 1) / GL_BM, N / GL_BN);
        hipLaunchKernelGGL(k_gemm_f32_mfma_128, gridl, dim3(256), 0, st, A, (long)lda, B, (long)ldb, C, (long)ldc, M, N, K,
                           K, 0L);
        LT_CHECK_LAUNCH();
        return LT_OK;
    }
    dim3 grid((M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN);
    LT_REQUIRE(grid.y <= 65535u, "lt_gemm_f32: N=%d too large", N);
    hipLaunchKernelGGL(k_gemm_f32_mfma<false>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, C,
                       (long)ldc, M, N, K, K > 0 ? K : 1, 0L, (const int32_t *)nullptr, 0.f, (const int32_t *)nullptr);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// C[0 .. *m_dev) = A B with the row count on the device: the grid covers m_bound rows (64 x 64 tiles, one K slice),
// tiles past the count exit.  Rows are summed exactly as lt_launch_gemm sums them.
int lt_launch_gemm_mdev(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                        int m_bound, const int32_t *m_dev, int N, int K, hipStream_t st) {
    if (m_bound == 0 || N == 0) return LT_OK;
    dim3 grid((m_bound + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN);
    LT_REQUIRE(grid.y <= 65535u, "lt_gemm: N=%d too large", N);
    lt_prof_scope prof_(LT_K_GEMM, st);
    hipLaunchKernelGGL(k_gemm_f32_mfma<false>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, C,
                       (long)ldc, m_bound, N, K, K > 0 ? K : 1, 0L, (const int32_t *)nullptr, 0.f, m_dev);
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// Sum of `splits` partial slabs in slab order (fixed order: deterministic, independent of M).
__global__ void k_sum_slabs(const float *__restrict__ slabs, long slab_stride, int splits, int M, int N,
                            long ld, float *__restrict__ C, long ldc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N) return;
    const int r = (int)(i / N), c = (int)(i % N);
    float acc = slabs[(long)r * ld + c];
    for (int z = 1; z < splits; ++z) acc += slabs[(long)z * slab_stride + (long)r * ld + c];
    C[(long)r * ldc + c] = acc;
}

// Slice length for the big (128x128-tile) split-K product: pick the slice count whose workgroups fill the
// 256 CUs in whole rounds (a 2.2-blocks-per-CU grid runs as long as a 3-per-CU one), charging each extra
// slab ~2 tile-steps of write + re-read.  Deterministic in (M, N, K); multiples of 16.
int lt_gemm_pick_kslice(int M, int N, int K) {
    const long tiles = (long)((M + GL_BM - 1) / GL_BM) * ((N + GL_BN - 1) / GL_BN);
    int best_s = 1;
    double best = 1e30;
    for (int s = 1; s <= 16; ++s) {
        const int ks = ((K + s - 1) / s + GL_BK - 1) / GL_BK * GL_BK;
        if (ks <= 0) break;
        const int slices = (K + ks - 1) / ks;
        const long rounds = (tiles * slices + 255) / 256;
        const double cost = (double)rounds * (ks / 16) + 2.2 * (slices - 1);   // in 16-deep tile-steps
        if (cost < best - 1e-9) { best = cost; best_s = s; }
    }
    const int ks = ((K + best_s - 1) / best_s + GL_BK - 1) / GL_BK * GL_BK;
    return ks > 0 ? ks : GL_BK;
}

size_t lt_gemm_splitk_slab_bytes(int M, int N, int K, int kslice) {
    const int splits = (K + kslice - 1) / kslice;
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

// Split-K GEMM: K is cut into fixed `kslice`-deep slices (a multiple of 16) so that
// M/64 x N/64 x K/kslice blocks put several workgroups on every CU (one 64x64 workgroup per CU is
// bound by the latency of its own tile loads); each output is the ordered sum of its slices, so
// a row's result does not depend on how many rows the call carries.
int lt_launch_gemm_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                          int M, int N, int K, int kslice, float *slabs, hipStream_t st,
                          const int32_t *gather_rows, float delta) {
    if (M == 0 || N == 0) return LT_OK;
    const int splits = (K + kslice - 1) / kslice;
    if (splits <= 1 && !gather_rows) return lt_launch_gemm(A, lda, B, ldb, C, ldc, M, N, K, st);
    const bool big = !gather_rows && M >= 1024 && N % GL_BN == 0;
    dim3 grid(big ? (M + GL_BM - 1) / GL_BM : (M + GM_BM - 1) / GM_BM,
              big ? (N + GL_BN - 1) / GL_BN : (N + GM_BN - 1) / GM_BN, splits);
    LT_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, "split-K GEMM: N=%d or K=%d too large", N, K);
    const long stride = (long)M * N;
    // a single slice writes C directly (no slab, no sum)
    float *dst = splits > 1 ? slabs : C;
    const long ldd = splits > 1 ? (long)N : (long)ldc;
    {
        lt_prof_scope prof_(LT_K_GEMM, st);
        if (big)
            hipLaunchKernelGGL(k_gemm_f32_mfma_128, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, dst, ldd,
                               M, N, K, kslice, stride);
        else if (N % GM_BN == 0 && gather_rows)
            hipLaunchKernelGGL(k_gemm_f32_mfma_deep<true>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, dst,
                               ldd, M, N, K, kslice, stride, gather_rows, delta);
        else if (N % GM_BN == 0)
            hipLaunchKernelGGL(k_gemm_f32_mfma_deep<false>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, dst,
                               ldd, M, N, K, kslice, stride, (const int32_t *)nullptr, 0.f);
        else if (gather_rows)
            hipLaunchKernelGGL(k_gemm_f32_mfma<true>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, dst, ldd,
                               M, N, K, kslice, stride, gather_rows, delta, (const int32_t *)nullptr);
        else
            hipLaunchKernelGGL(k_gemm_f32_mfma<false>, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, dst, ldd,
                               M, N, K, kslice, stride, (const int32_t *)nullptr, 0.f, (const int32_t *)nullptr);
        LT_CHECK_LAUNCH();
        if (splits > 1) {
            const long tot = (long)M * N;
            hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slabs, stride,
                               splits, M, N, (long)N, C, (long)ldc);
            LT_CHECK_LAUNCH();
        }
    }
    return LT_OK;
}

extern "C" int lt_gemm_f32(const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                           int64_t ldc, int32_t M, int32_t N, int32_t K, void *stream) {
    LT_REQUIRE(M >= 0 && N >= 0 && K >= 0, "lt_gemm_f32: negative dimension");
    LT_REQUIRE(A && B && C, "lt_gemm_f32: NULL pointer");
    LT_REQUIRE(lda >= K && ldb >= N && ldc >= N, "lt_gemm_f32: leading dimension too small");
    return lt_launch_gemm(A, lda, B, ldb, C, ldc, M, N, K, (hipStream_t)stream);
}
