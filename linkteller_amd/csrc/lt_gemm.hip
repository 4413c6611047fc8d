// Dense fp32 GEMM C[M,N] = A[M,K] * B[K,N] on the gfx950 matrix cores, exact fp32.
//   reference call site: torch.mm(input, self.weight), gcn/layers.py:31 -- the one
//   GEMM-shaped op on the path (X*W1: [N,3170]x[3170,256]); everything else is sparse.
//
// v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate; each output element is a k-ordered fmaf
// chain (no reduced-precision path exists on gfx950), 64 FLOP/clk/SIMD = the fp32 peak.
// Block = 4 waves as 2x2, each wave owns one 32x32 accumulator tile of a 64x64 block tile;
// K is walked in 16-deep tiles staged through LDS with a register prefetch of the next tile
// (one barrier per tile).  A-tile rows are padded to 17 floats: the MFMA A operand is read
// "column-wise" (lane -> row) and 17 is odd, so the 32 lanes of a half hit 32 distinct banks.
#include "lt_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GM_BM 64
#define GM_BN 64
#define GM_BK 16
#define GM_LDA (GM_BK + 1)

__global__ __launch_bounds__(256) void k_gemm_f32_mfma(const float *__restrict__ A, long lda,
                                                       const float *__restrict__ B, long ldb,
                                                       float *__restrict__ C, long ldc, int M, int N,
                                                       int K, int kslice, long slab_stride) {
    __shared__ __attribute__((aligned(16))) float As[2][GM_BM * GM_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][GM_BK * GM_BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int m0 = blockIdx.x * GM_BM;  // M tiles on x: neighbours share the B (weight) tiles in L2
    const int n0 = blockIdx.y * GM_BN;

    // staging coordinates: A tile 64 x 16 (thread -> row tid/4, 4 floats), B tile 16 x 64
    const int a_row = tid >> 2, a_col = (tid & 3) * 4;
    const int b_row = tid >> 4, b_col = (tid & 15) * 4;
    const bool a_row_ok = (m0 + a_row) < M;
    const float *a_ptr = A + (long)(m0 + a_row) * lda + a_col;
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

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int nk = (ke - kb + GM_BK - 1) / GM_BK;
    load_tiles(kb);
    store_tiles(0);
    __syncthreads();

    const int a_frag = (wr * 32 + (lane & 31)) * GM_LDA + (lane >> 5);
    const int b_frag = (lane >> 5) * GM_BN + wc * 32 + (lane & 31);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kb + (kt + 1) * GM_BK);
        const float *as = &As[buf][a_frag];
        const float *bs = &Bs[buf][b_frag];
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[kk], bs[kk * GM_BN], acc, 0, 0, 0);
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

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

int lt_launch_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                   int M, int N, int K, hipStream_t st) {
    if (M == 0 || N == 0) return LT_OK;
    dim3 grid((M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN);
    LT_REQUIRE(grid.y <= 65535u, "lt_gemm_f32: N=%d too large", N);
    lt_prof_scope prof_(LT_K_GEMM, st);
    hipLaunchKernelGGL(k_gemm_f32_mfma, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, C,
                       (long)ldc, M, N, K, K > 0 ? K : 1, 0L);
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

size_t lt_gemm_splitk_slab_bytes(int M, int N, int K, int kslice) {
    const int splits = (K + kslice - 1) / kslice;
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

// Split-K GEMM: K is cut into fixed `kslice`-deep slices (a multiple of 16) so that
// M/64 x N/64 x K/kslice blocks put several workgroups on every CU (one 64x64 workgroup per CU is
// bound by the latency of its own tile loads); each output is the ordered sum of its slices, so
// a row's result does not depend on how many rows the call carries.
int lt_launch_gemm_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                          int M, int N, int K, int kslice, float *slabs, hipStream_t st) {
    if (M == 0 || N == 0) return LT_OK;
    const int splits = (K + kslice - 1) / kslice;
    if (splits <= 1) return lt_launch_gemm(A, lda, B, ldb, C, ldc, M, N, K, st);
    dim3 grid((M + GM_BM - 1) / GM_BM, (N + GM_BN - 1) / GM_BN, splits);
    const long stride = (long)M * N;
    {
        lt_prof_scope prof_(LT_K_GEMM, st);
        hipLaunchKernelGGL(k_gemm_f32_mfma, grid, dim3(256), 0, st, A, (long)lda, B, (long)ldb, slabs,
                           (long)N, M, N, K, kslice, stride);
        LT_CHECK_LAUNCH();
        const long tot = (long)M * N;
        hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slabs, stride,
                           splits, M, N, (long)N, C, (long)ldc);
        LT_CHECK_LAUNCH();
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
