// The fp64-grade product S1d = X * W1 of dense features on the INT8 matrix cores of gfx950 (round 5: a lab; round 6: the product route
// "i8_split" of lt_fp64.hip's launch_dense_s1d)
// (v_mfma_i32_32x32x32_i8: 2x the bf16 rate, ~60x the f64 rate) -- the error-free-split form of the product LT_MODE_DELTA's kink
// test needs (reference: gcn/layers.py:31, `support = torch.mm(input, self.weight)`; round-4 review item 5).  Measurements:
// profiles/r05_i8_split_lab.txt, NOTES.md (rounds 5 and 6).
//
//   * per (row, K slice) of X and per (column, K slice) of W1 one power-of-two scale; X becomes a 39-bit fixed-point integer
//     (FIVE signed base-256 digits: an fp32 within 2^-15 of its row's largest value is represented exactly), W1 a 31-bit one (four);
//   * q_x * q_w = sum_{i,j} dx_i dw_j 256^(i+j): the FOURTEEN digit pairs of order i + j >= 3 go through the matrix cores, one int32
//     accumulator per ORDER (products <= 2^14, K < 2^15 deep, <= 4 pairs: below 2^31, so the integer sums are EXACT); the orders
//     below carry < 2^-38 of a term of full scale;
//   * the accumulators of an element combine exactly in int64, convert to fp64 with one rounding and are scaled by the power of
//     two: one fp64 partial per K slice (split-K slabs as lt_fp64.hip sums them).  Integer sums do not depend on their order: a
//     row has the same bits whichever tile, wave or rank formed it.
//
// Two kernels: k_gemm_i8split (register-staged tiles, 64 x 128 per workgroup of 4 waves, two workgroups per CU) and
// k_gemm_i8split3 (64 x 256, 8 waves, tiles by LDS-DMA two steps ahead, counted vmcnt, one barrier per step).
#ifdef LT_I8_LAB        // tools/i8_lab/i8_lab.hip: the lab supplies the few helpers of lt_internal.h itself
#include "i8_lab_shim.h"
#else
#include "lt_internal.h"
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define I8_BM 64
#define I8_BN 128
#define I8_KS 32            // K columns per step (one MFMA deep)
#define I8_XD 5             // digits of X (39-bit fixed point: an fp32 within 2^-15 of its row's largest value is EXACT)
#define I8_WD 4             // digits of W1 (31-bit fixed point)
#define I8_A_BYTES (I8_XD * I8_BM * I8_KS)
#define I8_B_BYTES (I8_WD * I8_BN * I8_KS)
#define I8_LDS_BYTES (2 * (I8_A_BYTES + I8_B_BYTES))

// e with |x| < 2^e for the largest |x| whose bits are `mbits` (clamped so that the scales stay normal numbers); non-finite -> INT_MIN
__device__ __forceinline__ int i8_exponent(unsigned mbits) {
    if (mbits >= 0x7f800000u) return INT_MIN;
    int e = (int)(mbits >> 23) - 126;
    return e < -96 ? -96 : e;
}
// W1: the four signed base-256 digits of rint(x * 2^(30 - e)), as the bytes of one word
__device__ __forceinline__ unsigned i8_digits4(float x, float sc) {
    const int q = (int)rintf(x * sc);
    return ((unsigned)q + 0x80808080u - 0x80000000u) ^ 0x00808080u;
}
// X: the five signed base-256 digits of rint(x * 2^(38 - e)): the low four as the bytes of `lo`, the fifth returned
__device__ __forceinline__ unsigned i8_digits5(float x, double sc, unsigned &lo) {
    const long long q = (long long)rint((double)x * sc);
    const long long y = q + 0x80808080ll;
    lo = (unsigned)y ^ 0x80808080u;
    return (unsigned)(y >> 32) & 0xffu;
}
// 4 words of 4 digits -> 4 words, word i = digit i of the four values
__device__ __forceinline__ void i8_transpose(unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned out[4]) {
    const unsigned t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
    const unsigned t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// ewb[slice][Hc] (zeroed) <- the largest |W1[k, c]| (as bits) over the slice's k.  grid (steps, Hc / 64), 256 threads = 64 columns x 4.
__global__ __launch_bounds__(256) void k_i8_w_max(const float *__restrict__ W, int F, int H, int Hc, int steps_per_slice,
                                                  unsigned *__restrict__ ewb) {
    __shared__ unsigned s_mx[4][64];
    const int stp = blockIdx.x, cl = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    unsigned mx = 0;
    if (c < H) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = stp * I8_KS + kq * 8 + u;
            if (k < F) mx = max(mx, __float_as_uint(W[(size_t)k * H + c]) & 0x7fffffffu);
        }
    }
    s_mx[kq][cl] = mx;
    __syncthreads();
    if (kq == 0) atomicMax(ewb + (size_t)(stp / steps_per_slice) * Hc + c, max(max(s_mx[0][cl], s_mx[1][cl]), max(s_mx[2][cl], s_mx[3][cl])));
}
// W1[F, H] (row-major, ld = H) -> Wd[step][digit][k half][Hc][16 k].  Same grid: thread (column, kq) cuts the 8 k of kq.
__global__ __launch_bounds__(256) void k_i8_w_digits(const float *__restrict__ W, int F, int H, int Hc, int steps_per_slice,
                                                     const unsigned *__restrict__ ewb, int8_t *__restrict__ Wd) {
    const int stp = blockIdx.x, cl = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int e = i8_exponent(ewb[(size_t)(stp / steps_per_slice) * Hc + c]);
    const float sc = e == INT_MIN ? 0.f : __uint_as_float((unsigned)(30 - e + 127) << 23);
    unsigned w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = stp * I8_KS + kq * 8 + u;
        const float x = (c < H && k < F) ? W[(size_t)k * H + c] : 0.f;
        w[u] = e == INT_MIN ? 0u : i8_digits4(x, sc);
    }
    unsigned lo[4], hi[4];
    i8_transpose(w[0], w[1], w[2], w[3], lo);
    i8_transpose(w[4], w[5], w[6], w[7], hi);
#pragma unroll
    for (int d = 0; d < I8_WD; ++d)
        *reinterpret_cast<uint2 *>(Wd + ((((size_t)stp * I8_WD + d) * 2 + (kq >> 1)) * Hc + c) * 16 + (kq & 1) * 8) = make_uint2(lo[d], hi[d]);
}

// AV: the alignment of X's rows in floats (4 / 2 / 1: ldx and the base pointer), i.e. the widest load a thread's eight k allow
template <int AV>
__device__ __forceinline__ void i8_load8(const float *p, int k, int F, float v[8]) {
    if (k + 7 < F) {
        if (AV == 4) {
            const float4 a = *reinterpret_cast<const float4 *>(p + k), b = *reinterpret_cast<const float4 *>(p + k + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else if (AV == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float2 a = *reinterpret_cast<const float2 *>(p + k + 2 * u);
                v[2 * u] = a.x; v[2 * u + 1] = a.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[k + u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u < F ? p[k + u] : 0.f;
    }
}

// slabs[slice][row - r0][col] (ld = H) = the slice's share of X[row, :] * W1[:, col], rows [r0, r0 + m).  grid (row tiles, column
// blocks, slices), 256 threads (2 x 2 waves of 32 x 64), I8_LDS_BYTES of dynamic LDS: two workgroups share a CU, one computes while
// the other waits for its tiles.  ORD_MIN: the digit pairs of order i + j >= ORD_MIN go through the matrix cores (3: fourteen pairs).
template <int AV, int ORD_MIN, int VAR = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_i8split(const float *__restrict__ X, long ldx, int m, int F, int H, int Hc,
                                                         const int8_t *__restrict__ Wd, const unsigned *__restrict__ ewb, int steps,
                                                         int steps_per_slice, double *__restrict__ slabs, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char i8_lds[];
    __shared__ int s_ex[I8_BM];
    unsigned char *As = i8_lds;                       // [2][5 digits][2 k halves][64 rows][16]
    unsigned char *Bs = i8_lds + 2 * I8_A_BYTES;      // [2][4 digits][2 k halves][128 columns][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * I8_BM, c0 = blockIdx.y * I8_BN, s = blockIdx.z;
    const int bn = min(I8_BN, Hc - c0);               // columns of this block (64 or 128)
    const int st0 = s * steps_per_slice, st1 = min(st0 + steps_per_slice, steps);
    const int nst = st1 - st0;

    // this thread's share of an X tile: row a_row, the eight k of a_k8 (rows past m read row m - 1: never stored)
    const int a_row = tid >> 2, a_k8 = (tid & 3) * 8;
    const float *xrow = X + (size_t)min(m0 + a_row, m - 1) * ldx;
    float ra[8];
    // the scale of (row, slice): the largest |x| over the slice's columns
    {
        // (eight steps' loads in flight at a time: the running maximum must not put a memory round trip on every step)
        unsigned mx = 0;
        for (int sb = st0; sb < st1; sb += 8) {
            float v[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) i8_load8<AV>(xrow, min(sb + u, st1 - 1) * I8_KS + a_k8, F, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) mx = max(mx, __float_as_uint(v[u][q]) & 0x7fffffffu);
        }
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 1, 64));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 2, 64));
        if ((tid & 3) == 0) s_ex[a_row] = i8_exponent(mx);
    }
    __syncthreads();
    const int a_e = s_ex[a_row];
    const double a_sc = a_e == INT_MIN ? 0.0 : __longlong_as_double((long long)(38 - a_e + 1023) << 52);

    i32x4 rb[I8_WD];
    const int b_half = tid / bn, b_col = tid - b_half * bn;       // (tid < 2 bn) this thread's 16 bytes of a W1 digit tile
    auto load_tiles = [&](int st) {
        if (VAR != 3) i8_load8<AV>(xrow, st * I8_KS + a_k8, F, ra);
#pragma unroll
        for (int j = 0; j < I8_WD; ++j)
            if (tid < bn * 2 && (VAR != 2 || st == st0)) rb[j] = *reinterpret_cast<const i32x4 *>(Wd + ((((size_t)st * I8_WD + j) * 2 + b_half) * Hc + c0 + b_col) * 16);
    };
    auto store_tiles = [&](int buf) {
        if (VAR == 4 && buf) return;
        unsigned w[8], top[8], lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 8; ++u) top[u] = i8_digits5(ra[u], a_sc, w[u]);
        i8_transpose(w[0], w[1], w[2], w[3], lo);
        i8_transpose(w[4], w[5], w[6], w[7], hi);
        // LDS images [digit][k half][row or column][16 k]: a wave's operand read is 32 x 16 contiguous bytes per half
        unsigned char *a = As + buf * I8_A_BYTES + (tid & 2 ? I8_BM * 16 : 0) + a_row * 16 + (tid & 1) * 8;
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<uint2 *>(a + d * (I8_BM * I8_KS)) = make_uint2(lo[d], hi[d]);
        *reinterpret_cast<uint2 *>(a + 4 * (I8_BM * I8_KS)) =
            make_uint2(top[0] | top[1] << 8 | top[2] << 16 | top[3] << 24, top[4] | top[5] << 8 | top[6] << 16 | top[7] << 24);
        unsigned char *bsm = Bs + buf * I8_B_BYTES;
#pragma unroll
        for (int j = 0; j < I8_WD; ++j)
            if (tid < bn * 2) *reinterpret_cast<i32x4 *>(bsm + j * (I8_BN * I8_KS) + b_half * (I8_BN * 16) + b_col * 16) = rb[j];
    };

    constexpr int OMAX = I8_XD + I8_WD - 2, NORD = OMAX + 1 - ORD_MIN;
    i32x16 acc[NORD][2];      // [order OMAX, OMAX - 1, ... ORD_MIN][column tile]
#pragma unroll
    for (int o = 0; o < NORD; ++o)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[o][t][i] = 0;

    const bool active = wc * 64 < bn;
    if (nst > 0) {
        load_tiles(st0);
        store_tiles(0);
    }
    __syncthreads();
    const int a_off = (lane >> 5) * (I8_BM * 16) + (wr * 32 + (lane & 31)) * 16;
    const int b_off = (lane >> 5) * (I8_BN * 16) + (wc * 64 + (lane & 31)) * 16;
    for (int it = 0; it < nst; ++it) {
        const int buf = it & 1;
        if (it + 1 < nst) load_tiles(st0 + it + 1);
        if (active && VAR != 1) {
            const unsigned char *a = As + buf * I8_A_BYTES + a_off;
            const unsigned char *bq = Bs + buf * I8_B_BYTES + b_off;
            i32x4 av[I8_XD], bv[2][I8_WD];
#pragma unroll
            for (int d = 0; d < I8_XD; ++d) av[d] = *reinterpret_cast<const i32x4 *>(a + d * (I8_BM * I8_KS));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int d = 0; d < I8_WD; ++d) bv[t][d] = *reinterpret_cast<const i32x4 *>(bq + d * (I8_BN * I8_KS) + t * 32 * 16);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < I8_XD; ++i)
#pragma unroll
                    for (int j = 0; j < I8_WD; ++j)
                        if (i + j >= ORD_MIN)
                            acc[OMAX - i - j][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], bv[t][j], acc[OMAX - i - j][t], 0, 0, 0);
            }
        }
        if (it + 1 < nst) store_tiles(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    // value = (sum_o A_o 256^(o - ORD_MIN)) * 256^ORD_MIN * 2^(ex - 38) * 2^(ew - 30)
    double *out = slabs + (size_t)s * slab_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int cn = c0 + wc * 64 + t * 32 + (lane & 31);
        if (cn >= H) continue;
        const int we = i8_exponent(ewb[(size_t)s * Hc + cn]);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rl = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const int cm = m0 + rl;
            if (cm >= m) continue;
            const int xe = s_ex[rl];
            long long v = 0;      // sum_o A_o 256^(o - ORD_MIN): below 2^60, one rounding on the way to fp64
#pragma unroll
            for (int o = 0; o < NORD; ++o) v = v * 256 + (long long)acc[o][t][reg];
            double r;
            if (xe == INT_MIN || we == INT_MIN) r = __longlong_as_double(0x7ff8000000000000ll);      // a non-finite operand in the slice
            else r = ldexp((double)v, xe + we - 68 + 8 * ORD_MIN);
            out[(size_t)cm * H + cn] = r;
        }
    }
}

// ---- v3: the same product with the tiles brought in by LDS-DMA, two steps ahead ------------------------------------------
// 64 x 256 tile, 8 waves (2 x 4 of 32 x 64), ONE workgroup per CU.  Every step each wave issues 1 + 4 global_load_lds_dwordx4:
// its eighth of the raw fp32 X tile (8 rows x 128 B) of step + 3 and four 1 KiB pieces of W1's digit tile of step + 2, then cuts
// the raw X tile of step + 1 into digits (LDS -> LDS) and runs the 28 MFMAs of the step; ONE counted wait (vmcnt(5): only this
// step's five loads may still be in flight) and one barrier end the step.  No ordinary global load in the loop.
#define I8_RAW_BYTES (I8_BM * I8_KS * 4)           // 8 KB
#define I8_B3_BYTES (I8_WD * 256 * I8_KS)          // 32 KB
#define I8_A3_BYTES (I8_XD * I8_BM * I8_KS)        // 10 KB
#define I8_LDS3_BYTES (3 * I8_RAW_BYTES + 3 * I8_B3_BYTES + 2 * I8_A3_BYTES)
typedef __attribute__((address_space(3))) void *i8_lds_ptr_t;
template <int N>
__device__ __forceinline__ void i8_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ORD_MIN, int VAR = 0>
__global__ __launch_bounds__(512) void k_gemm_i8split3(const float *__restrict__ X, long ldx, int m, int F, int H, int Hc,
                                                       const int8_t *__restrict__ Wd, const unsigned *__restrict__ ewb, int steps,
                                                       int steps_per_slice, double *__restrict__ slabs, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char i8_lds[];
    unsigned char *raw = i8_lds;                                    // [3][64 rows][32 floats]
    unsigned char *Bs = i8_lds + 3 * I8_RAW_BYTES;                  // [3][4 digits][2 k halves][256 columns][16]
    unsigned char *As = Bs + 3 * I8_B3_BYTES;                       // [2][5 digits][2 k halves][64 rows][16]
    int *s_ex = reinterpret_cast<int *>(As + 2 * I8_A3_BYTES);      // [64]  (inside the dynamic block: ONE LDS object)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int m0 = blockIdx.x * I8_BM, s = blockIdx.z;
    const int st0 = s * steps_per_slice, st1 = min(st0 + steps_per_slice, steps);
    const int nst = st1 - st0;

    // the scale of (row, slice): thread (row = tid >> 3, 4 k per step)
    const int a_row = tid >> 3, a_k4 = (tid & 7) * 4;
    {
        const float *xrow = X + (size_t)min(m0 + a_row, m - 1) * ldx;
        unsigned mx = 0;
        for (int sb = st0; sb < st1; sb += 8) {       // (no branch per element: indices clamped into the row, the surplus masked)
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(sb + u, st1 - 1) * I8_KS + a_k4;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = xrow[min(k + q, F - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(sb + u, st1 - 1) * I8_KS + a_k4;
#pragma unroll
                for (int q = 0; q < 4; ++q) mx = max(mx, k + q < F ? __float_as_uint(v[u][q]) & 0x7fffffffu : 0u);
            }
        }
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 1, 64));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 2, 64));
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 4, 64));
        if ((tid & 7) == 0) s_ex[a_row] = i8_exponent(mx);
    }
    __syncthreads();
    const int a_e = s_ex[a_row];
    const double a_sc = a_e == INT_MIN ? 0.0 : __longlong_as_double((long long)(38 - a_e + 1023) << 52);

    // LDS-DMA sources of this lane: its 16 bytes of the wave's 8 rows of X (clamped inside the matrix: a partial last step reads
    // neighbours, which meet zero digits of W1), and of the wave's four pieces of W1's digit tile
    const int g_row = min(m0 + wave * 8 + (lane >> 3), m - 1);
    const float *g_x = X + (size_t)g_row * ldx + (lane & 7) * 4;
    const long x_last = (long)(m - 1) * ldx + F - 4;              // last float a 16-byte load may start at
    auto issue_a = [&](int st) {
        long off = (long)g_row * ldx + (long)st * I8_KS + (lane & 7) * 4;
        off = off > x_last ? x_last : off;
        __builtin_amdgcn_global_load_lds(X + off, (i8_lds_ptr_t)(raw + (st % 3) * I8_RAW_BYTES + wave * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int st) {
        // the tile is 32 pieces of 1 KiB, piece p = (digit * 2 + half) * 4 + column quarter; wave w takes p = w, w + 8, w + 16, w + 24
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave + 8 * q, dh = p >> 2, cq = p & 3;
            const int8_t *src = Wd + (((size_t)st * (I8_WD * 2) + dh) * Hc + cq * 64 + lane) * 16;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(src),
                                             (i8_lds_ptr_t)(Bs + (st % 3) * I8_B3_BYTES + dh * (256 * 16) + cq * 1024), 16, 0, 0);
        }
    };
    (void)g_x;
    float4 v_tail = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float *xr = X + (size_t)min(m0 + a_row, m - 1) * ldx;
        const int kt = (F / 4) * 4;               // the first k of the piece that straddles the row's end (none if F % 4 == 0)
        if (kt < F && (kt & (I8_KS - 1)) == a_k4) {
            v_tail.x = xr[kt];
            if (kt + 1 < F) v_tail.y = xr[kt + 1];
            if (kt + 2 < F) v_tail.z = xr[kt + 2];
        }
    }
    const unsigned lds_base = (unsigned)(size_t)i8_lds;          // LDS byte address of the dynamic block
    auto convert = [&](int st) {          // raw X tile of step st -> digit planes As[st & 1]
        // (every read of LDS that a DMA wrote is inline asm: hipcc otherwise drains vmcnt(0) in front of it -- it cannot tell which
        //  bytes the loads in flight will write -- and the prefetch is gone; the waits and barriers below order them)
        float4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(v)
                     : "v"(lds_base + (unsigned)((st % 3) * I8_RAW_BYTES + a_row * 128 + (tid & 7) * 16))
                     : "memory");
        const int k = st * I8_KS + a_k4;
        if (k + 3 >= F) v = v_tail;      // the piece that straddles the end of a row (its DMA source was clamped): read before the loop
        unsigned w[4], top[4], o[4];
        top[0] = i8_digits5(k < F ? v.x : 0.f, a_sc, w[0]);
        top[1] = i8_digits5(k + 1 < F ? v.y : 0.f, a_sc, w[1]);
        top[2] = i8_digits5(k + 2 < F ? v.z : 0.f, a_sc, w[2]);
        top[3] = i8_digits5(k + 3 < F ? v.w : 0.f, a_sc, w[3]);
        i8_transpose(w[0], w[1], w[2], w[3], o);
        unsigned *a = reinterpret_cast<unsigned *>(As + (st & 1) * I8_A3_BYTES) + (tid & 4 ? I8_BM * 4 : 0) + a_row * 4 + (tid & 3);
#pragma unroll
        for (int d = 0; d < 4; ++d) a[d * (I8_BM * I8_KS / 4)] = o[d];
        a[4 * (I8_BM * I8_KS / 4)] = top[0] | top[1] << 8 | top[2] << 16 | top[3] << 24;
    };

    constexpr int OMAX = I8_XD + I8_WD - 2, NORD = OMAX + 1 - ORD_MIN;
    i32x16 acc[NORD][2];
#pragma unroll
    for (int o = 0; o < NORD; ++o)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[o][t][i] = 0;

    // prologue: steps 0 and 1 (X: 0, 1, 2) on their way; step 0 landed and cut
    if (nst > 0) { issue_a(st0); issue_b(st0); }
    if (nst > 1) { issue_a(st0 + 1); issue_b(st0 + 1); }
    if (nst > 2) issue_a(st0 + 2);
    i8_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (nst > 0) convert(st0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_off = (lane >> 5) * (I8_BM * 16) + (wr * 32 + (lane & 31)) * 16;
    const int b_off = (lane >> 5) * (256 * 16) + (wc * 64 + (lane & 31)) * 16;
    for (int it = 0; it < nst; ++it) {
        const int st = st0 + it;
        const bool steady = it + 3 < nst;
        if (VAR != 3) { if (it + 3 < nst) issue_a(st + 3); }
        if (VAR != 2) { if (it + 2 < nst) issue_b(st + 2); }
        if (VAR != 4) { if (it + 1 < nst) convert(st + 1); }
        if (VAR != 1) {
            const unsigned a_addr = lds_base + (unsigned)(3 * I8_RAW_BYTES + 3 * I8_B3_BYTES + (st & 1) * I8_A3_BYTES + a_off);
            const unsigned b_addr = lds_base + (unsigned)(3 * I8_RAW_BYTES + (st % 3) * I8_B3_BYTES + b_off);
            i32x4 av[I8_XD], bv[2][I8_WD];
            asm volatile("ds_read_b128 %0, %13\n\t"
                         "ds_read_b128 %5, %14\n\t"
                         "ds_read_b128 %1, %13 offset:2048\n\t"
                         "ds_read_b128 %6, %14 offset:8192\n\t"
                         "ds_read_b128 %2, %13 offset:4096\n\t"
                         "ds_read_b128 %7, %14 offset:16384\n\t"
                         "ds_read_b128 %3, %13 offset:6144\n\t"
                         "ds_read_b128 %8, %14 offset:24576\n\t"
                         "ds_read_b128 %4, %13 offset:8192\n\t"
                         "ds_read_b128 %9, %14 offset:512\n\t"
                         "ds_read_b128 %10, %14 offset:8704\n\t"
                         "ds_read_b128 %11, %14 offset:16896\n\t"
                         "ds_read_b128 %12, %14 offset:25088\n\t"
                         "s_waitcnt lgkmcnt(4)"
                         : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]), "=&v"(av[4]), "=&v"(bv[0][0]), "=&v"(bv[0][1]),
                           "=&v"(bv[0][2]), "=&v"(bv[0][3]), "=&v"(bv[1][0]), "=&v"(bv[1][1]), "=&v"(bv[1][2]), "=&v"(bv[1][3])
                         : "v"(a_addr), "v"(b_addr)
                         : "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[1][0]), "+v"(bv[1][1]), "+v"(bv[1][2]), "+v"(bv[1][3])::"memory");
#pragma unroll
                for (int i = 0; i < I8_XD; ++i)
#pragma unroll
                    for (int j = 0; j < I8_WD; ++j)
                        if (i + j >= ORD_MIN)
                            acc[OMAX - i - j][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], bv[t][j], acc[OMAX - i - j][t], 0, 0, 0);
            }
        }
        // next step's tiles (issued one step ago) must have landed; this step's five loads may stay in flight
        if (steady && VAR == 0) i8_wait_vmcnt<5>();
        else i8_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    double *out = slabs + (size_t)s * slab_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int cn = wc * 64 + t * 32 + (lane & 31);
        if (cn >= H) continue;
        const int we = i8_exponent(ewb[(size_t)s * Hc + cn]);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rl = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const int cm = m0 + rl;
            if (cm >= m) continue;
            const int xe = s_ex[rl];
            long long v = 0;
#pragma unroll
            for (int o = 0; o < NORD; ++o) v = v * 256 + (long long)acc[o][t][reg];
            double r;
            if (xe == INT_MIN || we == INT_MIN) r = __longlong_as_double(0x7ff8000000000000ll);
            else r = ldexp((double)v, xe + we - 68 + 8 * ORD_MIN);
            out[(size_t)cm * H + cn] = r;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// K slices: a function of the FULL product's shapes only (a row has the same bits whichever rank computed it): as many as keep
// the workgroups within one resident round (256 CUs x 2).
int lt_i8_steps(int F) { return (F + I8_KS - 1) / I8_KS; }
int lt_i8_steps_per_slice(int n, int H, int F) {
    const int steps = lt_i8_steps(F);
#ifdef LT_I8_LAB
    static const int forced = [] { const char *e = getenv("LT_I8_SLICES"); return e ? atoi(e) : 0; }();
#else
    const int forced = 0;
#endif
    const long tiles = (long)((n + I8_BM - 1) / I8_BM) * ((lt_round_up(H, 64) + I8_BN - 1) / I8_BN);
    long want = forced > 0 ? forced : 512 / tiles;
    if (want < 1) want = 1;
    int per = (int)((steps + want - 1) / want);
    if (per < 8 && forced <= 0) per = 8;
    if (per > steps) per = steps;
    if (per < 1) per = 1;
    return per;
}
int lt_i8_slices(int n, int H, int F) {
    const int per = lt_i8_steps_per_slice(n, H, F);
    const int steps = lt_i8_steps(F);
    return steps > 0 ? (steps + per - 1) / per : 1;
}
bool lt_i8_shapes_ok(int n, int H, int F) { return n >= 256 && F >= 256 && H >= 64 && H % 64 == 0 && F < (1 << 15); }
size_t lt_i8_wd_bytes(int H, int F) { return (size_t)lt_i8_steps(F) * I8_WD * lt_round_up(H, 64) * I8_KS; }
size_t lt_i8_ew_bytes(int n, int H, int F) { return (size_t)lt_i8_slices(n, H, F) * lt_round_up(H, 64) * sizeof(unsigned); }

// W1's digits (once per refresh), then rows [r0, r0 + m) of the product as lt_i8_slices(n, H, F) fp64 slabs of m * H
int lt_launch_i8_w_digits(const float *W1, int n, int F, int H, int8_t *Wd, unsigned *ewb, hipStream_t st, bool clear = true) {
    const int Hc = lt_round_up(H, 64), steps = lt_i8_steps(F), per = lt_i8_steps_per_slice(n, H, F);
    if (clear) LT_HIP(hipMemsetAsync(ewb, 0, lt_i8_ew_bytes(n, H, F), st));      // (clear = false: the caller's last launch left the words zero)
    hipLaunchKernelGGL(k_i8_w_max, dim3((unsigned)steps, (unsigned)(Hc / 64)), dim3(256), 0, st, W1, F, H, Hc, per, ewb);
    LT_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_i8_w_digits, dim3((unsigned)steps, (unsigned)(Hc / 64)), dim3(256), 0, st, W1, F, H, Hc, per, ewb, Wd);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
template <int ORD_MIN, int VAR = 0>
int lt_launch_gemm_i8split(const float *Xrows, long ldx, int m, int n, int F, int H, const int8_t *Wd, const unsigned *ewb,
                           double *slabs, hipStream_t st) {
    if (m <= 0) return LT_OK;
    const int Hc = lt_round_up(H, 64), steps = lt_i8_steps(F), per = lt_i8_steps_per_slice(n, H, F), slices = lt_i8_slices(n, H, F);
    dim3 grid((unsigned)((m + I8_BM - 1) / I8_BM), (unsigned)((Hc + I8_BN - 1) / I8_BN), (unsigned)slices);
    const int av = (ldx % 4 == 0 && (uintptr_t)Xrows % 16 == 0) ? 4 : ((ldx % 2 == 0 && (uintptr_t)Xrows % 8 == 0) ? 2 : 1);
#define I8_GO(AV_)                                                                                                                       \
    do {                                                                                                                                 \
        LT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_i8split<AV_, ORD_MIN, VAR>),                                         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS_BYTES));                                           \
        hipLaunchKernelGGL((k_gemm_i8split<AV_, ORD_MIN, VAR>), grid, dim3(256), I8_LDS_BYTES, st, Xrows, ldx, m, F, H, Hc, Wd, ewb, steps,   \
                           per, slabs, (long)m * H);                                                                                     \
    } while (0)
    if (av == 4) I8_GO(4);
    else if (av == 2) I8_GO(2);
    else I8_GO(1);
#undef I8_GO
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// v3 launcher (H == 256 only in the lab); slices chosen so that the workgroups fit one resident round of one per CU
template <int ORD_MIN, int VAR = 0>
int lt_launch_gemm_i8split3(const float *Xrows, long ldx, int m, int n, int F, int H, const int8_t *Wd, const unsigned *ewb,
                            double *slabs, hipStream_t st) {
    const int Hc = lt_round_up(H, 64), steps = lt_i8_steps(F), per = lt_i8_steps_per_slice(n, H, F), slices = lt_i8_slices(n, H, F);
    dim3 grid((unsigned)((m + I8_BM - 1) / I8_BM), 1, (unsigned)slices);
    const size_t lds = I8_LDS3_BYTES + 64 * sizeof(int);
    LT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_i8split3<ORD_MIN, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_gemm_i8split3<ORD_MIN, VAR>), grid, dim3(512), lds, st, Xrows, ldx, m, F, H, Hc, Wd, ewb, steps, per, slabs, (long)m * H);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
