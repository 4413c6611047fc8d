// Can FULL stage A run its P = 32 chains on the matrix pipe?  One v_mfma_f32_32x32x1_2b_f32 per (entry, component):
// A = the entry's coefficient in every lane (32 probes x 2 blocks: all rows equal -- the 32 recomputations of the
// row), B = lane l's component t of the gathered S1 row (columns 4 l + t: the layout the VALU kernel reads with one
// ds_read_b128), D_t = 32 accumulators.  Checks, bit for bit against the fmaf chain z = fma(a_k, s_k, z) from the bias:
//   * the accumulation of the instruction is ONE fused multiply-add per k (denormals included),
//   * after 16 v_permlane32_swap per component the registers are the VALU kernel's (lane = 4 consecutive columns,
//     every probe in a register), probe order 8 g + 4 half + r.
// Then times the candidate inner loops (MFMA chains alone, the epilogue-like VALU mix alone, both in one stream and
// in separate waves).
//   hipcc -O3 --offload-arch=gfx950 tools/fold_test/mfma_chain_test.hip -o /tmp/mfma_chain_test && /tmp/mfma_chain_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void swap32(float &a, float &b) {   // lanes 32-63 of a <-> lanes 0-31 of b
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// z[p][col] for p < 32, col < 256 through the MFMA route
__global__ __launch_bounds__(64) void k_chain(const float *a, const float *S, const float *bias, int K, float *z) {
    const int lane = threadIdx.x;
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 4 * lane);
    f32x32 D[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float own = b[t];
        const float oth = __shfl_xor(own, 32, 64);
        const float lo = lane < 32 ? own : oth, hi = lane < 32 ? oth : own;   // block 0: columns 4 n + t, block 1: 4 (32 + n) + t
#pragma unroll
        for (int v = 0; v < 16; ++v) { D[t][v] = lo; D[t][16 + v] = hi; }
    }
    for (int k = 0; k < K; ++k) {
        const float ak = a[k];
        const f32x4 s = *reinterpret_cast<const f32x4 *>(S + (size_t)k * 256 + 4 * lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) D[t] = __builtin_amdgcn_mfma_f32_32x32x1f32(ak, s[t], D[t], 0, 0, 0);
    }
    // to the VALU layout: register v of block 0 keeps probes of half 0, register v of block 1 gets half 1
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float x = D[t][v], y = D[t][16 + v];
            swap32(x, y);
            D[t][v] = x; D[t][16 + v] = y;
        }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int p0 = 8 * (v / 4) + (v % 4), p1 = p0 + 4;
            z[(size_t)p0 * 256 + 4 * lane + t] = D[t][v];
            z[(size_t)p1 * 256 + 4 * lane + t] = D[t][16 + v];
        }
}

// ---- timing -------------------------------------------------------------------------------------------------------
// MODE 1: 4 MFMA per trip (one entry); MODE 2: an epilogue-like VALU mix per trip, scaled to the epilogue's share
// (per 18-entry row: 128 v_max, 128 v_pk_fma/mul, 128 v_mov, ~190 permlane/dpp adds -> per entry ~ 7 + 7 + 7 + 10);
// MODE 3: both in one instruction stream; MODE 4: even waves MODE 1, odd waves MODE 2 (co-resident on a SIMD)
template <int MODE>
__global__ __launch_bounds__(64) void k_time(float *out, int iters, float a, float b) {
    f32x32 D[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 32; ++v) D[t][v] = a * v + t;
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = f32x2{b + i, a - i}; }
    const bool mf = MODE == 1 || MODE == 3 || (MODE == 4 && (blockIdx.x & 1) == 0);
    const bool va = MODE == 2 || MODE == 3 || (MODE == 4 && (blockIdx.x & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (mf) {
#pragma unroll
            for (int t = 0; t < 4; ++t) D[t] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, D[t], 0, 0, 0);
        }
        if (va) {
#pragma unroll
            for (int i = 0; i < 7; ++i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x[i]));
#pragma unroll
            for (int i = 0; i < 7; ++i) y[i] = __builtin_elementwise_fma(y[i], f32x2{a, a}, y[(i + 1) & 7]);
#pragma unroll
            for (int i = 0; i < 7; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i].x));
#pragma unroll
            for (int i = 0; i < 5; ++i) { swap32(x[i], x[i + 1]); x[i] += x[i + 1]; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += D[t][0] + D[t][17];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
float run(int waves_per_simd) {
    float *out;
    const int blocks = 256 * 4 * waves_per_simd;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(64), 0, 0, out, 20000, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

// which VALU instructions run beside an fp32 MFMA stream of the SAME wave?  KIND 0: none, 1: v_max_f32, 2: v_mov_b32,
// 3: v_permlane32_swap, 4: v_pk_fma_f32, 5: v_add_f32, 6: v_and_b32 (integer), 7: ds_read_b128
template <int KIND, bool MF>
__global__ __launch_bounds__(64) void k_kind(float *out, int iters, float a, float b) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;
    f32x32 D[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 32; ++v) D[t][v] = a * v + t;
    float x[16];
    f32x2 y[16];
    f32x4 q[4] = {};
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = a + i; y[i] = f32x2{b + i, a - i}; }
    for (int it = 0; it < iters; ++it) {
        if (MF) {
#pragma unroll
            for (int t = 0; t < 4; ++t) D[t] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, D[t], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 1) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x[i]));
            if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
            if (KIND == 3 && i < 8) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]));
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(y[(i + 1) & 15]));
            if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
            if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
            if (KIND == 7 && i < 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"((unsigned)(threadIdx.x * 16)));
        }
        if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = q[0].x + q[1].y + q[2].z + q[3].w;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += D[t][0] + D[t][17];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND, bool MF>
float run_kind(int waves_per_simd) {
    float *out;
    const int blocks = 256 * 4 * waves_per_simd;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_kind<KIND, MF>), dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_kind<KIND, MF>), dim3(blocks), dim3(64), 0, 0, out, 20000, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
template <int KIND>
void report(const char *name) {
    for (int w = 1; w <= 2; ++w)
        std::printf("  %-22s %d wave(s)/SIMD: alone %.2f ms | with 4 mfma per trip %.2f ms (mfma alone %.2f)\n", name, w,
                    run_kind<KIND, false>(w), run_kind<KIND, true>(w), run_kind<0, true>(w));
}

int main() {
    const int K = 37;
    std::vector<float> a(K), S((size_t)K * 256), bias(256), z(32 * 256), ref(256);
    srand(5);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int trial = 0; trial < 3; ++trial) {
        // trial 0: ordinary magnitudes; 1: products and sums in the denormal range; 2: heavy cancellation
        const float sc = trial == 1 ? 1e-20f : 1.f;
        for (auto &v : a) v = rnd() * (trial == 1 ? 1e-19f : 1.f);
        for (auto &v : S) v = rnd() * sc;
        for (auto &v : bias) v = trial == 1 ? rnd() * 1e-39f : (trial == 2 ? 1e6f * rnd() : rnd());
        if (trial == 2) for (int k = 0; k + 1 < K; k += 2) { a[k + 1] = -a[k]; for (int c = 0; c < 256; ++c) S[(size_t)(k + 1) * 256 + c] = S[(size_t)k * 256 + c] * (1.f + 1e-6f * rnd()); }
        for (int c = 0; c < 256; ++c) {
            float acc = bias[c];
            for (int k = 0; k < K; ++k) acc = fmaf(a[k], S[(size_t)k * 256 + c], acc);
            ref[c] = acc;
        }
        float *da, *dS, *db, *dz;
        hipMalloc(&da, K * 4); hipMalloc(&dS, S.size() * 4); hipMalloc(&db, 256 * 4); hipMalloc(&dz, z.size() * 4);
        hipMemcpy(da, a.data(), K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, bias.data(), 256 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, da, dS, db, K, dz);
        hipMemcpy(z.data(), dz, z.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, denorm = 0;
        for (int p = 0; p < 32; ++p)
            for (int c = 0; c < 256; ++c) {
                if (std::memcmp(&z[(size_t)p * 256 + c], &ref[c], 4) != 0 && bad++ < 5)
                    std::printf("  mismatch p=%d col=%d: mfma %.9g (%08x) fmaf %.9g (%08x)\n", p, c, z[(size_t)p * 256 + c],
                                *(unsigned *)&z[(size_t)p * 256 + c], ref[c], *(unsigned *)&ref[c]);
                if (p == 0 && ref[c] != 0.f && std::fabs(ref[c]) < 1.1754944e-38f) ++denorm;
            }
        std::printf("trial %d: %d of %d values differ from the fmaf chain (denormal results among the 256 columns: %d)\n", trial, bad, 32 * 256, denorm);
        hipFree(da); hipFree(dS); hipFree(db); hipFree(dz);
    }
    for (int w = 1; w <= 4; ++w) {
        const float m1 = run<1>(w), m2 = run<2>(w), m3 = run<3>(w), m4 = run<4>(w);
        std::printf("%d wave(s)/SIMD: mfma only %.2f ms | valu mix only %.2f ms | both, one stream %.2f ms | both, alternate waves %.2f ms (each kind: half the waves)\n", w, m1, m2, m3, m4);
    }
    std::printf("16 VALU instructions of one kind per trip beside 4 x v_mfma_f32_32x32x1_2b_f32 (256 cycles):\n");
    report<1>("v_max_f32 x16");
    report<2>("v_mov_b32 x16");
    report<3>("v_permlane32_swap x8");
    report<4>("v_pk_fma_f32 x16");
    report<5>("v_add_f32 x16");
    report<6>("v_and_b32 x16");
    report<7>("ds_read_b128 x4");
    return 0;
}
