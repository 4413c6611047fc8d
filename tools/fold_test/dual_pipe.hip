// Do the VALU (v_pk_fma_f32) and the matrix pipe (v_mfma_f32_16x16x4_f32) of a SIMD run side by side for ONE wave's
// instruction stream?  Times three kernels with the same loop trip count: 8 packed FMAs per trip, 1 MFMA per trip
// (both ~32 cycles of their pipe), and both interleaved.  If the pipes overlap, "both" costs about max(), not sum.
//   hipcc -O3 --offload-arch=gfx950 tools/fold_test/dual_pipe.hip -o /tmp/dual_pipe && /tmp/dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 1: VALU only, 2: MFMA only, 3: both
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b + i};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const f32x2 m = {a, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m, m);
        }
        if (MODE & 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            if (MODE == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
    }
    float s = c0[0] + c1[1];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(int waves_per_simd) {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 200000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        const float a = run<1>(w), b = run<2>(w), c = run<3>(w);
        std::printf("%d wave(s)/SIMD: 8 x v_pk_fma %.2f ms | mfma (x2 in its own loop) %.2f ms | 8 x v_pk_fma + 1 mfma %.2f ms\n", w, a, b, c);
    }
    return 0;
}
