// Peak rate of the f32 MFMA shapes on this GPU (no memory traffic): tells what a GEMM can hope for.
//   hipcc -O3 --offload-arch=gfx950 tools/fold_test/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

template <int SHAPE>
void run(int blocks_per_cu) {
    float *out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = SHAPE == 32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2;
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * flop_per_mfma;
    std::printf("mfma f32 %s, %d blocks/CU: %.3f ms  %.1f TFLOP/s\n", SHAPE == 32 ? "32x32x2" : "16x16x4", blocks_per_cu, ms,
                flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<32>(1); run<32>(2); run<16>(1); run<16>(2);
    return 0;
}
