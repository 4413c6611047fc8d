// Does v_mfma_f32_16x16x4_f32 accumulate its four k-products as a k-ordered chain of fused multiply-adds, i.e. is
// D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))) bit for bit?  (v_mfma_f32_32x32x2_f32 -- the instruction of every
// GEMM kernel of the library -- checked the same way; a kernel built on the 16x16x4 form gives the same bits only if both do.)
//   hipcc -O3 --offload-arch=gfx950 tools/fold_test/mfma16_order_test.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float *A, const float *B, int K, float *C) {   // A [16][K], B [K][16], C [16][16]
    const int l = threadIdx.x;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * K + k + l / 16], B[(k + l / 16) * 16 + l % 16], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}
__global__ void k32(const float *A, const float *B, int K, float *C) {   // A [32][K], B [K][32], C [32][32]
    const int l = threadIdx.x;
    f32x16 d;
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    for (int k = 0; k < K; k += 2)
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * K + k + l / 32], B[(k + l / 32) * 32 + l % 32], d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l / 32)) * 32 + l % 32] = d[r];
}

int main() {
    const int K = 64;
    srand(3);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
    for (int T : {16, 32})
        for (int trial = 0; trial < 3; ++trial) {
            std::vector<float> A((size_t)T * K), B((size_t)K * T), C((size_t)T * T), R((size_t)T * T);
            const float sa = trial == 1 ? 1e-19f : 1.f, sb = trial == 1 ? 1e-20f : 1.f;
            for (auto &v : A) v = rnd() * sa;
            for (auto &v : B) v = rnd() * sb;
            if (trial == 2) for (int i = 0; i < T; ++i) for (int k = 0; k + 1 < K; k += 2) A[i * K + k + 1] = -A[i * K + k] * (1.f + 1e-6f * rnd());
            for (int i = 0; i < T; ++i)
                for (int j = 0; j < T; ++j) {
                    float acc = 0.f;
                    for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * T + j], acc);
                    R[i * T + j] = acc;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            if (T == 16) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
            else hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (size_t i = 0; i < C.size(); ++i)
                if (std::memcmp(&C[i], &R[i], 4) != 0 && bad++ < 3)
                    std::printf("   mismatch %zu: mfma %.9g fmaf-chain %.9g\n", i, C[i], R[i]);
            std::printf("%s trial %d: %d of %zu outputs differ from the k-ordered fmaf chain\n", T == 16 ? "16x16x4" : "32x32x2", trial, bad, C.size());
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    return 0;
}
