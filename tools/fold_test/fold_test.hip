#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__global__ void k(float* out) {
    int lane = threadIdx.x;
    // value v at lane l = 1000*v + l
    unsigned a = 1000 + lane, b = 2000 + lane, c = 3000 + lane, d = 4000 + lane;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r.x; out[64 + lane] = r.y;
    u32x2 q = __builtin_amdgcn_permlane16_swap(c, d, false, false);
    out[128 + lane] = q.x; out[192 + lane] = q.y;
    float z = (float)lane;
    out[256 + lane] = dpp_mov<0x128>(z);
    out[320 + lane] = dpp_mov<0x124>(z);
    out[384 + lane] = dpp_mov<0x4E>(z);
    out[448 + lane] = dpp_mov<0xB1>(z);
}
int main() {
    float *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"p32.x", "p32.y", "p16.x", "p16.y", "ror8", "ror4", "qp4E", "qpB1"};
    for (int i = 0; i < 8; ++i) { printf("%s:", names[i]); for (int l = 0; l < 64; ++l) printf(" %g", h[i*64+l]); printf("\n"); }
    return 0;
}
