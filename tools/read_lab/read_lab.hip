// How fast can one wave per row read a [n, F] fp32 matrix whose rows are 12.7 KB (twitch: n = 4385, F = 3170)?  Variants of the
// load pattern of k_s1d_feature_rows (lt_fp64.hip), no processing beyond a sum.  Between timed launches a 64 MB scratch is
// written, as a step of the real pipeline would do.
//   hipcc -O3 --offload-arch=gfx950 tools/read_lab/read_lab.hip -o tools/read_lab/read_lab && tools/read_lab/read_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// V = floats per lane and load (1, 2, 4: 4 uses 16-byte loads from the row start rounded down), W = waves per block,
// ROWS_PER_WAVE rows handled one after the other by a wave
template <int V, int W, int UN>
__global__ __launch_bounds__(64 * W) void k_read(int n, int F, const float *__restrict__ X, long ldx, float *__restrict__ out,
                                                 int rows_per_wave) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int i = (blockIdx.x * W + wid) * rows_per_wave + rr;
        if (i >= n) break;
        const float *xr = X + (long)i * ldx;
        if (V == 4) {
            const unsigned long a = (unsigned long)xr;
            const float *x0 = (const float *)(a & ~15ul);
            const int shift = (int)((a & 15ul) >> 2);
            for (int j0 = 0; j0 < F + shift; j0 += 256 * UN) {
                f32x4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int j = j0 + u * 256 + lane * 4;
                    v[u] = j + 3 < F + shift ? *reinterpret_cast<const f32x4 *>(x0 + j) : f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
            }
        } else if (V == 2) {
            for (int j0 = 0; j0 < F; j0 += 128 * UN) {
                f32x2 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int j = j0 + u * 128 + lane * 2;
                    v[u] = j + 1 < F ? *reinterpret_cast<const f32x2 *>(xr + j) : f32x2{0, 0};
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) acc += v[u][0] + v[u][1];
            }
        } else {
            for (int j0 = 0; j0 < F; j0 += 64 * UN) {
                float v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) { const int j = j0 + u * 64 + lane; v[u] = j < F ? xr[j] : 0.f; }
#pragma unroll
                for (int u = 0; u < UN; ++u) acc += v[u];
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_fill(float *p, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = 1.f; }
// plain streaming read of the whole buffer, 16 bytes per lane, grid-stride
__global__ __launch_bounds__(256) void k_stream(const f32x4 *__restrict__ p, long n4, float *out) {
    float acc = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) { const f32x4 v = p[i]; acc += v[0] + v[1] + v[2] + v[3]; }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F> float timeit(F &&launch, float *scratch, long ns, bool dirty) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f; const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        if (dirty) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, scratch, ns);
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    return sum / reps * 1e3f;
}
int main() {
    const int n = 4385, F = 3170; const long ldx = F;
    float *X, *out, *scratch; const long ns = 16L << 20;
    hipMalloc(&X, (size_t)n * ldx * 4 + 64); hipMalloc(&out, 64); hipMalloc(&scratch, ns * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, X, (long)n * ldx);
    const double mb = (double)n * F * 4 / 1e6;
    for (int dirty = 0; dirty < 2; ++dirty) {
        printf("--- %s\n", dirty ? "64 MB written between launches" : "back to back (X stays in the Infinity Cache)");
        auto rep = [&](const char *name, float us) { printf("%-44s %7.1f us  %6.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6); };
        rep("stream float4, 2048 blocks", timeit([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const f32x4 *)X, (long)n * ldx / 4, out); }, scratch, ns, dirty));
        rep("row/wave float2 x26, 4 waves/block", timeit([&] { hipLaunchKernelGGL((k_read<2, 4, 26>), dim3((n + 3) / 4), dim3(256), 0, 0, n, F, X, ldx, out, 1); }, scratch, ns, dirty));
        rep("row/wave float2 x13 (2 trips)", timeit([&] { hipLaunchKernelGGL((k_read<2, 4, 13>), dim3((n + 3) / 4), dim3(256), 0, 0, n, F, X, ldx, out, 1); }, scratch, ns, dirty));
        rep("row/wave float4 x13, 4 waves/block", timeit([&] { hipLaunchKernelGGL((k_read<4, 4, 13>), dim3((n + 3) / 4), dim3(256), 0, 0, n, F, X, ldx, out, 1); }, scratch, ns, dirty));
        rep("row/wave float1 x50", timeit([&] { hipLaunchKernelGGL((k_read<1, 4, 50>), dim3((n + 3) / 4), dim3(256), 0, 0, n, F, X, ldx, out, 1); }, scratch, ns, dirty));
        rep("row/wave float4 x13, 1 wave/block", timeit([&] { hipLaunchKernelGGL((k_read<4, 1, 13>), dim3(n), dim3(64), 0, 0, n, F, X, ldx, out, 1); }, scratch, ns, dirty));
        rep("2 rows/wave float4 x13", timeit([&] { hipLaunchKernelGGL((k_read<4, 4, 13>), dim3((n + 7) / 8), dim3(256), 0, 0, n, F, X, ldx, out, 2); }, scratch, ns, dirty));
        rep("4 rows/wave float4 x13", timeit([&] { hipLaunchKernelGGL((k_read<4, 4, 13>), dim3((n + 15) / 16), dim3(256), 0, 0, n, F, X, ldx, out, 4); }, scratch, ns, dirty));
    }
    return 0;
}
