// Lab: the fp64-grade product X * W1 as an error-free split on the int8 matrix cores (i8_split.hip), measured against the f64
// matrix-core route's job.  Builds the twitch-RU shapes (4385 x 3170 times 3170 x 256), two kinds of features,
// checks 256 rows against a host fp64 product and times the three launches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLT_I8_LAB -I tools/i8_lab tools/i8_lab/i8_lab.hip -o /tmp/i8_lab && /tmp/i8_lab
#define LT_I8_LAB
#include "../../linkteller_amd/csrc/lt_i8_split.hip.h"
#include <cmath>
#include <random>
#include <vector>

__global__ void k_sum(const double *slabs, long stride, int splits, long total, double *C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double a = slabs[i];
    for (int z = 1; z < splits; ++z) a += slabs[(long)z * stride + i];
    C[i] = a;
}

template <int ORD_MIN, int VAR = 0, int V3 = 0>
static void run(const char *kind, const std::vector<float> &X, const std::vector<float> &W, int n, int F, int H) {
    float *dX, *dW; int8_t *Wd; unsigned *ew; double *slabs, *C;
    const int slices = lt_i8_slices(n, H, F);
    LT_HIP(hipMalloc(&dX, X.size() * 4)); LT_HIP(hipMalloc(&dW, W.size() * 4));
    LT_HIP(hipMalloc(&Wd, lt_i8_wd_bytes(H, F))); LT_HIP(hipMalloc(&ew, lt_i8_ew_bytes(n, H, F)));
    LT_HIP(hipMalloc(&slabs, (size_t)slices * n * H * 8)); LT_HIP(hipMalloc(&C, (size_t)n * H * 8));
    LT_HIP(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    LT_HIP(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e[4];
    for (auto &x : e) LT_HIP(hipEventCreate(&x));
    float t[3] = {0, 0, 0};
    const int reps = 20;
    for (int r = -3; r < reps; ++r) {
        LT_HIP(hipEventRecord(e[0], 0));
        lt_launch_i8_w_digits(dW, n, F, H, Wd, ew, 0);
        LT_HIP(hipEventRecord(e[1], 0));
        if (V3) lt_launch_gemm_i8split3<ORD_MIN, VAR>(dX, F, n, n, F, H, Wd, ew, slabs, 0);
        else lt_launch_gemm_i8split<ORD_MIN, VAR>(dX, F, n, n, F, H, Wd, ew, slabs, 0);
        LT_HIP(hipEventRecord(e[2], 0));
        hipLaunchKernelGGL(k_sum, dim3((unsigned)(((long)n * H + 255) / 256)), dim3(256), 0, 0, slabs, (long)n * H, slices, (long)n * H, C);
        LT_HIP(hipEventRecord(e[3], 0));
        LT_HIP(hipDeviceSynchronize());
        if (r >= 0)
            for (int i = 0; i < 3; ++i) { float ms; LT_HIP(hipEventElapsedTime(&ms, e[i], e[i + 1])); t[i] += ms; }
    }
    std::vector<double> got((size_t)n * H);
    LT_HIP(hipMemcpy(got.data(), C, got.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, mean = 0; long cnt = 0;
    for (int r = 0; r < n; r += (n + 255) / 256) {
        std::vector<double> ref(H, 0.0);
        for (int k = 0; k < F; ++k) {
            const double x = X[(size_t)r * F + k];
            if (x != 0.0) for (int c = 0; c < H; ++c) ref[c] += x * (double)W[(size_t)k * H + c];
        }
        double mx = 0;
        for (int c = 0; c < H; ++c) mx = std::fmax(mx, std::fabs(ref[c]));
        for (int c = 0; c < H; ++c) { const double d = std::fabs(got[(size_t)r * H + c] - ref[c]) / mx; worst = std::fmax(worst, d); mean += d; ++cnt; }
    }
    if (V3) printf("v3 (LDS-DMA, two steps ahead): ");
    if (VAR) printf("VARIANT %d (1 no MFMA, 2 no W1 tile loads, 3 no X tile loads, 4 v2: one LDS buffer / v3: no digit cutting): ", VAR);
    printf("%-7s orders >= %d (%2d digit pairs): |rows - host f64| / row max: worst %.2e mean %.2e | W1 digits %.1f us, product %.1f us, "
           "slab sum (%d slices) %.1f us\n", kind, ORD_MIN, ORD_MIN == 3 ? 14 : (ORD_MIN == 4 ? 10 : 17), worst, mean / cnt, t[0] / reps * 1e3,
           t[1] / reps * 1e3, slices, t[2] / reps * 1e3);
    fflush(stdout);
    hipFree(dX); hipFree(dW); hipFree(Wd); hipFree(ew); hipFree(slabs); hipFree(C);
}

int main() {
    const int n = 4385, F = 3170, H = 256;
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float> W((size_t)F * H), Xg((size_t)n * F), Xt((size_t)n * F);
    const float lim = std::sqrt(6.f / (F + H));
    for (auto &w : W) w = (2 * ud(g) - 1) * lim;
    for (auto &x : Xg) x = nd(g);
    // standardised indicator columns (what one-hot features become after the reference's row / column normalisation): a column
    // holds two values, (0 - p) / s and (1 - p) / s -- rare columns make the outliers that set a row's scale
    for (int k = 0; k < F; ++k) {
        const float p = std::pow(10.f, -1.f - 2.5f * ud(g)), s = std::sqrt(p * (1 - p));
        for (int r = 0; r < n; ++r) Xt[(size_t)r * F + k] = ((ud(g) < p ? 1.f : 0.f) - p) / s;
    }
    run<4>("gauss", Xg, W, n, F, H);
    run<3>("gauss", Xg, W, n, F, H);
    if (getenv("I8_V3")) {
        run<3, 0, 1>("gauss", Xg, W, n, F, H);
        run<3, 1, 1>("gauss", Xg, W, n, F, H);
        run<3, 2, 1>("gauss", Xg, W, n, F, H);
        run<3, 3, 1>("gauss", Xg, W, n, F, H);
        run<3, 4, 1>("gauss", Xg, W, n, F, H);
        return 0;
    }
    if (getenv("I8_VARIANTS")) {
        run<3, 1>("gauss", Xg, W, n, F, H);
        run<3, 2>("gauss", Xg, W, n, F, H);
        run<3, 3>("gauss", Xg, W, n, F, H);
        run<3, 4>("gauss", Xg, W, n, F, H);
        return 0;
    }
    run<4>("twitch", Xt, W, n, F, H);
    run<3>("twitch", Xt, W, n, F, H);
    return 0;
}
