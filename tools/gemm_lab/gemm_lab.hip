// When does each workgroup of the X*W1 product start and end?  Includes the library's GEMM translation unit with
// LT_GEMM_TRACE (start / end of every workgroup on the constant 100 MHz clock, XCC and hardware id) and runs the
// twitch-RU product (4385 x 3170 x 256, 7 K slices, 490 workgroups) a few times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ilinkteller_amd/csrc -c tools/gemm_lab/gemm_lab.hip -o /tmp/gemm_lab.o && \
//   hipcc --offload-arch=gfx950 /tmp/gemm_lab.o linkteller_amd/csrc/lt_core.o -o tools/gemm_lab/gemm_lab     (lt_core.o: lt_set_error, the profile hooks)
#define LT_GEMM_TRACE
#include "../../linkteller_amd/csrc/lt_gemm.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4385, K = argc > 2 ? atoi(argv[2]) : 3170, N = argc > 3 ? atoi(argv[3]) : 256;
    float *A, *B, *C, *slabs;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)K * N * 4);
    const int ks = argc > 4 ? atoi(argv[4]) : lt_gemm_pick_kslice(M, N, K);      // argv[4]: K slice (the product slices the probe rows as it slices X*W1)
    const int splits = (K + ks - 1) / ks;
    hipMalloc(&slabs, lt_gemm_splitk_slab_bytes(M, N, K, ks) + 16);
    const bool big = M >= 1024 && N % 128 == 0;     // fewer rows: the 64 x 64 kernel (k_gemm_f32_mfma_deep)
    const int nwg = big ? ((M + 127) / 128) * (N / 128) * splits : ((M + 63) / 64) * ((N + 63) / 64) * splits;
    unsigned long long *trace;
    hipMalloc(&trace, (size_t)nwg * 3 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_lt_gemm_trace), &trace, sizeof(trace));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> h((size_t)nwg * 3);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(trace, 0, (size_t)nwg * 3 * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        lt_launch_gemm_splitk(A, K, B, N, C, N, M, N, K, ks, slabs, 0, nullptr, 0.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[3 * i]); t1 = std::max(t1, h[3 * i + 1]); }
        std::vector<double> st(nwg), du(nwg);
        for (int i = 0; i < nwg; ++i) { st[i] = (h[3 * i] - t0) * 0.01; du[i] = (h[3 * i + 1] - h[3 * i]) * 0.01; }
        std::vector<double> s2 = st, d2 = du;
        std::sort(s2.begin(), s2.end()); std::sort(d2.begin(), d2.end());
        std::printf("rep %d: %d workgroups (kslice %d, %d slices); events (GEMM + slab sum) %.1f us; first start -> last end %.1f us\n", rep, nwg, ks,
                    splits, ms * 1e3, (t1 - t0) * 0.01);
        std::printf("   start offsets us: min %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p90 %.1f  max %.1f\n", s2[0], s2[nwg / 4], s2[nwg / 2],
                    s2[3 * nwg / 4], s2[9 * nwg / 10], s2[nwg - 1]);
        std::printf("   durations us:     min %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p90 %.1f  max %.1f\n", d2[0], d2[nwg / 4], d2[nwg / 2],
                    d2[3 * nwg / 4], d2[9 * nwg / 10], d2[nwg - 1]);
        if (rep == 3) {
            // workgroups per (XCC, CU): HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 (gfx9)
            std::vector<int> per(8 * 512, 0);
            for (int i = 0; i < nwg; ++i) {
                const unsigned hw = (unsigned)h[3 * i + 2], xcc = (unsigned)(h[3 * i + 2] >> 32);
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                per[xcc * 512 + se * 32 + sh * 16 + cu]++;
            }
            int hist[8] = {};
            int used = 0;
            for (int v : per) if (v) { hist[std::min(v, 7)]++; ++used; }
            std::printf("   CUs used %d; CUs with 1 / 2 / 3 / 4+ workgroups: %d / %d / %d / %d\n", used, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
            std::printf("   late starters (start > 5 us): ");
            int late = 0;
            for (int i = 0; i < nwg; ++i) if (st[i] > 5.0) ++late;
            std::printf("%d\n", late);
        }
    }
    return 0;
}
