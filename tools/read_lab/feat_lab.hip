// Where does k_s1d_feature_rows spend its 21 us?  Includes the library's fp64 translation unit with LT_FD_TRACE (seven
// stamps per row wave on the constant 100 MHz clock: entry, reference vector staged, first / middle / last compare step
// done, list walked, row stored) and runs the twitch-RU shape (4385 x 3170 -> 256) in the production configuration
// (deferred cref: 50 slab blocks in front, fixed-point rows) a few times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DFD_MIN_WAVES=5 -Iinclude -Ilinkteller_amd/csrc -c tools/read_lab/feat_lab.hip -o /tmp/feat_lab.o && \
//   (cd linkteller_amd/csrc && hipcc --offload-arch=gfx950 /tmp/feat_lab.o lt_core.o lt_gemm.o lt_spmm.o lt_forward.o lt_influence.o lt_gcn3.o lt_dp.o -o ../../tools/read_lab/feat_lab)
#define LT_FD_TRACE
#include "../../linkteller_amd/csrc/lt_fp64.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

static int g_flags = 0, g_stagger = 0;
template <int VEC>
static void run(int n, int F, int H, const float *X, const float *ref, const float *W1, double *S1d, int *hint, double *slabs, int *zstate,
                float *S1x, unsigned *gate, double *cref, double *S1qs, unsigned long long *trace, const char *name) {
    const int nslab = (F + 63) / 64;
    const unsigned blocks = (unsigned)((n + FD_WAVES - 1) / FD_WAVES + nslab);
    const size_t smem = fd_smem_bytes(F);
    const size_t nw = (size_t)blocks * FD_WAVES;
    std::vector<unsigned long long> h(nw * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(trace, 0, nw * 8 * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_s1d_feature_rows<VEC, true>), dim3(blocks), dim3(64 * FD_WAVES), smem, 0, n, F, H, H, X, (long)F, ref, W1,
                           (const double *)nullptr, S1d, fd_hint_cap(F), hint, nslab, slabs, zstate, S1x, gate, cref, S1qs, lt_bits_job{}, 0, g_flags, g_stagger);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep < 3) continue;
        hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<std::vector<double>> ph(7);
        for (size_t w = (size_t)nslab * FD_WAVES; w < nw; ++w) {
            if (!h[w * 8] || !h[w * 8 + 6]) continue;
            t0 = std::min(t0, h[w * 8]);
            t1 = std::max(t1, h[w * 8 + 6]);
        }
        for (size_t w = (size_t)nslab * FD_WAVES; w < nw; ++w) {
            if (!h[w * 8] || !h[w * 8 + 6]) continue;
            for (int k = 0; k < 7; ++k) ph[k].push_back((h[w * 8 + k] - t0) * 0.01);
        }
        std::printf("%s: event time %.1f us; first row-wave entry -> last row stored %.1f us; %zu row waves\n", name, ms * 1e3, (t1 - t0) * 0.01, ph[0].size());
        if ((g_stagger & 255) > 1) {       // per start group: when its waves enter, and how long a wave takes from entry to its store
            const int G = g_stagger & 255;
            const size_t nrb = blocks - nslab;
            for (int gq = 0; gq < G; ++gq) {
                std::vector<double> en, du;
                for (size_t w = (size_t)nslab * FD_WAVES; w < nw; ++w) {
                    if (!h[w * 8] || !h[w * 8 + 6]) continue;
                    const size_t bk = w / FD_WAVES - nslab;
                    if ((int)(bk * G / nrb) != gq) continue;
                    en.push_back((h[w * 8] - t0) * 0.01);
                    du.push_back((h[w * 8 + 6] - h[w * 8]) * 0.01);
                }
                std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
                if (!en.empty())
                    std::printf("   group %d: entry p50 %.1f us; entry -> stored per wave p10 %.1f p50 %.1f p90 %.1f max %.1f us\n", gq, en[en.size() / 2],
                                du[du.size() / 10], du[du.size() / 2], du[9 * du.size() / 10], du.back());
            }
        }
        const char *nm[7] = {"entry", "ref staged (barrier)", "compare step 0 done", "compare step UN/2 done", "last compare step done",
                             "list walked", "row stored"};
        for (int k = 0; k < 7; ++k) {
            std::vector<double> v = ph[k];
            std::sort(v.begin(), v.end());
            const size_t m = v.size();
            std::printf("   %-26s us since the first entry: min %5.1f  p10 %5.1f  p50 %5.1f  p90 %5.1f  max %5.1f\n", nm[k], v[0], v[m / 10], v[m / 2],
                        v[9 * m / 10], v[m - 1]);
        }
        {   // the waves that are stored last (the slowest 5 %): where did THEY spend their time
            std::vector<std::pair<double, size_t>> order;
            for (size_t i = 0; i < ph[0].size(); ++i) order.push_back({ph[6][i], i});
            std::sort(order.begin(), order.end());
            const size_t m = order.size(), k0 = m - m / 20;
            double acc[7] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t q = k0; q < m; ++q)
                for (int k = 0; k < 7; ++k) acc[k] += ph[k][order[q].second];
            std::printf("   the last 5 %% of the waves (mean, us since the first entry):");
            for (int k = 0; k < 7; ++k) std::printf("  %s %.1f", nm[k], acc[k] / (m - k0));
            std::printf("\n");
            double acc2[7] = {0, 0, 0, 0, 0, 0, 0};
            const size_t k1 = m / 20;
            for (size_t q = 0; q < k1; ++q)
                for (int k = 0; k < 7; ++k) acc2[k] += ph[k][order[q].second];
            std::printf("   the first 5 %% of the waves:");
            for (int k = 0; k < 7; ++k) std::printf("  %s %.1f", nm[k], acc2[k] / k1);
            std::printf("\n");
        }
        // per-wave phase lengths
        const char *pn[6] = {"staging", "-> first data compared", "-> half the row compared", "-> whole row compared", "walk", "convert + store"};
        for (int k = 0; k < 6; ++k) {
            std::vector<double> v;
            for (size_t i = 0; i < ph[0].size(); ++i) v.push_back(ph[k + 1][i] - ph[k][i]);
            std::sort(v.begin(), v.end());
            const size_t m = v.size();
            std::printf("   phase %-26s per wave: p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f us\n", pn[k], v[m / 10], v[m / 2], v[9 * m / 10], v[m - 1]);
        }
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4385, F = argc > 2 ? atoi(argv[2]) : 3170, H = argc > 3 ? atoi(argv[3]) : 256;
    g_flags = argc > 4 ? atoi(argv[4]) : 0;
    g_stagger = argc > 5 ? atoi(argv[5]) : 0;      // (gap in 10-ns ticks) << 8 | groups
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> hx((size_t)n * F), href((size_t)F + FD_REF_PAD, 0.f), hw((size_t)F * H);
    for (int j = 0; j < F; ++j) href[j] = -0.0776f - 1e-4f * (j % 7);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < F; ++j) hx[(size_t)i * F + j] = U(rng) < 0.006f ? 12.88f + 0.01f * (j % 5) : href[j];
    for (auto &v : hw) v = U(rng) - 0.5f;
    float *X, *ref, *W1, *S1x;
    double *S1d, *slabs, *cref, *S1qs;
    int *hint, *zstate;
    unsigned *gate;
    unsigned long long *trace;
    hipMalloc(&X, hx.size() * 4); hipMalloc(&ref, href.size() * 4); hipMalloc(&W1, hw.size() * 4);
    hipMalloc(&S1x, (size_t)n * H * 4); hipMalloc(&S1d, (size_t)n * H * 8); hipMalloc(&slabs, (size_t)((F + 63) / 64) * H * 8);
    hipMalloc(&cref, (size_t)H * 8); hipMalloc(&S1qs, (size_t)n * 8); hipMalloc(&hint, 4); hipMalloc(&zstate, (size_t)n * 4);
    hipMalloc(&gate, 4); hipMemset(gate, 0, 4); hipMemset(hint, 0, 4);
    const size_t nw = ((size_t)(n + FD_WAVES - 1) / FD_WAVES + (F + 63) / 64) * FD_WAVES;
    hipMalloc(&trace, nw * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_fd_trace), &trace, sizeof(trace));
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ref, href.data(), href.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<2>(n, F, H, X, ref, W1, S1d, hint, slabs, zstate, S1x, gate, cref, S1qs, trace, "VEC = 2 (8-byte loads, 52 compare steps)");
    run<1>(n, F, H, X, ref, W1, S1d, hint, slabs, zstate, S1x, gate, cref, S1qs, trace, "VEC = 1 (4-byte loads, two trips)");
    return 0;
}
