// k_s1d_feature_ring (the persistent LDS-ring form of the product rows, lt_feature_ring.hip.h) against k_s1d_feature_rows
// (one wave per row) on the same device arrays: results compared (fixed-point rows + scales + cref, and the fp64 rows of the
// non-deferred form, against each other and against a host fp64 sum), launch times by HIP events (mean of 50 back-to-back
// launches and the median of 20 single launches).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ilinkteller_amd/csrc -c tools/read_lab/ring_lab.hip -o /tmp/ring_lab.o && \
//   (cd linkteller_amd/csrc && hipcc --offload-arch=gfx950 /tmp/ring_lab.o lt_core.o lt_gemm.o lt_spmm.o lt_forward.o lt_influence.o lt_gcn3.o lt_dp.o -o ../../tools/read_lab/ring_lab)
//   tools/read_lab/ring_lab [n F H density special]
#define LT_FR_TRACE
#include "../../linkteller_amd/csrc/lt_fp64.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

struct Bufs {
    float *X, *ref, *W1, *S1x;
    double *S1d, *slabs, *cref, *S1qs;
    int *hint, *zstate;
    unsigned *gate;
};

static int g_flags = 0, g_stagger = 0;
static void launch_old(int n, int F, int H, const Bufs &b, bool defer) {
    const int nslab = defer ? (F + 63) / 64 : 0;
    const unsigned blocks = (unsigned)((n + FD_WAVES - 1) / FD_WAVES + nslab);
    hipLaunchKernelGGL((k_s1d_feature_rows<2, true>), dim3(blocks), dim3(64 * FD_WAVES), fd_smem_bytes(F), 0, n, F, H, H, b.X, (long)F, b.ref, b.W1,
                       defer ? (const double *)nullptr : b.cref, b.S1d, fd_hint_cap(F), b.hint, nslab, b.slabs, b.zstate, defer ? b.S1x : (float *)nullptr,
                       defer ? b.gate : (unsigned *)nullptr, b.cref, b.S1qs, lt_bits_job{}, 0, g_flags, g_stagger);
}
static int g_cus = 256;
static int g_nsl = 0;
static int ring_rowblocks(int n) { return (int)fd_ring_grid(n, g_nsl); }      // (all blocks: the slab blocks take rows too)
static void launch_ring(int n, int F, int H, const Bufs &b, bool defer) {
    const int nch = fr_chunks(F), nsl = defer ? (F + 63) / 64 : 0;
    g_nsl = nsl;
    static int g_parity = 0;
    const int parity = g_parity;
    g_parity ^= 1;
    const unsigned grid = fd_ring_grid(n, nsl);
#define LAB_FR(N_)                                                                                                             \
    hipLaunchKernelGGL(k_s1d_feature_ring<N_>, dim3(grid), dim3(64 * FR_WAVES), fr_smem_bytes(nch), 0, n, F, H, b.X, (long)F, b.ref, b.W1,  \
                       defer ? (const double *)nullptr : b.cref, b.S1d, fd_hint_cap(F) < FR_CAP ? fd_hint_cap(F) : FR_CAP, b.hint, nsl, b.slabs,  \
                       b.zstate, defer ? b.S1x : (float *)nullptr, b.gate, b.cref, b.S1qs, parity)
    switch (nch) {
    case 9: LAB_FR(9); break;
    case 10: LAB_FR(10); break;
    case 11: LAB_FR(11); break;
    case 12: LAB_FR(12); break;
    case 13: LAB_FR(13); break;
    default: std::printf("no ring kernel for %d chunks per row\n", nch); exit(2);
    }
#undef LAB_FR
}

template <class L>
static void time_it(const char *name, L launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> one;
    for (int i = 0; i < 20; ++i) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        one.push_back(t);
    }
    std::sort(one.begin(), one.end());
    std::printf("%-44s %7.2f us per launch (50 back to back)   %7.2f us median single launch by events\n", name, ms * 1e3 / 50, one[10] * 1e3);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4385, F = argc > 2 ? atoi(argv[2]) : 3170, H = argc > 3 ? atoi(argv[3]) : 256;
    const float dens = argc > 4 ? (float)atof(argv[4]) : 0.006f;
    const bool special = argc > 5 ? atoi(argv[5]) != 0 : true;      // 0: no dense row (its piecewise path IS the launch time)
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::printf("n = %d, F = %d, H = %d, density %.4f, %d CUs, ring LDS %zu B, %d chunks per row\n", n, F, H, dens, g_cus, fr_smem_bytes(fr_chunks(F)), fr_chunks(F));
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> hx((size_t)n * F), href((size_t)F + FD_REF_PAD, 0.f), hw((size_t)F * H);
    for (int j = 0; j < F; ++j) href[j] = -0.0776f - 1e-4f * (j % 7);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < F; ++j) hx[(size_t)i * F + j] = U(rng) < dens ? 12.88f + 0.01f * (j % 5) : href[j];
    // a few special rows: empty difference, a dense one (beyond the list), differences in the first / last columns
    if (n > 8) {
        for (int j = 0; j < F; ++j) hx[(size_t)3 * F + j] = href[j];
        if (special) for (int j = 0; j < F; ++j) hx[(size_t)5 * F + j] = (j % 3 == 0) ? 1.f + 0.001f * j : href[j];
        hx[(size_t)6 * F + 0] = 3.f; hx[(size_t)6 * F + 1] = 4.f; hx[(size_t)6 * F + F - 1] = 5.f; hx[(size_t)6 * F + F - 2] = 6.f;
        hx[(size_t)7 * F + 0] = 3.f; hx[(size_t)7 * F + 1] = 4.f; hx[(size_t)7 * F + F - 1] = 5.f; hx[(size_t)7 * F + F - 2] = 6.f;
        hx[(size_t)(n - 1) * F + F - 1] = 7.f; hx[(size_t)(n - 1) * F + 0] = 8.f;
    }
    for (auto &v : hw) v = U(rng) - 0.5f;
    Bufs b;
    hipMalloc(&b.X, hx.size() * 4); hipMalloc(&b.ref, href.size() * 4); hipMalloc(&b.W1, hw.size() * 4);
    hipMalloc(&b.S1x, (size_t)n * H * 4); hipMalloc(&b.S1d, (size_t)n * H * 8); hipMalloc(&b.slabs, fd_slab_doubles(F, H) * 8);
    hipMalloc(&b.cref, (size_t)H * 8); hipMalloc(&b.S1qs, (size_t)n * 8); hipMalloc(&b.hint, 4); hipMalloc(&b.zstate, (size_t)n * 4);
    hipMalloc(&b.gate, FR_GATE_WORDS * 4); hipMemset(b.gate, 0, FR_GATE_WORDS * 4); hipMemset(b.hint, 0, 4);
    hipMemcpy(b.X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b.ref, href.data(), href.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b.W1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    if (fd_ring_allow_lds() != LT_OK) { std::printf("hipFuncSetAttribute failed\n"); return 1; }

    // host reference: full fp64 rows (exact products, fp64 sum in column order)
    const int nchk = std::min(n, 64);
    std::vector<int> rows;
    for (int i = 0; i < nchk; ++i) rows.push_back(i < 16 ? i : (int)((long)i * (n - 1) / (nchk - 1)));
    rows.push_back(n - 1);
    std::vector<double> want(rows.size() * H);
    for (size_t q = 0; q < rows.size(); ++q)
        for (int c = 0; c < H; ++c) {
            double a = 0.0;
            for (int j = 0; j < F; ++j) a += (double)hx[(size_t)rows[q] * F + j] * (double)hw[(size_t)j * H + c];
            want[q * H + c] = a;
        }
    auto fetch = [&](bool defer, std::vector<double> &out) {
        std::vector<float> sx((size_t)n * H);
        std::vector<double> sq(n), cr(H), sd((size_t)n * H);
        out.assign((size_t)n * H, 0.0);
        hipDeviceSynchronize();
        if (defer) {
            hipMemcpy(sx.data(), b.S1x, sx.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(sq.data(), b.S1qs, sq.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(cr.data(), b.cref, cr.size() * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < H; ++c) out[(size_t)i * H + c] = (double)reinterpret_cast<const int *>(sx.data())[(size_t)i * H + c] * sq[i] + cr[c];
        } else {
            hipMemcpy(out.data(), b.S1d, out.size() * 8, hipMemcpyDeviceToHost);
        }
    };
    auto check = [&](const char *name, const std::vector<double> &got) {
        double worst = 0.0, scale = 0.0;
        for (size_t q = 0; q < rows.size(); ++q)
            for (int c = 0; c < H; ++c) {
                worst = std::max(worst, std::fabs(got[(size_t)rows[q] * H + c] - want[q * H + c]));
                scale = std::max(scale, std::fabs(want[q * H + c]));
            }
        std::printf("  %-40s max |got - host fp64| over %zu rows = %.3e (largest value %.3e)\n", name, rows.size(), worst, scale);
    };
    auto cmp = [&](const char *name, const std::vector<double> &a, const std::vector<double> &c) {
        double worst = 0.0;
        size_t at = 0;
        for (size_t i = 0; i < a.size(); ++i) { const double d = std::fabs(a[i] - c[i]); if (d > worst) { worst = d; at = i; } }
        std::printf("  %-40s max |ring - rows| over all rows = %.3e (row %zu)\n", name, worst, at / H);
    };
    int hint = 0;
    for (int defer = 1; defer >= 0; --defer) {
        std::vector<double> a, r;
        if (!defer) {       // cref for the non-deferred form: from the deferred run above (left in b.cref)
        }
        hipMemset(b.S1x, 0xff, (size_t)n * H * 4); hipMemset(b.S1d, 0xff, (size_t)n * H * 8); hipMemset(b.hint, 0, 4);
        launch_old(n, F, H, b, defer);
        fetch(defer, a);
        hipMemcpy(&hint, b.hint, 4, hipMemcpyDeviceToHost);
        const int hint_old = hint;
        hipMemset(b.S1x, 0xff, (size_t)n * H * 4); hipMemset(b.S1d, 0xff, (size_t)n * H * 8); hipMemset(b.hint, 0, 4);
        if (defer) hipMemset(b.cref, 0xff, (size_t)H * 8);
        launch_ring(n, F, H, b, defer);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { std::printf("ring launch failed: %s\n", hipGetErrorString(e)); return 1; }
        fetch(defer, r);
        hipMemcpy(&hint, b.hint, 4, hipMemcpyDeviceToHost);
        std::printf("%s form (dense hint: rows %d, ring %d):\n", defer ? "deferred cref, fixed-point rows" : "cref in the rows, fp64 rows", hint_old, hint);
        check("rows kernel", a);
        check("ring kernel", r);
        cmp(defer ? "dequantised rows + cref" : "fp64 rows", r, a);
        // twice the same launch: the same bits
        std::vector<double> r2;
        launch_ring(n, F, H, b, defer);
        fetch(defer, r2);
        std::printf("  ring kernel run twice: %s\n", r2 == r ? "same bits" : "BITS DIFFER");
    }
    {   // timeline of the ring kernel (deferred form): stamps 0 row begun, 1 landed, 2 list made, 3 walked, 4 finished; per wave and row slot
        const size_t nw = (size_t)ring_rowblocks(n) * FR_WAVES;
        unsigned long long *trace;
        hipMalloc(&trace, nw * FR_SLOTS * 8 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_fr_trace), &trace, sizeof(trace));
        std::vector<unsigned long long> h(nw * FR_SLOTS * 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(trace, 0, nw * FR_SLOTS * 8 * 8);
            hipDeviceSynchronize();
            launch_ring(n, F, H, b, true);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < h.size(); ++i) if (h[i]) { t0 = std::min(t0, h[i]); t1 = std::max(t1, h[i]); }
        std::printf("ring timeline: first stamp -> last stamp %.2f us\n", (t1 - t0) * 0.01);
        const char *nm[5] = {"row begun", "landed", "list made", "walked", "finished"};
        for (int slot = 0; slot < FR_SLOTS; ++slot) {
            for (int k = 0; k < 5; ++k) {
                std::vector<double> v;
                for (size_t w = 0; w < nw; ++w) if (h[(w * FR_SLOTS + slot) * 8 + k]) v.push_back((h[(w * FR_SLOTS + slot) * 8 + k] - t0) * 0.01);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                const size_t m = v.size();
                std::printf("   row %d of a wave, %-10s (%5zu waves) us since first stamp: min %5.2f p10 %5.2f p50 %5.2f p90 %5.2f max %5.2f\n", slot, nm[k], m,
                            v[0], v[m / 10], v[m / 2], v[9 * m / 10], v[m - 1]);
            }
        }
        {   // when does a workgroup begin, and when is it done
            std::vector<std::pair<double, int>> st;
            std::vector<double> en;
            for (int bk = 0; bk < ring_rowblocks(n); ++bk) {
                unsigned long long a = ~0ull, z = 0;
                for (int w = 0; w < FR_WAVES; ++w)
                    for (int slot = 0; slot < FR_SLOTS; ++slot)
                        for (int k = 0; k < 5; ++k) { const unsigned long long t = h[(((size_t)bk * FR_WAVES + w) * FR_SLOTS + slot) * 8 + k]; if (t) { a = std::min(a, t); z = std::max(z, t); } }
                if (z) { st.push_back({(a - t0) * 0.01, bk}); en.push_back((z - t0) * 0.01); }
            }
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            std::printf("   workgroup begins: p50 %.2f p75 %.2f p90 %.2f p95 %.2f max %.2f us; ends: p10 %.2f p50 %.2f p90 %.2f max %.2f us; the 12 latest to begin:", st[st.size() / 2].first,
                        st[3 * st.size() / 4].first, st[9 * st.size() / 10].first, st[95 * st.size() / 100].first, st.back().first, en[en.size() / 10], en[en.size() / 2], en[9 * en.size() / 10], en.back());
            for (size_t i = st.size() - 12; i < st.size(); ++i) std::printf(" %d@%.1f", st[i].second, st[i].first);
            std::printf("\n");
        }
        trace = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_fr_trace), &trace, sizeof(trace));
    }
    time_it("k_s1d_feature_rows<2> (deferred, fixed point)", [&] { launch_old(n, F, H, b, true); });
    {   // the flag-bit form of the row-per-wave kernel against its ballot form
        std::vector<double> a0, a1;
        g_flags = 0; launch_old(n, F, H, b, true); fetch(true, a0);
        g_flags = 1; launch_old(n, F, H, b, true); fetch(true, a1);
        check("rows kernel, flag bits", a1);
        cmp("rows kernel: flag bits - ballots", a1, a0);
        time_it("k_s1d_feature_rows<2> flag bits (deferred)", [&] { launch_old(n, F, H, b, true); });
        g_flags = 0;
    }
    {   // staggered start of the row blocks: groups x gap (us)
        std::vector<double> a0, a1;
        g_stagger = 0; launch_old(n, F, H, b, true); fetch(true, a0);
        for (int groups : {2, 3, 4, 6, 8})
            for (int gap_us10 : {10, 15, 20, 30, 40, 60}) {      // 0.1 us units
                g_stagger = (gap_us10 * 10) << 8 | groups;
                launch_old(n, F, H, b, true); fetch(true, a1);
                char nm[96];
                snprintf(nm, sizeof nm, "rows<2> stagger %d groups x %.1f us%s", groups, gap_us10 * 0.1, a1 == a0 ? "" : "  BITS DIFFER");
                time_it(nm, [&] { launch_old(n, F, H, b, true); });
            }
        g_stagger = 0;
    }
    time_it("k_s1d_feature_ring    (deferred, fixed point)", [&] { launch_ring(n, F, H, b, true); });
    time_it("k_s1d_feature_rows<2> (cref in the rows, fp64)", [&] { launch_old(n, F, H, b, false); });
    time_it("k_s1d_feature_ring    (cref in the rows, fp64)", [&] { launch_ring(n, F, H, b, false); });
    return 0;
}
