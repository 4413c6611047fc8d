// spmm_lab -- development harness for the CSR SpMM  out[n,H] = A_hat * S  on R-MAT graphs (BASELINE configs[4]
// shape).  Not part of the product: it times candidate kernels against the library's lt_spmm_csr_f32 on the same
// device arrays and checks them bit for bit against it (every variant keeps the k-ordered fmaf chain per output
// column).  Build: make -C tools/spmm_lab ; run: tools/spmm_lab/spmm_lab <scale> [edge_factor] [H] [reps] [variants]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>
#include <algorithm>
#include <functional>
#include <type_traits>
#include <chrono>
#include <random>
#include <string>
#include <vector>

#include "linkteller_hip.h"

#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 fma4(float a, f32x4 s, f32x4 acc) {
    acc.x = fmaf(a, s.x, acc.x); acc.y = fmaf(a, s.y, acc.y);
    acc.z = fmaf(a, s.z, acc.z); acc.w = fmaf(a, s.w, acc.w);
    return acc;
}

// lane k of the GL-lane group this lane belongs to
template <int GL>
__device__ __forceinline__ int bcast_i(int x, int k) {
    if constexpr (GL == 64) return __builtin_amdgcn_readlane(x, k);
    else return __shfl(x, k, GL);
}
template <int GL, int K>
__device__ __forceinline__ int bcast_c(int x) {
    if constexpr (GL == 16) return __builtin_amdgcn_update_dpp(0, x, 0x150 + K, 0xf, 0xf, false);   // row_newbcast:K
    else return bcast_i<GL>(x, K);
}

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
}

// ------------------------------------------------------------------------------------------------
// work items: (first entry, entry count, destination row in `out` -- or n + segment id in `partial`)
// sorted by length class, longest first; a row of more than SEG entries is cut into segments
// ------------------------------------------------------------------------------------------------
struct Work {
    int n_items = 0;
    int32_t *e0 = nullptr, *cnt = nullptr, *dst = nullptr;   // device
    int n_long = 0, n_seg = 0;
    int32_t *long_row = nullptr, *long_segptr = nullptr;       // device
    float *partial = nullptr;                                  // [n_seg, H]
};

// Sliced SpMM.  A GL-lane group owns one work item (row / segment) x one slice of 4*GL columns; the slice is
// chosen from the XCD the block lands on (blocks are dealt round-robin over the 8 XCDs), so an XCD's L2 only
// ever sees 1/NS of S.
// cache policy of a gather.  POL 0: plain; 1: nt; 2: sc1; 3: sc0 sc1; 4: sc1 nt
template <int POL>
__device__ __forceinline__ void gload(f32x4 &d, unsigned off, const char *base) {
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(d) : "v"(off), "s"(base) : "memory");
    else if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, %2 sc1 nt" : "=v"(d) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}

// HINT 0: compiler-managed plain gathers.  HINT > 0: bit 31 of a column index marks a COLD column (few readers): its
// gather uses cache policy HINT, hot columns use plain loads; all gathers are inline asm with one explicit wait.
// SEQ: blocks walk the slices one after the other (all XCDs on one slice at a time) instead of slice = f(XCD).
template <int GL, int U, bool NT, bool SORTED, int HINT = 0, bool SEQ = false, bool PF = false>
__global__ __launch_bounds__(256) void k_spmm_sliced(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt,
    const int32_t *__restrict__ w_dst, const int32_t *__restrict__ rowptr, int n,
    const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S, int lds, int ncols, float *__restrict__ out, int ldo,
    float *__restrict__ partial, int ldp, int ns /* slices */, int slice_fixed = -1) {
    constexpr int GPW = 64 / GL;            // groups per wave
    constexpr int IPB = 4 * GPW;            // items per block
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int j = lane & (GL - 1);
    int slice, chunk;
    if (slice_fixed >= 0) {                 // (round 6: one launch per slice -- the slices strictly one after the other)
        slice = slice_fixed;
        chunk = blockIdx.x;
    } else if (SEQ) {
        const int bps = gridDim.x / ns;     // blocks per slice
        slice = blockIdx.x / bps;
        chunk = blockIdx.x % bps;
    } else {
        const int xcd = blockIdx.x & 7;
        const int q = blockIdx.x >> 3;
        const int xps = 8 / ns;             // XCDs per slice
        slice = xcd % ns;
        chunk = q * xps + xcd / ns;
    }
    int it = chunk * IPB + wv * GPW + lane / GL;
    if (GL == 64) it = __builtin_amdgcn_readfirstlane(it);
    if (it >= n_items) return;
    int e0, cnt, dst;
    if (SORTED) { e0 = w_e0[it]; cnt = w_cnt[it]; dst = w_dst[it]; }
    else { e0 = rowptr[it]; cnt = rowptr[it + 1] - e0; dst = it; }
    const int coff = slice * 4 * GL + 4 * j;
    const bool active = coff < ncols;
    const unsigned rowbytes = (unsigned)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const char *Sb = reinterpret_cast<const char *>(S);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int e1 = e0 + cnt;
    int nxc = 0;
    float nxa = 0.f;
    if (PF && e0 + j < e1) { nxc = __builtin_nontemporal_load(col + e0 + j); nxa = __builtin_nontemporal_load(val + e0 + j); }
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        int myc = 0;
        float mya = 0.f;
        if (PF) {
            myc = nxc; mya = nxa;
            nxc = 0; nxa = 0.f;
            if (me + GL < e1) { nxc = __builtin_nontemporal_load(col + me + GL); nxa = __builtin_nontemporal_load(val + me + GL); }
        } else if (me < e1) {
            if (NT) { myc = __builtin_nontemporal_load(col + me); mya = __builtin_nontemporal_load(val + me); }
            else { myc = col[me]; mya = val[me]; }
        }
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                f32x4 s[U];
                float a[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int ch = bcast_c<GL, k>(myc);
                    const int c = HINT ? (ch & 0x7fffffff) : ch;
                    a[u] = __builtin_bit_cast(float, bcast_c<GL, k>(__builtin_bit_cast(int, mya)));
                    s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (k < left && active) {
                        if constexpr (HINT == 0) {
                            s[u] = *reinterpret_cast<const f32x4 *>(Sb + (size_t)((unsigned)c * rowbytes + loff));
                        } else {
                            const unsigned off = (unsigned)c * rowbytes + loff;
                            if (ch < 0) gload<HINT>(s[u], off, Sb);
                            else gload<0>(s[u], off, Sb);
                        }
                    }
                });
                if constexpr (HINT != 0) {
                    static_assert(U == 8 || U == 4, "explicit wait written for 4 or 8 gathers");
                    if constexpr (U == 8)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
                    else
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
                }
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    if (kb + u < left) acc = fma4(a[u], s[u], acc);
                });
            }
        });
    }
    if (!active) return;
    float *d = dst < n ? out + (size_t)dst * ldo + coff : partial + (size_t)(dst - n) * ldp + coff;
    if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(d));
    else *reinterpret_cast<f32x4 *>(d) = acc;
}

// ------------------------------------------------------------------------------------------------
// GATHER CEILING (round 4): the sliced kernel above with everything but its gathers removed -- the SAME work items in
// the SAME order, the SAME column stream (col only: 4 B per entry against 4 * GL * 4 B gathered), the same
// slice = f(XCD) placement and U gathers in flight per lane -- no val stream, no fmaf chain (the loaded words are
// XOR-folded, one VALU op each, so that the loads stay live), no result rows (one word per item and slice, 1/64 of the
// real stores).  What it measures is what the memory system delivers for this index stream at this hit distribution:
// no row-gather SpMM that issues these gathers can run faster, whatever its arithmetic.
// ------------------------------------------------------------------------------------------------
template <int GL, int U>
__global__ __launch_bounds__(256) void k_gather_ceiling(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt, const int32_t *__restrict__ col,
    const float *__restrict__ S, int lds, int ncols, unsigned *__restrict__ sink, int ns) {
    constexpr int GPW = 64 / GL;
    constexpr int IPB = 4 * GPW;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int j = lane & (GL - 1);
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int xps = 8 / ns;
    const int slice = xcd % ns;
    const int chunk = q * xps + xcd / ns;
    int it = chunk * IPB + wv * GPW + lane / GL;
    if (GL == 64) it = __builtin_amdgcn_readfirstlane(it);
    if (it >= n_items) return;
    const int e0 = w_e0[it], cnt = w_cnt[it];
    const int coff = slice * 4 * GL + 4 * j;
    const bool active = coff < ncols;
    const unsigned rowbytes = (unsigned)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const char *Sb = reinterpret_cast<const char *>(S);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 x = {0u, 0u, 0u, 0u};
    const int e1 = e0 + cnt;
    int nxc = 0;
    if (e0 + j < e1) nxc = __builtin_nontemporal_load(col + e0 + j);
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        nxc = 0;
        if (me + GL < e1) nxc = __builtin_nontemporal_load(col + me + GL);
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                u32x4 s[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = bcast_c<GL, k>(myc);
                    s[u] = u32x4{0u, 0u, 0u, 0u};
                    if (k < left && active) s[u] = *reinterpret_cast<const u32x4 *>(Sb + (size_t)((unsigned)c * rowbytes + loff));
                });
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    x ^= s[u];
                });
            }
        });
    }
    if (active && j == 0) sink[(size_t)it * ns + slice] = x.x ^ x.y ^ x.z ^ x.w;
}


// ------------------------------------------------------------------------------------------------
// Round 5: EIGHT column slices of 128 B -- one per XCD, so that an XCD's 4 MiB L2 holds twice the rows
// of S (32 768 slices instead of 16 384) -- gathered as 16 lanes x 8 B, so that an item costs the instruction count of the
// 4-slice kernel (the round-4 attempt at narrower slices used 8 lanes x 16 B: twice the items per wave-instruction's bytes).
// Same work items, order, prefetched (col, val) stream and fmaf chains as k_spmm_sliced<16, U, .., PF = true>.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(256) void k_spmm_sliced8(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt,
    const int32_t *__restrict__ w_dst, int n, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S, int lds, int ncols, float *__restrict__ out, int ldo,
    float *__restrict__ partial, int ldp) {
    constexpr int GL = 16, GPW = 4, IPB = 16;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int j = lane & (GL - 1);
    const int slice = blockIdx.x & 7;           // slice = XCD
    const int chunk = blockIdx.x >> 3;
    const int it = chunk * IPB + wv * GPW + lane / GL;
    if (it >= n_items) return;
    const int e0 = w_e0[it], cnt = w_cnt[it], dst = w_dst[it];
    const int coff = slice * 2 * GL + 2 * j;
    const bool active = coff < ncols;
    const unsigned rowbytes = (unsigned)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const char *Sb = reinterpret_cast<const char *>(S);
    f32x2 acc = {0.f, 0.f};
    const int e1 = e0 + cnt;
    int nxc = 0;
    float nxa = 0.f;
    if (e0 + j < e1) { nxc = __builtin_nontemporal_load(col + e0 + j); nxa = __builtin_nontemporal_load(val + e0 + j); }
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        const float mya = nxa;
        nxc = 0; nxa = 0.f;
        if (me + GL < e1) { nxc = __builtin_nontemporal_load(col + me + GL); nxa = __builtin_nontemporal_load(val + me + GL); }
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                f32x2 s[U];
                float a[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = bcast_c<GL, k>(myc);
                    a[u] = __builtin_bit_cast(float, bcast_c<GL, k>(__builtin_bit_cast(int, mya)));
                    s[u] = f32x2{0.f, 0.f};
                    if (k < left && active) s[u] = *reinterpret_cast<const f32x2 *>(Sb + (size_t)((unsigned)c * rowbytes + loff));
                });
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    if (kb + u < left) { acc.x = fmaf(a[u], s[u].x, acc.x); acc.y = fmaf(a[u], s[u].y, acc.y); }
                });
            }
        });
    }
    if (!active) return;
    float *d = dst < n ? out + (size_t)dst * ldo + coff : partial + (size_t)(dst - n) * ldp + coff;
    __builtin_nontemporal_store(acc, reinterpret_cast<f32x2 *>(d));
}

template <int U>
__global__ __launch_bounds__(256) void k_gather_ceiling8(
    int n_items, const int32_t *__restrict__ w_e0, const int32_t *__restrict__ w_cnt, const int32_t *__restrict__ col,
    const float *__restrict__ S, int lds, int ncols, unsigned *__restrict__ sink) {
    constexpr int GL = 16, GPW = 4, IPB = 16;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int j = lane & (GL - 1);
    const int slice = blockIdx.x & 7;
    const int chunk = blockIdx.x >> 3;
    const int it = chunk * IPB + wv * GPW + lane / GL;
    if (it >= n_items) return;
    const int e0 = w_e0[it], cnt = w_cnt[it];
    const int coff = slice * 2 * GL + 2 * j;
    const bool active = coff < ncols;
    const unsigned rowbytes = (unsigned)lds * 4u;
    const unsigned loff = (unsigned)coff * 4u;
    const char *Sb = reinterpret_cast<const char *>(S);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 x = {0u, 0u};
    const int e1 = e0 + cnt;
    int nxc = 0;
    if (e0 + j < e1) nxc = __builtin_nontemporal_load(col + e0 + j);
    for (int eb = e0; eb < e1; eb += GL) {
        const int me = eb + j;
        const int myc = nxc;
        nxc = 0;
        if (me + GL < e1) nxc = __builtin_nontemporal_load(col + me + GL);
        const int left = e1 - eb;
        static_for<GL / U>([&](auto kbt) {
            constexpr int kb = decltype(kbt)::value * U;
            if (kb < left) {
                u32x2 s[U];
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    constexpr int k = kb + u;
                    const int c = bcast_c<GL, k>(myc);
                    s[u] = u32x2{0u, 0u};
                    if (k < left && active) s[u] = *reinterpret_cast<const u32x2 *>(Sb + (size_t)((unsigned)c * rowbytes + loff));
                });
                static_for<U>([&](auto ut) {
                    constexpr int u = decltype(ut)::value;
                    x ^= s[u];
                });
            }
        });
    }
    if (active && j == 0) sink[(size_t)it * 8 + slice] = x.x ^ x.y;
}

// long rows: partials added in segment order
__global__ void k_combine(int n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_segptr,
                          const float *__restrict__ partial, int ldp, int ncols, float *__restrict__ out, int ldo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_long * ncols) return;
    const int li = i / ncols, c = i % ncols;
    float acc = partial[(size_t)long_segptr[li] * ldp + c];
    for (int sg = long_segptr[li] + 1; sg < long_segptr[li + 1]; ++sg) acc += partial[(size_t)sg * ldp + c];
    out[(size_t)long_row[li] * ldo + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// host: R-MAT graph -> symmetric, deduplicated, + I, FirstOrderGCN-like values
// ------------------------------------------------------------------------------------------------
struct HostCsr {
    int n = 0;
    std::vector<int32_t> rowptr, col;
    std::vector<float> val;
};

static HostCsr make_rmat(int scale, long draws, uint64_t seed) {
    const int n = 1 << scale;
    std::mt19937_64 rng(seed);
    std::vector<uint64_t> keys;
    keys.reserve((size_t)draws * 2 + n);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (long i = 0; i < draws; ++i) {
        uint64_t r = 0, c = 0;
        for (int b = 0; b < scale; ++b) {
            const double x = U(rng);
            // a 0.57 -> (0,0), b 0.19 -> (0,1), c 0.19 -> (1,0), d 0.05 -> (1,1)
            const int down = x >= 0.76, right = (x >= 0.57 && x < 0.76) || x >= 0.95;
            r |= (uint64_t)down << b;
            c |= (uint64_t)right << b;
        }
        if (r == c) continue;
        keys.push_back(r << 32 | c);
        keys.push_back(c << 32 | r);
    }
    for (int i = 0; i < n; ++i) keys.push_back((uint64_t)i << 32 | (uint64_t)i);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    HostCsr g;
    g.n = n;
    g.rowptr.assign((size_t)n + 1, 0);
    g.col.resize(keys.size());
    g.val.resize(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        g.rowptr[(keys[i] >> 32) + 1]++;
        g.col[i] = (int32_t)(keys[i] & 0xffffffffu);
    }
    for (int i = 0; i < n; ++i) g.rowptr[i + 1] += g.rowptr[i];
    std::vector<float> dinv((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int d = g.rowptr[i + 1] - g.rowptr[i] - 1;
        dinv[i] = d > 0 ? 1.0f / sqrtf((float)d) : 0.f;
    }
    for (int r = 0; r < n; ++r)
        for (int e = g.rowptr[r]; e < g.rowptr[r + 1]; ++e)
            g.val[e] = g.col[e] == r ? 1.0f : dinv[r] * dinv[g.col[e]];
    return g;
}

// a graph WITH locality: every node draws its neighbours within +-window of its own index (symmetrised, + I)
static HostCsr make_banded(int scale, int deg, int window, uint64_t seed) {
    const int n = 1 << scale;
    std::mt19937_64 rng(seed);
    std::vector<uint64_t> keys;
    keys.reserve((size_t)n * (deg + 1) * 2);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < deg / 2; ++k) {
            long j = (long)i + (long)(rng() % (2 * window + 1)) - window;
            if (j < 0) j += n;
            if (j >= n) j -= n;
            if (j == i) continue;
            keys.push_back((uint64_t)i << 32 | (uint64_t)j);
            keys.push_back((uint64_t)j << 32 | (uint64_t)i);
        }
        keys.push_back((uint64_t)i << 32 | (uint64_t)i);
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    HostCsr g;
    g.n = n;
    g.rowptr.assign((size_t)n + 1, 0);
    g.col.resize(keys.size());
    g.val.resize(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        g.rowptr[(keys[i] >> 32) + 1]++;
        g.col[i] = (int32_t)(keys[i] & 0xffffffffu);
    }
    for (int i = 0; i < n; ++i) g.rowptr[i + 1] += g.rowptr[i];
    for (int r = 0; r < n; ++r)
        for (int e = g.rowptr[r]; e < g.rowptr[r + 1]; ++e)
            g.val[e] = g.col[e] == r ? 1.0f : 1.0f / (float)(g.rowptr[r + 1] - g.rowptr[r]);
    return g;
}

template <typename T>
static T *upload(const std::vector<T> &v, size_t pad = 0) {
    T *d = nullptr;
    CK(hipMalloc((void **)&d, (v.size() + pad) * sizeof(T) + 16));
    CK(hipMemset(d, 0, (v.size() + pad) * sizeof(T) + 16));
    if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static Work build_work(const HostCsr &g, int H, int SEG, int ordering) {
    Work w;
    struct Item { int32_t e0, cnt, dst; };
    std::vector<Item> items;
    std::vector<int32_t> lrow, lptr(1, 0);
    int nseg = 0;
    for (int r = 0; r < g.n; ++r) {
        const int d = g.rowptr[r + 1] - g.rowptr[r];
        if (d <= SEG) { items.push_back({g.rowptr[r], d, r}); continue; }
        lrow.push_back(r);
        for (int b = g.rowptr[r]; b < g.rowptr[r + 1]; b += SEG) {
            items.push_back({b, std::min(SEG, g.rowptr[r + 1] - b), g.n + nseg});
            ++nseg;
        }
        lptr.push_back(nseg);
    }
    if (ordering == 0) {
        // length classes (ceil(cnt / 16) blocks), longest first, natural order inside a class
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return (a.cnt + 15) / 16 > (b.cnt + 15) / 16; });
    } else {
        // segments of long rows first, by the column their first entry reads (concurrent waves then gather from one
        // sliding window of S); then the short rows by length class
        const int n = g.n;
        const int32_t *colp = g.col.data();
        std::stable_sort(items.begin(), items.end(), [n, colp](const Item &a, const Item &b) {
            const bool sa = a.dst >= n, sb = b.dst >= n;
            if (sa != sb) return sa;
            if (sa) return colp[a.e0] < colp[b.e0];
            return (a.cnt + 15) / 16 > (b.cnt + 15) / 16;
        });
        if (ordering == 3) {
            // short rows: by length class, then by the column of their first entry (does a sweep help them as it helps the segments?)
            std::stable_sort(items.begin(), items.end(), [n, colp](const Item &a, const Item &b) {
                const bool sa = a.dst >= n, sb = b.dst >= n;
                if (sa != sb) return sa;
                if (sa) return false;
                const int ca = (a.cnt + 15) / 16, cb = (b.cnt + 15) / 16;
                if (ca != cb) return ca > cb;
                return colp[a.e0 + a.cnt / 2] < colp[b.e0 + b.cnt / 2];     // median column
            });
        }
        if (ordering == 4) {
            // short rows FIRST (hot columns stay in L2 while nothing sweeps it), the segments after them
            std::stable_partition(items.begin(), items.end(), [n](const Item &a) { return a.dst < n; });
        }
        if (ordering == 2) {
            // blocks of 16 items are dealt to a slice's two XCDs alternately (chunk parity): give the even chunks the
            // first half of the column-ordered segments and the odd chunks the second half
            size_t nseg_items = 0;
            while (nseg_items < items.size() && items[nseg_items].dst >= n) ++nseg_items;
            const size_t blocks = nseg_items / 16, hb = blocks / 2;
            std::vector<Item> re(items.begin(), items.end());
            for (size_t bi = 0; bi < 2 * hb; ++bi) {
                const size_t src = (bi & 1) ? hb + bi / 2 : bi / 2;
                for (int k = 0; k < 16; ++k) re[bi * 16 + k] = items[src * 16 + k];
            }
            items.swap(re);
        }
    }
    std::vector<int32_t> e0(items.size()), cnt(items.size()), dst(items.size());
    for (size_t i = 0; i < items.size(); ++i) { e0[i] = items[i].e0; cnt[i] = items[i].cnt; dst[i] = items[i].dst; }
    w.n_items = (int)items.size();
    w.e0 = upload(e0); w.cnt = upload(cnt); w.dst = upload(dst);
    w.n_long = (int)lrow.size(); w.n_seg = nseg;
    w.long_row = upload(lrow); w.long_segptr = upload(lptr);
    CK(hipMalloc((void **)&w.partial, (size_t)std::max(nseg, 1) * H * sizeof(float)));
    return w;
}

struct Ctx {
    HostCsr *g; int H;
    int32_t *rowptr, *col; float *val, *S, *out, *ref;
    int32_t *colh[4];   // col with bit 31 set on cold columns, for hot sets of 8K / 12K / 16K / 24K columns
    Work work;
    Work work_col;   // segments ordered by first column
    Work work_half;  // ... and dealt so that each of a slice's two XCDs sweeps its own half of the column range
    Work work_o3, work_o4;   // round-4 orderings (build_work)
    lt_graph *lg;
    unsigned *sink;  // [n_items * 8] of the gather ceilings
};

template <int GL, int U, int ORDER = 2>
static void run_ceiling(Ctx &c, hipStream_t st) {
    Work &W = ORDER == 4 ? c.work_o4 : (ORDER == 3 ? c.work_o3 : (ORDER == 2 ? c.work_half : (ORDER ? c.work_col : c.work)));
    const int ns = c.H / (4 * GL) > 0 ? (c.H + 4 * GL - 1) / (4 * GL) : 1;
    const int xps = 8 / ns;
    constexpr int IPB = 4 * (64 / GL);
    const int chunks = (W.n_items + IPB - 1) / IPB;
    const int grid = 8 * ((chunks + xps - 1) / xps);
    hipLaunchKernelGGL((k_gather_ceiling<GL, U>), dim3(grid), dim3(256), 0, st, W.n_items, W.e0, W.cnt, c.col, c.S, c.H, c.H,
                       c.sink, ns);
}

template <int GL, int U, bool NT, bool SORTED, int HINT = 0, bool SEQ = false, int HOTSET = 0, int ORDER = 0, bool PF = false>
static void run_sliced(Ctx &c, hipStream_t st) {
    Work &W = ORDER == 2 ? c.work_half : (ORDER ? c.work_col : c.work);
    const int ns = c.H / (4 * GL) > 0 ? (c.H + 4 * GL - 1) / (4 * GL) : 1;
    const int xps = 8 / ns;
    constexpr int IPB = 4 * (64 / GL);
    const int n_items = SORTED ? W.n_items : c.g->n;
    const int chunks = (n_items + IPB - 1) / IPB;
    const int grid = SEQ ? chunks * ns : 8 * ((chunks + xps - 1) / xps);
    const int32_t *colp = HINT ? c.colh[HOTSET] : c.col;
    hipLaunchKernelGGL((k_spmm_sliced<GL, U, NT, SORTED, HINT, SEQ, PF>), dim3(grid), dim3(256), 0, st, n_items, W.e0, W.cnt,
                       W.dst, c.rowptr, c.g->n, colp, c.val, c.S, c.H, c.H, c.out, c.H, W.partial, c.H, ns);
    if (SORTED && W.n_long > 0) {
        const int tot = W.n_long * c.H;
        hipLaunchKernelGGL(k_combine, dim3((tot + 255) / 256), dim3(256), 0, st, W.n_long, W.long_row,
                           W.long_segptr, W.partial, c.H, c.H, c.out, c.H);
    }
}


// TEMPORAL slicing (round 6): the column slices one after the other with ALL XCDs on the same slice, so that a
// pass's gather working set (n x 256 B = 537 MB at scale 21, n x 128 B = 268 MB) sits in front of the 256 MiB Infinity Cache
// instead of all slices' at once; the index stream is re-read per pass.  One launch per slice (strictly sequential).
template <int GL, int U, int ORDER = 2>
static void run_temporal(Ctx &c, hipStream_t st) {
    Work &W = ORDER == 2 ? c.work_half : (ORDER ? c.work_col : c.work);
    const int ns = (c.H + 4 * GL - 1) / (4 * GL);
    constexpr int IPB = 4 * (64 / GL);
    const int chunks = (W.n_items + IPB - 1) / IPB;
    for (int sl = 0; sl < ns; ++sl)
        hipLaunchKernelGGL((k_spmm_sliced<GL, U, true, true, 0, false, true>), dim3(chunks), dim3(256), 0, st, W.n_items, W.e0, W.cnt,
                           W.dst, c.rowptr, c.g->n, c.col, c.val, c.S, c.H, c.H, c.out, c.H, W.partial, c.H, ns, sl);
    if (W.n_long > 0) {
        const int tot = W.n_long * c.H;
        hipLaunchKernelGGL(k_combine, dim3((tot + 255) / 256), dim3(256), 0, st, W.n_long, W.long_row, W.long_segptr, W.partial, c.H,
                           c.H, c.out, c.H);
    }
}

template <int U, int ORDER = 2>
static void run_sliced8(Ctx &c, hipStream_t st) {
    Work &W = ORDER == 2 ? c.work_half : (ORDER ? c.work_col : c.work);
    if (c.H > 256) { fprintf(stderr, "sliced8: H <= 256\n"); exit(1); }
    const int chunks = (W.n_items + 15) / 16;
    hipLaunchKernelGGL((k_spmm_sliced8<U>), dim3(8 * chunks), dim3(256), 0, st, W.n_items, W.e0, W.cnt, W.dst, c.g->n, c.col, c.val,
                       c.S, c.H, c.H, c.out, c.H, W.partial, c.H);
    if (W.n_long > 0) {
        const int tot = W.n_long * c.H;
        hipLaunchKernelGGL(k_combine, dim3((tot + 255) / 256), dim3(256), 0, st, W.n_long, W.long_row, W.long_segptr, W.partial, c.H,
                           c.H, c.out, c.H);
    }
}
template <int U, int ORDER = 2>
static void run_ceiling8(Ctx &c, hipStream_t st) {
    Work &W = ORDER == 2 ? c.work_half : (ORDER ? c.work_col : c.work);
    const int chunks = (W.n_items + 15) / 16;
    hipLaunchKernelGGL((k_gather_ceiling8<U>), dim3(8 * chunks), dim3(256), 0, st, W.n_items, W.e0, W.cnt, c.col, c.S, c.H, c.H, c.sink);
}

static void run_lib(Ctx &c, hipStream_t st) {
    if (lt_spmm_csr_f32(c.lg, c.S, c.H, c.H, nullptr, 0, c.out, c.H, st) != LT_OK) {
        fprintf(stderr, "lt_spmm_csr_f32: %s\n", lt_last_error());
        exit(1);
    }
}

static void run_lib_rows(Ctx &c, hipStream_t st) { lt_set_tuning("tiled_min_bytes", 1LL << 60); run_lib(c, st); lt_set_tuning("tiled_min_bytes", LT_TUNING_DEFAULT); }
static void run_lib_tiled(Ctx &c, hipStream_t st) { lt_set_tuning("tiled_min_bytes", 0); run_lib(c, st); lt_set_tuning("tiled_min_bytes", LT_TUNING_DEFAULT); }
struct Variant { const char *name; void (*fn)(Ctx &, hipStream_t); };

int main(int argc, char **argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 16;
    const int ef = argc > 2 ? atoi(argv[2]) : 16;
    const int H = argc > 3 ? atoi(argv[3]) : 256;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    const char *only = argc > 5 ? argv[5] : "";
    auto t0 = std::chrono::steady_clock::now();
    // LAB_BANDED=<window>: a graph with locality instead of R-MAT (edge factor = average degree / 2)
    const int banded = getenv("LAB_BANDED") ? atoi(getenv("LAB_BANDED")) : 0;
    // LAB_DRAWS=<n>: directed R-MAT draws (BASELINE configs[4]: 40000000 at scale 21) instead of edge_factor << scale
    const long draws = getenv("LAB_DRAWS") ? atol(getenv("LAB_DRAWS")) : ((long)ef << scale);
    HostCsr g = banded ? make_banded(scale, 2 * ef, banded, 42) : make_rmat(scale, draws, 42);
    const long nnz = (long)g.col.size();
    int maxd = 0;
    for (int r = 0; r < g.n; ++r) maxd = std::max(maxd, g.rowptr[r + 1] - g.rowptr[r]);
    printf("rmat scale %d: n %d nnz %ld max row %d (host build %.1f s)\n", scale, g.n, nnz, maxd,
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    Ctx c;
    c.g = &g; c.H = H;
    c.rowptr = upload(g.rowptr); c.col = upload(g.col, 64); c.val = upload(g.val, 64);
    {
        std::vector<float> s((size_t)g.n * H);
        std::mt19937 r2(7);
        std::normal_distribution<float> N(0.f, 1.f);
        for (auto &x : s) x = N(r2);
        c.S = upload(s);
    }
    CK(hipMalloc((void **)&c.out, (size_t)g.n * H * sizeof(float)));
    CK(hipMalloc((void **)&c.ref, (size_t)g.n * H * sizeof(float)));
    c.work = build_work(g, H, 128, 0);
    c.work_col = build_work(g, H, 128, 1);
    CK(hipMalloc((void **)&c.sink, ((size_t)g.col.size() / 8 + (size_t)g.n + 1024) * 8 * sizeof(unsigned)));
    c.work_o3 = build_work(g, H, 128, 3);
    c.work_o4 = build_work(g, H, 128, 4);
    c.work_half = build_work(g, H, 128, 2);   // the library's canonical order: 128-entry segments added in order (lt_rows.hip.h row_dot)
    {
        // hot sets by in-degree (= row length: the matrix is symmetric)
        std::vector<int> deg((size_t)g.n);
        for (int r = 0; r < g.n; ++r) deg[r] = g.rowptr[r + 1] - g.rowptr[r];
        std::vector<int> sorted(deg);
        std::sort(sorted.begin(), sorted.end(), std::greater<int>());
        // what an LDS-resident hot block could hold (160 KB = 640 pieces of 256 B / 160 whole rows) up to the Infinity Cache
        for (int k : {160, 640, 2560, 32768, 65536, 131072, 262144, 524288}) {
            if (k > g.n) break;
            const int thr = sorted[k - 1];
            long hot = 0;
            for (size_t e = 0; e < g.col.size(); ++e) hot += deg[g.col[e]] >= thr;
            printf("top %7d columns (degree >= %d): %.1f%% of the entries\n", k, thr, 100.0 * hot / g.col.size());
        }
        const int hs[4] = {8192, 12288, 16384, 24576};
        for (int h = 0; h < 4; ++h) {
            const int thr = sorted[std::min(hs[h], g.n) - 1];   // columns with deg > thr are hot (at most hs[h] of them)
            std::vector<int32_t> ch(g.col.size());
            long hot = 0;
            for (size_t e = 0; e < ch.size(); ++e) {
                const bool cold = deg[g.col[e]] <= thr;
                hot += !cold;
                ch[e] = g.col[e] | (cold ? (int32_t)0x80000000 : 0);
            }
            c.colh[h] = upload(ch, 64);
            printf("hot set %d: degree > %d, %.1f%% of the entries\n", hs[h], thr, 100.0 * hot / ch.size());
        }
    }
    if (lt_graph_create(g.n, nnz, g.rowptr.data(), g.col.data(), g.val.data(), &c.lg) != LT_OK) {
        fprintf(stderr, "lt_graph_create: %s\n", lt_last_error());
        return 1;
    }
    printf("work items %d (long rows %d, segments %d)\n", c.work.n_items, c.work.n_long, c.work.n_seg);
    const double alg = (double)nnz * 8 + ((double)g.n + 1) * 4 + 2.0 * g.n * H * 4;
    const double gather = (double)nnz * H * 4;

    std::vector<Variant> vs = {
        {"lib", run_lib},
        {"lib_rows", run_lib_rows},
        {"lib_tiled", run_lib_tiled},
        {"g16_col", run_sliced<16, 8, true, true, 0, false, 0, 1>},
        {"g16_col_pf", run_sliced<16, 8, true, true, 0, false, 0, 1, true>},
        {"g16_half", run_sliced<16, 8, true, true, 0, false, 0, 2>},
        {"g16_half_pf", run_sliced<16, 8, true, true, 0, false, 0, 2, true>},
        {"g32_col", run_sliced<32, 8, true, true, 0, false, 0, 1>},
        {"g32_col_pf", run_sliced<32, 8, true, true, 0, false, 0, 1, true>},
        {"g16_col_u4_pf", run_sliced<16, 4, true, true, 0, false, 0, 1, true>},
        {"g16_half_u16_pf", run_sliced<16, 16, true, true, 0, false, 0, 2, true>},
        {"g8_half_pf", run_sliced<8, 8, true, true, 0, false, 0, 2, true>},
        {"g8_col_pf", run_sliced<8, 8, true, true, 0, false, 0, 1, true>},
        {"g32_half_pf", run_sliced<32, 8, true, true, 0, false, 0, 2, true>},
        {"seq4_half_pf", run_sliced<16, 8, true, true, 0, true, 0, 2, true>},      // one launch, blocks in slice order (4 x 256 B)
        {"seq8_half_pf", run_sliced<8, 8, true, true, 0, true, 0, 2, true>},       // (8 x 128 B)
        {"t4_half_pf", run_temporal<16, 8, 2>},                                    // one launch per slice: 4 passes of 256 B
        {"t8_half_pf", run_temporal<8, 8, 2>},                                     // 8 passes of 128 B
        {"t2_half_pf", run_temporal<32, 8, 2>},                                    // 2 passes of 512 B
        {"w8_half_u8", run_sliced8<8, 2>},
        {"w8_half_u16", run_sliced8<16, 2>},
        {"w8_col_u8", run_sliced8<8, 1>},
        {"ceilw8_u8", run_ceiling8<8>},
        {"ceilw8_u16", run_ceiling8<16>},
        {"ceil16_u8", run_ceiling<16, 8>},
        {"ceil16_u16", run_ceiling<16, 16>},
        {"ceil8_u8", run_ceiling<8, 8>},
        {"ceil32_u8", run_ceiling<32, 8>},
        {"ceil32_u16", run_ceiling<32, 16>},
        {"ceil64_u8", run_ceiling<64, 8>},
        {"ceil16_u8_len", run_ceiling<16, 8, 0>},
        {"ceil16_u8_col", run_ceiling<16, 8, 1>},
        {"ceil16_u8_o3", run_ceiling<16, 8, 3>},
        {"ceil16_u8_o4", run_ceiling<16, 8, 4>},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    bool have_ref = false;
    std::vector<float> h_ref, h_out;
    for (auto &v : vs) {
        if (only[0] && strcmp(v.name, "lib") != 0) {
            // comma-separated tokens; a token selects the variants whose name it is a prefix of
            bool sel = false;
            std::string o(only);
            for (size_t b = 0; b < o.size();) {
                size_t e = o.find(',', b);
                if (e == std::string::npos) e = o.size();
                if (e > b && strncmp(v.name, o.c_str() + b, e - b) == 0) sel = true;
                b = e + 1;
            }
            if (!sel) continue;
        }
        CK(hipMemsetAsync(c.out, 0xff, (size_t)g.n * H * sizeof(float), st));
        v.fn(c, st);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        // correctness: bit-equal to the library kernel
        const char *verdict = "ref";
        if (!strncmp(v.name, "ceil", 4)) {      // (also ceilw8_*)
            verdict = "(gathers only)";
        } else if (!have_ref) {
            CK(hipMemcpy(c.ref, c.out, (size_t)g.n * H * sizeof(float), hipMemcpyDeviceToDevice));
            have_ref = true;
        } else {
            h_ref.resize((size_t)g.n * H); h_out.resize((size_t)g.n * H);
            CK(hipMemcpy(h_ref.data(), c.ref, h_ref.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h_out.data(), c.out, h_out.size() * 4, hipMemcpyDeviceToHost));
            // rows of up to 512 entries: the library sums them in the same order -> bit-equal; longer rows: the
            // library nests 512-entry segments around the 128-entry ones, so only closeness is checked there
            size_t bad = 0, loose = 0;
            double worst = 0.0;
            for (int r = 0; r < g.n; ++r) {
                const bool exact = g.rowptr[r + 1] - g.rowptr[r] <= 512;
                for (int k = 0; k < H; ++k) {
                    const size_t i = (size_t)r * H + k;
                    if (exact) bad += memcmp(&h_ref[i], &h_out[i], 4) != 0;
                    else {
                        const double d = fabs((double)h_ref[i] - h_out[i]) / (1.0 + fabs((double)h_ref[i]));
                        if (!(d <= 1e-4)) ++loose;
                        worst = std::max(worst, d);
                    }
                }
            }
            static char buf[96];
            snprintf(buf, sizeof buf, "%s (short rows: %zu words differ; long rows: worst rel %.1e)",
                     bad || loose ? "MISMATCH" : "ok", bad, worst);
            verdict = buf;
        }
        v.fn(c, st);
        CK(hipEventRecord(ea, st));
        for (int i = 0; i < reps; ++i) v.fn(c, st);
        CK(hipEventRecord(eb, st));
        CK(hipEventSynchronize(eb));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, ea, eb));
        ms /= reps;
        printf("%-18s %9.4f ms   algorithmic %8.1f GB/s (%.1f%% of 8 TB/s)   gather %6.2f TB/s   %s\n", v.name, ms,
               alg / ms / 1e6, alg / ms / 1e6 / 80.0, gather / ms / 1e9, verdict);
        fflush(stdout);
    }
    return 0;
}
