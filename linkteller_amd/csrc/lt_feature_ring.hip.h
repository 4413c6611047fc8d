// The feature-difference product rows as a PERSISTENT kernel with an LDS ring (round 6) -- included by lt_fp64.hip.
//
// k_s1d_feature_rows (lt_fp64.hip) gives every row of X its own wave and issues all of X's loads at once: the rows of a CU
// then arrive together, and the compare steps, the list walk and the store of ALL rows queue behind the last byte -- the phases
// add (12 us of reads become 23, profiles/r04_feat_lab_timeline.txt).  Here ONE workgroup of FR_WAVES waves per CU owns a
// contiguous range of rows and every wave keeps exactly one row in flight AHEAD of the row it works on:
//   * a row travels global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR destination) into
//     the wave's private ring of `nch` slots (nch = the 1-KiB chunks of a row); the chunk of the NEXT row is issued into a slot
//     the moment the chunk of this row has been compared out of it, so the CU's memory pipe never runs dry while a wave
//     compares, walks W1 and stores;
//   * the wait in front of a chunk is a COUNTED s_waitcnt vmcnt(nch - 1): the oldest DMA has landed, the others stay in flight;
//   * the waves of a workgroup claim rows from an LDS counter (a row is ~2 us of work: static ranges would leave 7 waves idle
//     behind the one with a row more);
//   * rows are only 8-byte aligned (F = 3170): a chunk starts at the row's 16-byte floor, `shift` (0 or 2 floats) is the
//     row's offset inside it, the reference vector sits in LDS two floats in so that either shift reads it 8-byte aligned.
// The list of differing columns is walked as in the row-per-wave kernel (FR_P W1 rows in flight: one trip for the usual
// row); the order of a list is (chunk, float of the lane's four, lane) -- a fixed order of the data alone, so a row's fp64
// sum is reproducible, but NOT the order of k_s1d_feature_rows (chunk of 128, float of two, lane): the two kernels agree to
// fp64 rounding (1e-16), not bit for bit (the same rule as the three routes of DESIGN 5.3).
// cref = m W1 (deferred form): the first `nsl` workgroups each sum a K range of m W1 before their rows (8 waves x one trip;
// their row ranges are shorter by the same bytes), the workgroup whose slab arrives last adds the slabs -- the hand-off of
// k_s1d_feature_rows (sc1 stores, per-wave vmcnt(0), workgroup barrier, agent-scope ticket; the argument is at that code site).
#define FR_WAVES 8
#define FR_CAP 256                 // list entries a wave's LDS holds
#define FR_P 24                    // W1 rows in flight per lane while a list is walked
#define FR_USE (FR_CAP - FR_P)     // rows with more differing columns take the piecewise path (the walk pads a list to a batch)
#define FR_SLABS 32                // workgroups that carry a K range of m W1
#define FR_LDS_MAX (160 * 1024)

static inline size_t fr_smem_bytes(int nch) {
    return ((size_t)nch * 1024 + 16) + (size_t)FR_WAVES * nch * 1024 + (size_t)FR_WAVES * FR_CAP * (sizeof(double) + sizeof(int)) + 16;
}
static inline int fr_chunks(int F) { return (F + 2 + 255) / 256; }

template <int N> __device__ __forceinline__ void fr_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wave-uniform n: all but the n youngest vector-memory operations of this wave are done
__device__ __forceinline__ void fr_wait_dyn(int n) {
    switch (n) {
    case 0: fr_wait<0>(); break;   case 1: fr_wait<1>(); break;   case 2: fr_wait<2>(); break;   case 3: fr_wait<3>(); break;
    case 4: fr_wait<4>(); break;   case 5: fr_wait<5>(); break;   case 6: fr_wait<6>(); break;   case 7: fr_wait<7>(); break;
    case 8: fr_wait<8>(); break;   case 9: fr_wait<9>(); break;   case 10: fr_wait<10>(); break; case 11: fr_wait<11>(); break;
    case 12: fr_wait<12>(); break; case 13: fr_wait<13>(); break; case 14: fr_wait<14>(); break; case 15: fr_wait<15>(); break;
    default: fr_wait<0>(); break;
    }
}

typedef __attribute__((address_space(3))) void *fr_lds_ptr_t;

// max over the wave's 64 lanes (DPP inside a row of 16, the four rows by v_readlane): ~12 VALU instead of six ds_bpermute round trips
__device__ __forceinline__ unsigned fr_wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));     // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));     // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
// the largest of 64 non-negative finite doubles, exactly (their bit patterns order as integers: high words first, then the low
// words of the lanes that hold the largest high word)
__device__ __forceinline__ double fr_wave_max_nonneg(double v) {
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const unsigned mh = fr_wave_max_u32(hi);
    const unsigned ml = fr_wave_max_u32(hi == mh ? lo : 0u);
    return __hiloint2double((int)mh, (int)ml);
}

#ifdef LT_FR_TRACE      // tools/read_lab/ring_lab.hip: stamps of every wave's rows on the constant 100 MHz clock
__device__ unsigned long long *g_fr_trace = nullptr;
#define FR_STAMP(k_)                                                                                                   \
    do {                                                                                                               \
        if (lane == 0 && g_fr_trace && fr_slot < 4)                                                                    \
            g_fr_trace[(((size_t)blockIdx.x * FR_WAVES + wid) * 4 + fr_slot) * 8 + (k_)] = wall_clock64();             \
    } while (0)
#else
#define FR_STAMP(k_)
#endif

// NCHT: the chunks of a row at compile time (the counted wait is an immediate), 0 = any (a branch per wait)
template <int NCHT>
__global__ __launch_bounds__(64 * FR_WAVES) void k_s1d_feature_ring(
    int n, int F, int H, const float *__restrict__ X, long ldx, const float *__restrict__ ref, const float *__restrict__ W1,
    const double *__restrict__ cref, double *__restrict__ S1d, int hint_cap, int *__restrict__ dense_hint, int nsl,
    double *__restrict__ slabs, int32_t *__restrict__ zstate, float *__restrict__ S1x, unsigned *__restrict__ gate,
    double *__restrict__ cref_out, double *__restrict__ S1qs, int nch_arg, int w_all, int w_cut) {
    const int nch = NCHT ? NCHT : nch_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char fr_smem[];
    float *sref = reinterpret_cast<float *>(fr_smem);                                   // [nch * 256 + 4]: sref[j + 2] = m[j]
    float *ring_all = reinterpret_cast<float *>(fr_smem + (size_t)nch * 1024 + 16);    // [FR_WAVES][nch][256]
    double *mv_all = reinterpret_cast<double *>(ring_all + (size_t)FR_WAVES * nch * 256);   // [FR_WAVES][FR_CAP]
    int *mj_all = reinterpret_cast<int *>(mv_all + FR_WAVES * FR_CAP);                  // [FR_WAVES][FR_CAP]
    int *s_misc = mj_all + FR_WAVES * FR_CAP;                                           // [0] the row counter, [1] "last slab"
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x, G = gridDim.x;
    // row ranges: a workgroup's weight is w_all, less w_cut for the ones that carry a slab (same bytes through every CU)
    const long wtot = (long)G * w_all - (long)nsl * w_cut;
    auto start_of = [&](int b) { return (int)(((long)n * ((long)b * w_all - (long)min(b, nsl) * w_cut)) / wtot); };
    const int rb0 = start_of(blk), rb1 = blk + 1 == G ? n : start_of(blk + 1);
    float *ring = ring_all + (size_t)wid * nch * 256;
    double *mv = mv_all + wid * FR_CAP;
    int *mj = mj_all + wid * FR_CAP;
    const char *x_end = reinterpret_cast<const char *>(X + (long)(n - 1) * ldx + F);
    const unsigned lane_off = 16u * (unsigned)lane;

    // chunk u of `row` into slot u of this wave's ring
    auto issue = [&](int row, int u) {
        const char *rp = reinterpret_cast<const char *>(X + (long)row * ldx);
        const char *cb = rp - (reinterpret_cast<uintptr_t>(rp) & 15) + (size_t)u * 1024;
        if (row == n - 1) {
            // the last row's chunks may reach past the matrix: a lane whose 16 bytes lie wholly behind it reads the chunk's first
            // 16 instead (its columns are >= F: masked when compared); a window that only straddles the end stays inside its own
            // 16-byte unit (same page)
            const unsigned off = (cb + lane_off < x_end) ? lane_off : 0u;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(cb + off), (fr_lds_ptr_t)(ring + u * 256), 16, 0, 0);
        } else {
            // uniform chunk base + 32-bit lane offset: the saddr form of global_load_lds, no VALU (the two empty asm statements keep
            // hipcc from re-associating the address into a 64-bit VALU add per chunk: k_full_stageA_lds)
            unsigned off = lane_off;
            asm("" : "+s"(cb));
            asm("" : "+v"(off));
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(cb + off), (fr_lds_ptr_t)(ring + u * 256), 16, 0, 0);
        }
    };
    int cur = rb0 + wid < rb1 ? rb0 + wid : -1;          // wave-uniform: the row being worked on; the first one is static
    if (cur >= 0)
        for (int u = 0; u < nch; ++u) issue(cur, u);
    // ---- the reference vector into LDS (behind the first row's DMAs: one round trip covers both).  One pass: nch <= 15 is
    // at most 3844 floats for 8 x 512 slots; `ref` is allocated FD_REF_PAD floats past F, so no upper clamp.
    {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = ref[max(u * 64 * FR_WAVES + tid - 2, 0)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int js = u * 64 * FR_WAVES + tid;
            if (js < nch * 256 + 4) sref[js] = (js >= 2 && js - 2 < F) ? r[u] : 0.f;
        }
    }
    if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
    // ---- this workgroup's K range of m W1 (deferred cref)
    if (blk < nsl) {
        const int kper = (F + nsl - 1) / nsl;                        // k's per slab
        const int kw = (kper + FR_WAVES - 1) / FR_WAVES;             // ... per wave
        const int kb1 = min(F, (blk + 1) * kper);
        const int k0 = blk * kper + wid * kw, k1 = min(kb1, k0 + kw);
        const int c0 = 4 * lane;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (c0 < H) {
            for (int kk = k0; kk < k1; kk += 16) {
                f32x4 w[16];
                float m[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = min(kk + u, F - 1);
                    m[u] = kk + u < k1 ? ref[k] : 0.f;
                    w[u] = ld4(W1 + (size_t)k * H + c0);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[t] = fma((double)m[u], (double)w[u][t], a[t]);
            }
        }
        double *part = mv_all;                                        // [FR_WAVES][256] (the lists are not in use yet)
#pragma unroll
        for (int t = 0; t < 4; ++t) part[wid * 256 + ((c0 + t) & 255)] = a[t];
        __syncthreads();
        if (wid == 0) {
            // the eight waves' partial sums in wave order, then out as device-scope stores (they bypass this XCD's L2)
            if (c0 < H) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double s = part[c0 + t];
#pragma unroll
                    for (int w = 1; w < FR_WAVES; ++w) s += part[w * 256 + c0 + t];
                    __hip_atomic_store(slabs + (size_t)blk * 256 + c0 + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "this hand-off is written against gfx950's memory system (see k_s1d_feature_rows)"
#endif
            // (memory-order argument: k_s1d_feature_rows' -- sc1 stores drained by the storing wave, the ticket's returned value
            // names the last workgroup, whose sc1 loads come after the barrier its wave 0 joins once the add has returned)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                s_misc[1] = __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsl - 1u ? 1 : 0;
        }
    }
    __syncthreads();
    if (blk < nsl && s_misc[1]) {
        // the last slab's workgroup: wave z adds slabs z, z + 8, ... in order, then the eight sums in wave order
        const int c0 = 4 * lane;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (c0 < H) {
            double t[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (no branch between the loads: a slab past the last is read as the last and not added)
                    const int z = min(wid + FR_WAVES * u, nsl - 1);
                    t[u][q] = __hip_atomic_load(slabs + (size_t)z * 256 + c0 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += wid + FR_WAVES * u < nsl ? t[u][q] : 0.0;
        }
        double *part = mv_all;
        __syncthreads();                                              // (wave 0 has read the partial sums of the slab itself)
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wid * 256 + ((c0 + q) & 255)] = acc[q];
        __syncthreads();
        if (tid < 256) {
            double s = part[tid];
#pragma unroll
            for (int w = 1; w < FR_WAVES; ++w) s += part[w * 256 + tid];
            if (tid < H) cref_out[tid] = s;
        }
        if (tid == 0) __hip_atomic_store(gate, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        __syncthreads();                                              // (the lists start over the partial sums)
    }
    if (cur < 0) return;
    const int c0 = 4 * lane;
    const bool own = c0 < H;                                          // (H % 4 == 0: a lane holds four hidden columns or none)
    const unsigned long long lt = (1ull << lane) - 1ull;
    f64x4 crefv = {0.0, 0.0, 0.0, 0.0};
    if (cref && own) crefv = *reinterpret_cast<const f64x4 *>(cref + c0);
    auto claim = [&]() {
        int r = 0;
        if (lane == 0) r = atomicAdd(&s_misc[0], 1);
        r = __builtin_amdgcn_readfirstlane(r);
        const int row = rb0 + FR_WAVES + r;
        return row < rb1 ? row : -1;
    };
    int nxt = claim();
    [[maybe_unused]] int fr_slot = 0;
    // fixed-point words / fp64 values of a finished row, and the store of them
    auto finish_row = [&](const double (&acc)[4], int (&q)[4], double &scale, f64x4 &o) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = own ? (cref ? crefv[t] + acc[t] : acc[t]) : 0.0;
        if (S1x) {      // 32-bit fixed point with one scale per row (k_s1d_feature_rows: the same words for the same fp64 values)
            const double mx = fr_wave_max_nonneg(fmax(fmax(fabs(o[0]), fabs(o[1])), fmax(fabs(o[2]), fabs(o[3]))));
            scale = mx > 0.0 ? mx * (1.0 / 2147483000.0) : 1.0;
            const double inv = 1.0 / scale;
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = (int)rint(o[t] * inv);
        }
    };
    auto store_row = [&](int row, const int (&q)[4], double scale, const f64x4 &o) {
        if (S1x) {
            if (own) *reinterpret_cast<int4 *>(S1x + (size_t)row * H + c0) = make_int4(q[0], q[1], q[2], q[3]);
            if (lane == 0) S1qs[row] = scale;
        } else if (own) {
            *reinterpret_cast<f64x4 *>(S1d + (size_t)row * H + c0) = o;
        }
        if (zstate && lane == 0) zstate[row] = 0;
    };
    if constexpr (NCHT > 0) {
        // ---- whole rows at a time.  What a wave does per row is a chain of dependent instructions with one other wave on its SIMD to
        // hide behind (a dependent VALU -> SALU -> branch hop is ~10-20 cycles here: the row-per-wave kernel's 52 ballot steps cost
        // 3.6 us of a wave's row, tools/read_lab/ring_lab timeline), so the row is made of FEW instructions:
        //   1. the row's chunks and the reference values go to registers in one burst of LDS reads; one bit per value says
        //      whether it differs (xor, min, shift-or: plain VALU, four independent accumulators);
        //   2. the lane's three lowest flagged values are read again from LDS in ONE trip (a register array cannot be indexed
        //      by a lane's own bit number) and appended level by level -- order (level, lane), a function of the data alone;
        //      lanes with more take a loop;
        //   3. the W1 rows of the list are asked for (inline asm, no padding), the next row's DMAs go out BEHIND them -- loads
        //      return in order, so the walk's hand-counted wait leaves those NCHT DMAs in flight -- then the fp64 sums;
        //   4. the finished words are stored one row later: the drain at the top of a row never waits for a store.
        int prow = -1, pq[4] = {0, 0, 0, 0};
        double pscale = 1.0;
        f64x4 po = {0.0, 0.0, 0.0, 0.0};
        while (cur >= 0) {
            const char *rp = reinterpret_cast<const char *>(X + (long)cur * ldx);
            const int shift = (int)((reinterpret_cast<uintptr_t>(rp) & 15) >> 2);
            FR_STAMP(0);
            fr_wait<0>();                                             // the row has landed (nothing younger than its DMAs is in flight)
            FR_STAMP(1);
            const int jb0 = 4 * lane - shift;                         // column of the lane's first value of chunk 0
            unsigned fl[4] = {0u, 0u, 0u, 0u};                        // bit (4 u + v) & 31 of fl[u / 8 * 2 + (v & 1)... ] -- see `word`
            {
                f32x4 xv[NCHT];
                f32x2_ r01[NCHT], r23[NCHT];
#pragma unroll
                for (int u = 0; u < NCHT; ++u) {
                    xv[u] = *reinterpret_cast<const f32x4 *>(ring + u * 256 + 4 * lane);
                    r01[u] = *reinterpret_cast<const f32x2_ *>(sref + u * 256 + jb0 + 2);
                    r23[u] = *reinterpret_cast<const f32x2_ *>(sref + u * 256 + jb0 + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NCHT; ++u) {
                    const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                    const float rr[4] = {r01[u].x, r01[u].y, r23[u].x, r23[u].y};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        // (bitwise: +0 against -0 is listed with a difference of exactly zero; NaNs are refused at baseline creation)
                        unsigned t = min(__float_as_uint(xs[v]) ^ __float_as_uint(rr[v]), 1u);
                        // chunks 1 .. NCHT - 3 lie inside the row for every F with NCHT chunks; the others are masked by column
                        if (u == 0 || u >= NCHT - 2) t = (unsigned)(u * 256 + jb0 + v) < (unsigned)F ? t : 0u;
                        const int bit = 4 * u + v;                    // flags word bit / 32, two accumulators per word (v & 1)
                        fl[(bit >> 5) * 2 + (v & 1)] |= t << (bit & 31);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long flags = ((unsigned long long)(fl[2] | fl[3]) << 32) | (unsigned long long)(fl[0] | fl[1]);
            // levels 0 .. 2 in one LDS trip
            int total = 0;
            {
                int jq[3];
                float xq[3], rq[3];
                bool has[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    has[t] = flags != 0ull;
                    const int bit = has[t] ? __ffsll((long long)flags) - 1 : 0;
                    flags &= flags - 1ull;                            // (0 stays 0)
                    const int u = bit >> 2, v = bit & 3;
                    jq[t] = u * 256 + jb0 + v;
                    xq[t] = ring[u * 256 + 4 * lane + v];
                    rq[t] = sref[max(jq[t], 0) + 2];
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const unsigned long long m = __ballot(has[t]);
                    const int pos = total + __popcll(m & lt);
                    if (has[t] && pos < FR_USE) { mj[pos] = jq[t]; mv[pos] = (double)xq[t] - (double)rq[t]; }
                    total += __popcll(m);
                }
            }
            while (__ballot(flags != 0ull)) {                         // lanes with more than three (a handful of rows)
                const bool has = flags != 0ull;
                const unsigned long long m = __ballot(has);
                if (has) {
                    const int bit = __ffsll((long long)flags) - 1;
                    flags &= flags - 1ull;
                    const int u = bit >> 2, v = bit & 3, j = u * 256 + jb0 + v;
                    const float xq = ring[u * 256 + 4 * lane + v], rq = sref[j + 2];
                    const int pos = total + __popcll(m & lt);
                    if (pos < FR_USE) { mj[pos] = j; mv[pos] = (double)xq - (double)rq; }
                }
                total += __popcll(m);
            }
            if (prow >= 0) store_row(prow, pq, pscale, po);           // (a row late: see 4.)
            FR_STAMP(2);
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            const unsigned c0b = own ? 16u * (unsigned)lane : 0u;     // byte offset of the lane's columns in a W1 row
            // entries [e, e + cnt), cnt <= FR_P: their W1 rows asked for (inline asm: with an LDS-DMA in flight hipcc (ROCm 7.2) puts
            // s_waitcnt vmcnt(0) in front of the first use of any ordinary load's result -- that would drain the DMAs issued just
            // behind them), (dma) the next row's DMAs, the hand-counted wait -- exactly the NCHT DMAs are younger than the last
            // load; "memory" clobbers keep every other memory operation outside, an extra one would only make the wait stricter --
            // then the sums in list order
            auto batch = [&](int e, int cnt, bool dma) {
                f32x4 w[FR_P];
#pragma unroll
                for (int k = 0; k < FR_P; ++k) {
                    if (k < cnt) {
                        const char *p = reinterpret_cast<const char *>(W1) + (size_t)(unsigned)mj[e + k] * (unsigned)(4 * H) + c0b;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[k]) : "v"(p) : "memory");
                    }
                }
                if (dma) {
#pragma unroll
                    for (int u = 0; u < NCHT; ++u) issue(nxt, u);
                    fr_wait<NCHT>();
                } else {
                    fr_wait<0>();
                }
#pragma unroll
                for (int k = 0; k < FR_P; ++k) {
                    if (k < cnt) {
                        asm volatile("" : "+v"(w[k]));               // (the value exists from here on)
                        const double d = mv[e + k];
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = fma(d, (double)w[k][t], acc[t]);
                    }
                }
            };
            auto walk = [&](int cnt, bool first) {
                first = first && nxt >= 0;                            // (whether the next row's DMAs are still to be issued)
                for (int e = 0; e < cnt; e += FR_P) { batch(e, min(cnt - e, FR_P), first); first = false; }
                if (first) {                                          // (an empty list)
#pragma unroll
                    for (int u = 0; u < NCHT; ++u) issue(nxt, u);
                }
            };
            if (total <= FR_USE) {
                walk(total, true);
            } else {
                // a dense row (see the chunk-wise form below)
                if (nxt >= 0) {
#pragma unroll
                    for (int u = 0; u < NCHT; ++u) issue(nxt, u);
                }
                const float *xr = X + (long)cur * ldx;
                for (int j0 = 0; j0 < F; j0 += 64) {
                    const int j = j0 + lane;
                    const float xs = j < F ? xr[j] : 0.f;
                    const float r = j < F ? sref[j + 2] : 0.f;
                    const bool diff = j < F && xs != r;
                    const unsigned long long m = __ballot(diff);
                    if (diff) { const int pos = __popcll(m & lt); mj[pos] = j; mv[pos] = (double)xs - (double)r; }
                    walk(__popcll(m), false);
                }
            }
            FR_STAMP(3);
            if (total > hint_cap && lane == 0) *dense_hint = 1;
            finish_row(acc, pq, pscale, po);
            prow = cur;
            FR_STAMP(4);
            ++fr_slot;
            cur = nxt;
            if (cur >= 0) nxt = claim();
        }
        if (prow >= 0) store_row(prow, pq, pscale, po);
        return;
    }
    while (cur >= 0) {
        const char *rp = reinterpret_cast<const char *>(X + (long)cur * ldx);
        const int shift = (int)((reinterpret_cast<uintptr_t>(rp) & 15) >> 2);         // floats between the chunk's start and the row's
        int total = 0;                                                // wave-uniform: differing columns of the row
        if (nxt < 0) fr_wait<0>();                                    // the wave's last row: nothing is issued behind its chunks
#pragma unroll 1
        for (int u = 0; u < nch; ++u) {
            if (nxt >= 0) {                                           // chunk u has landed; the nch - 1 younger DMAs stay in flight
                if constexpr (NCHT > 0) fr_wait<NCHT - 1>();
                else fr_wait_dyn(nch - 1);
            }
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(ring + u * 256 + 4 * lane);
            const int jb = u * 256 + 4 * lane - shift;                // column of xv[0]
            const f32x2_ r01 = *reinterpret_cast<const f32x2_ *>(sref + jb + 2);
            const f32x2_ r23 = *reinterpret_cast<const f32x2_ *>(sref + jb + 4);
            const float rr[4] = {r01.x, r01.y, r23.x, r23.y};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool diff = (unsigned)(jb + v) < (unsigned)F && xv[v] != rr[v];
                const unsigned long long m = __ballot(diff);
                if (m) {
                    const int pos = total + __popcll(m & lt);
                    if (diff && pos < FR_USE) { mj[pos] = jb + v; mv[pos] = (double)xv[v] - (double)rr[v]; }
                    total += __popcll(m);
                }
            }
            // the slot is free (its values are in registers, compared): the next row's chunk goes into it
            asm volatile("" ::: "memory");
            if (nxt >= 0) issue(nxt, u);
        }
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        // entries [e, e + NB) of the list, all NB W1 rows in one trip
        auto batch = [&](int e, auto nb_tag) {
            constexpr int NB = decltype(nb_tag)::value;
            f32x4 w[NB];
            double d[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                d[k] = mv[e + k];
                w[k] = ld4(W1 + (size_t)mj[e + k] * H + c0);
            }
            __builtin_amdgcn_sched_barrier(0);      // (every load of the batch goes out before the first wait: one trip)
#pragma unroll
            for (int k = 0; k < NB; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = fma(d[k], (double)w[k][t], acc[t]);
        };
        auto walk = [&](int cnt) {
            const int padded = (cnt + 7) & ~7;
            if (lane < padded - cnt) { mj[cnt + lane] = 0; mv[cnt + lane] = 0.0; }      // (d = 0: the term adds exactly nothing)
            if (!own) return;
            int e = 0;
            for (; e + FR_P <= padded; e += FR_P) batch(e, std::integral_constant<int, FR_P>{});
            if (padded - e == 16) batch(e, std::integral_constant<int, 16>{});
            else if (padded - e == 8) batch(e, std::integral_constant<int, 8>{});
        };
        if (total <= FR_USE) {
            walk(total);
        } else {
            // a dense row: read again piece by piece, every piece's list walked before the next is made (slow and correct; the
            // hint below moves the baseline off this route)
            const float *xr = X + (long)cur * ldx;
            for (int j0 = 0; j0 < F; j0 += 64) {
                const int j = j0 + lane;
                const float xs = j < F ? xr[j] : 0.f;
                const float r = j < F ? sref[j + 2] : 0.f;
                const bool diff = j < F && xs != r;
                const unsigned long long m = __ballot(diff);
                if (diff) { const int pos = __popcll(m & lt); mj[pos] = j; mv[pos] = (double)xs - (double)r; }
                walk(__popcll(m));
            }
        }
        if (total > hint_cap && lane == 0) *dense_hint = 1;
        {
            int q[4] = {0, 0, 0, 0};
            double scale = 1.0;
            f64x4 o;
            finish_row(acc, q, scale, o);
            store_row(cur, q, scale, o);
        }
        cur = nxt;
        if (cur >= 0) nxt = claim();
    }
}
