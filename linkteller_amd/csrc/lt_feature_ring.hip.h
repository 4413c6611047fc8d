// The feature-difference product rows as a PERSISTENT kernel with an LDS ring (round 6) -- included by lt_fp64.hip.
//
// k_s1d_feature_rows (lt_fp64.hip) gives every row of X its own wave and issues all of X's loads at once: the rows of a CU
// then arrive together, and the list walks (20 W1 rows per row of X: 1.6 x the bytes of X through the L1s) and the stores of ALL
// rows queue behind the last byte (profiles/r04_feat_lab_timeline.txt).  Here a wave works on one row while the NEXT one is
// landing, row after row:
//   * two workgroups of FR_WAVES waves per CU, all resident from the start; a wave's first row is static (block, wave), every
//     further one is claimed from a counter, asked for BEHIND the next row's DMAs and read a row later (under load a trip to the
//     counter is 3-4 us and loads return in order: asked in front of the walk's loads it was 4-8 us of every row); the blocks
//     that first form a slice of m W1 join the rows afterwards with claimed rows only; launches alternate between two sets
//     of counters and every launch zeroes the set the next one uses (a last-wave reset was one more trip, on one word, in
//     every wave's last row);
//   * a row travels global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR destination) into the
//     wave's private ring of NCHT slots (the 1-KiB chunks of a row);
//   * rows are only 8-byte aligned (F = 3170): a chunk starts at the row's 16-byte floor, `shift` (0 or 2 floats) is the
//     row's offset inside it, the reference vector sits in LDS two floats in so that either shift reads it 8-byte aligned.
// What a wave does per row is a chain of dependent instructions with few other waves on its SIMD to hide behind (a dependent
// VALU -> SALU -> branch hop is 10-20 cycles: the row-per-wave kernel's 52 ballot steps were 3.6 us of a wave's row in this
// structure, tools/read_lab/ring_lab), so the row is made of FEW instructions:
//   1. the row's chunks and the reference values go to registers in one burst of LDS reads; one bit per value says whether it
//      differs (xor, min, shift-or: plain VALU, four independent accumulators);
//   2. the lane's three lowest flagged values are read again from LDS in ONE trip (a register array cannot be indexed by a
//      lane's own bit number) and appended level by level -- order (level, lane), a function of the data alone, so a row's fp64
//      sum is reproducible; it is NOT the order of k_s1d_feature_rows (chunk of 128, float of two, lane): the two kernels agree
//      to fp64 rounding (1e-16), not bit for bit (the same rule as the three routes of DESIGN 5.3); lanes with more loop;
//   3. the W1 rows of the list are asked for (inline asm, no padding), the next row's DMAs go out BEHIND them -- loads return in
//      order, so the walk's hand-counted wait leaves those NCHT DMAs in flight -- then the fp64 sums;
//   4. the finished words are stored one row later: the drain at the top of a row never waits for a store's acknowledgement.
// cref = m W1 (deferred form): the first `nslab` blocks of the launch are k_s1d_feature_rows' slab blocks (fd_slab_block: the
// same bits).
#define FR_WAVES 4
#define FR_CAP 256                 // list entries a wave's LDS holds; rows with more differing columns take the piecewise path
#define FR_P 24                    // W1 rows in flight per lane while a list is walked
#define FR_LDS_MAX (80 * 1024)     // two workgroups per CU
#define FR_NCH_MIN 9
#define FR_NCH_MAX 13
#define FR_GROUPS 32               // row counters, each on a 128-byte line of its own; a block's group starts with its XCD's number: the line stays in one L2
#define FR_GATE_WORDS (64 + 64 * FR_GROUPS)    // gate[0] the slab blocks' ticket, gate[64 + 32 (g + FR_GROUPS p)] group g's row counter of parity p

static inline size_t fr_smem_bytes(int nch) {
    return ((size_t)nch * 1024 + 16) + (size_t)FR_WAVES * nch * 1024 + (size_t)FR_WAVES * FR_CAP * (sizeof(double) + sizeof(int)) + 16;
}
static inline int fr_chunks(int F) { return (F + 2 + 255) / 256; }

template <int N> __device__ __forceinline__ void fr_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((address_space(3))) void *fr_lds_ptr_t;

// (fr_wave_max_u32 / fr_wave_max_nonneg: lt_fp64.hip, in front of k_s1d_feature_rows, which takes its row maximum the same way)

#define FR_SLOTS 6
#ifdef LT_FR_TRACE      // tools/read_lab/ring_lab.hip: stamps of every wave's rows on the constant 100 MHz clock
__device__ unsigned long long *g_fr_trace = nullptr;
#define FR_STAMP(k_)                                                                                                   \
    do {                                                                                                               \
        if (lane == 0 && g_fr_trace && fr_slot < FR_SLOTS)                                                             \
            g_fr_trace[(((size_t)blockIdx.x * FR_WAVES + wid) * FR_SLOTS + fr_slot) * 8 + (k_)] = wall_clock64();           \
    } while (0)
#else
#define FR_STAMP(k_)
#endif

// The rows beyond the static ones are cut into FR_GROUPS pools, a counter each; a block claims from the pool of (its XCD, block / 8 mod 4).
template <int NCHT>
__global__ __launch_bounds__(64 * FR_WAVES) void k_s1d_feature_ring(
    int n, int F, int H, const float *__restrict__ X, long ldx, const float *__restrict__ ref, const float *__restrict__ W1,
    const double *__restrict__ cref, double *__restrict__ S1d, int hint_cap, int *__restrict__ dense_hint, int nslab,
    double *__restrict__ slabs, int32_t *__restrict__ zstate, float *__restrict__ S1x, unsigned *__restrict__ gate,
    double *__restrict__ cref_out, double *__restrict__ S1qs, int parity) {
    static_assert(NCHT >= FR_NCH_MIN && NCHT <= FR_NCH_MAX, "chunks per row");
    extern __shared__ __attribute__((aligned(16))) unsigned char fr_smem[];
    const bool slab = (int)blockIdx.x < nslab;                        // (block-uniform) a slice of m W1 first, then rows like the others
    if (slab) {
        fd_slab_block(fr_smem, nslab, F, H, H, ref, W1, slabs, gate, cref_out);
        __syncthreads();                                              // (its LDS is the reference vector's from here on)
    }
    float *sref = reinterpret_cast<float *>(fr_smem);                                   // [NCHT * 256 + 4]: sref[j + 2] = m[j]
    float *ring_all = reinterpret_cast<float *>(fr_smem + (size_t)NCHT * 1024 + 16);   // [FR_WAVES][NCHT][256]
    double *mv_all = reinterpret_cast<double *>(ring_all + (size_t)FR_WAVES * NCHT * 256);   // [FR_WAVES][FR_CAP]
    int *mj_all = reinterpret_cast<int *>(mv_all + FR_WAVES * FR_CAP);                  // [FR_WAVES][FR_CAP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nstatic = ((int)gridDim.x - nslab) * FR_WAVES;          // the static rows are [0, nstatic): one per wave of the other blocks
    float *ring = ring_all + (size_t)wid * NCHT * 256;
    double *mv = mv_all + wid * FR_CAP;
    int *mj = mj_all + wid * FR_CAP;
    const char *x_end = reinterpret_cast<const char *>(X + (long)(n - 1) * ldx + F);
    const unsigned lane_off = 16u * (unsigned)lane;
    [[maybe_unused]] int fr_slot = 0;

    // the NCHT chunks of `row` into the slots of this wave's ring
    auto issue_row = [&](int row) __attribute__((always_inline)) {
        const uintptr_t rp = reinterpret_cast<uintptr_t>(X + (long)row * ldx);
        // (uniform already; the two readfirstlanes say so to hipcc: the chunk bases then live in SGPRs)
        const uintptr_t base = ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rp >> 32)) << 32) |
                               (uintptr_t)((unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rp) & ~15u);
        if (row == n - 1) {
            // the last row's chunks may reach past the matrix: a lane whose 16 bytes lie wholly behind it reads the chunk's first
            // 16 instead (its columns are >= F: masked when compared); a window that only straddles the end stays inside its own
            // 16-byte unit (same page)
#pragma unroll
            for (int u = 0; u < NCHT; ++u) {
                const char *cb = reinterpret_cast<const char *>(base) + (size_t)u * 1024;
                const unsigned off = (cb + lane_off < x_end) ? lane_off : 0u;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(cb + off), (fr_lds_ptr_t)(ring + u * 256), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NCHT; ++u) {
                // uniform chunk base + 32-bit lane offset: the saddr form of global_load_lds, no VALU (the two empty asm statements
                // keep hipcc from re-associating the address into a 64-bit VALU add per chunk: k_full_stageA_lds)
                const char *cb = reinterpret_cast<const char *>(base) + (size_t)u * 1024;
                unsigned off = lane_off;
                asm("" : "+s"(cb));
                asm("" : "+v"(off));
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(cb + off), (fr_lds_ptr_t)(ring + u * 256), 16, 0, 0);
            }
        }
    };
    // The claim of a further row: one returning atomic per wave, asked for at the top of a row and read behind the walk's wait of the
    // same row (inline asm: with hipcc's own bookkeeping the first use of the result would drain the DMAs issued in between --
    // see `batch`).  Between the two statements the register holds nothing yet; nothing reads it there.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int grp = (int)(((xcc & 7u) * 4u + ((blockIdx.x >> 3) & 3u)) % FR_GROUPS);
    const int pool_per = n > nstatic ? (n - nstatic + FR_GROUPS - 1) / FR_GROUPS : 0;
    const int pool0 = nstatic + grp * pool_per;                       // this group's rows: [pool0, pool0 + pool_len)
    const int pool_len = max(0, min(n, pool0 + pool_per) - pool0);
    unsigned *const counter = gate + 64 + 32 * (grp + FR_GROUPS * parity);
    if (blockIdx.x == 0 && threadIdx.x < FR_GROUPS) gate[64 + 32 * ((int)threadIdx.x + FR_GROUPS * (parity ^ 1))] = 0u;      // the next launch's set
    unsigned claim_raw = 0u;
    auto claim_ask = [&]() __attribute__((always_inline)) {
        // (lane 0 alone adds: the other 63 would each take a row)
        const unsigned one = 1u;
        unsigned long long keep;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                     : "=v"(claim_raw), "=&s"(keep)
                     : "v"(counter), "v"(one)
                     : "memory");
    };
    auto claim_get = [&]() __attribute__((always_inline)) {        // (behind a wait that covers the atomic)
        asm volatile("" : "+v"(claim_raw));
        const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane((int)claim_raw);
        return got < (unsigned)pool_len ? pool0 + (int)got : -1;
    };
    int cur = slab ? -1 : ((int)blockIdx.x - nslab) * FR_WAVES + wid; // wave-uniform: the row being worked on; the first one is static
    if (cur >= n) cur = -1;
    const bool more = pool_len > 0;                                   // (whether there is anything to claim at all)
    if (cur >= 0) issue_row(cur);
    if (more && (slab || cur >= 0)) claim_ask();                      // the second row (a slab block's wave: its first)
    // ---- the reference vector into LDS (behind the first row's DMAs: one round trip covers both).  `ref` is allocated
    // FD_REF_PAD floats past F, so no upper clamp.
    {
        constexpr int PER = (NCHT * 256 + 4 + 64 * FR_WAVES - 1) / (64 * FR_WAVES);
        float r[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) r[u] = ref[max(u * 64 * FR_WAVES + tid - 2, 0)];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int js = u * 64 * FR_WAVES + tid;
            if (js < NCHT * 256 + 4) sref[js] = (js >= 2 && js - 2 < F) ? r[u] : 0.f;
        }
    }
    fr_wait<0>();                                                     // (the claim and the first row are in, whatever hipcc counts)
    __syncthreads();
    const int c0 = 4 * lane;
    const bool own = c0 < H;                                          // (H % 4 == 0: a lane holds four hidden columns or none)
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nxt = -1;
    if (slab) {
        cur = more ? claim_get() : -1;
        if (cur < 0) return;
        issue_row(cur);
        claim_ask();
        fr_wait<0>();
        nxt = claim_get();
    } else {
        if (cur < 0) return;
        nxt = more ? claim_get() : -1;
    }
    f64x4 crefv = {0.0, 0.0, 0.0, 0.0};
    if (cref && own) crefv = *reinterpret_cast<const f64x4 *>(cref + c0);
    // fixed-point words / fp64 values of a finished row, and the store of them
    auto finish_row = [&](const double (&acc)[4], int (&q)[4], double &scale, f64x4 &o) __attribute__((always_inline)) {
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
    auto store_row = [&](int row, const int (&q)[4], double scale, const f64x4 &o) __attribute__((always_inline)) {
        if (S1x) {
            if (own) *reinterpret_cast<int4 *>(S1x + (size_t)row * H + c0) = make_int4(q[0], q[1], q[2], q[3]);
            if (lane == 0) S1qs[row] = scale;
        } else if (own) {
            *reinterpret_cast<f64x4 *>(S1d + (size_t)row * H + c0) = o;
        }
        if (zstate && lane == 0) zstate[row] = 0;
    };
    bool asked = false;                                               // a claim is in flight
    int prow = -1, pq[4] = {0, 0, 0, 0};
    double pscale = 1.0;
    f64x4 po = {0.0, 0.0, 0.0, 0.0};
    while (cur >= 0) {
        const char *rp = reinterpret_cast<const char *>(X + (long)cur * ldx);
        const int shift = (int)((reinterpret_cast<uintptr_t>(rp) & 15) >> 2);
        FR_STAMP(0);
        fr_wait<0>();                                                 // the row has landed (nothing younger than its DMAs is in flight)
        FR_STAMP(1);
        if (asked) {                                                  // (asked behind this row's DMAs, in the previous row's walk)
            nxt = claim_get();
                    asked = false;
        }
        const int jb0 = 4 * lane - shift;                             // column of the lane's first value of chunk 0
        unsigned fl[4] = {0u, 0u, 0u, 0u};
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
                    const int bit = 4 * u + v;                        // flags word bit / 32, two accumulators per word (v & 1)
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
                flags &= flags - 1ull;                                // (0 stays 0)
                const int u = bit >> 2, v = bit & 3;
                jq[t] = u * 256 + jb0 + v;
                xq[t] = ring[u * 256 + 4 * lane + v];
                rq[t] = sref[max(jq[t], 0) + 2];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const unsigned long long m = __ballot(has[t]);
                const int pos = total + __popcll(m & lt);
                if (has[t] && pos < FR_CAP) { mj[pos] = jq[t]; mv[pos] = (double)xq[t] - (double)rq[t]; }
                total += __popcll(m);
            }
        }
        while (__ballot(flags != 0ull)) {                             // lanes with more than three (a handful of rows)
            const bool has = flags != 0ull;
            const unsigned long long m = __ballot(has);
            if (has) {
                const int bit = __ffsll((long long)flags) - 1;
                flags &= flags - 1ull;
                const int u = bit >> 2, v = bit & 3, j = u * 256 + jb0 + v;
                const float xq = ring[u * 256 + 4 * lane + v], rq = sref[j + 2];
                const int pos = total + __popcll(m & lt);
                if (pos < FR_CAP) { mj[pos] = j; mv[pos] = (double)xq - (double)rq; }
            }
            total += __popcll(m);
        }
        if (prow >= 0) store_row(prow, pq, pscale, po);               // (a row late: see 4.)
        FR_STAMP(2);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const unsigned c0b = own ? 16u * (unsigned)lane : 0u;         // byte offset of the lane's columns in a W1 row
        // entries [e, e + cnt), cnt <= FR_P: their W1 rows asked for (inline asm: with an LDS-DMA in flight hipcc (ROCm 7.2) puts
        // s_waitcnt vmcnt(0) in front of the first use of any ordinary load's result -- that would drain the DMAs issued just
        // behind them), (dma) the next row's DMAs and the claim, the hand-counted wait -- exactly NCHT + 1 operations are younger than the last load;
        // "memory" clobbers keep every other memory operation outside, an extra one would only make the wait stricter -- then
        // the sums in list order
        auto batch = [&](int e, int cnt, bool dma) __attribute__((always_inline)) {
            f32x4 w[FR_P];
#pragma unroll
            for (int k = 0; k < FR_P; ++k) {
                if (k < cnt) {
                    const char *p = reinterpret_cast<const char *>(W1) + (size_t)(unsigned)mj[e + k] * (unsigned)(4 * H) + c0b;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[k]) : "v"(p) : "memory");
                }
            }
            if (dma) {
                issue_row(nxt);
                claim_ask();                                          // the row after that one; read at the top of the next row
                asked = true;
                fr_wait<NCHT + 1>();
            } else {
                fr_wait<0>();
            }
#pragma unroll
            for (int k = 0; k < FR_P; ++k) {
                if (k < cnt) {
                    asm volatile("" : "+v"(w[k]));                   // (the value exists from here on)
                    const double d = mv[e + k];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = fma(d, (double)w[k][t], acc[t]);
                }
            }
        };
        bool dma_due = nxt >= 0;                                      // (whether the next row's DMAs are still to be issued)
        auto walk = [&](int cnt) __attribute__((always_inline)) {
            for (int e = 0; e < cnt; e += FR_P) { batch(e, min(cnt - e, FR_P), dma_due); dma_due = false; }
        };
        if (total <= FR_CAP) {
            walk(total);
            if (dma_due) { issue_row(nxt); claim_ask(); asked = true; }       // (an empty list)
        } else {
            // a dense row: read again piece by piece, every piece's list walked before the next is made (slow and correct; the
            // hint below moves the baseline off this route)
            if (dma_due) { issue_row(nxt); claim_ask(); asked = true; }
            dma_due = false;
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
            fr_wait<0>();
        }
        FR_STAMP(3);
        if (total > hint_cap && lane == 0) *dense_hint = 1;
        finish_row(acc, pq, pscale, po);
        prow = cur;
        FR_STAMP(4);
        ++fr_slot;
        cur = nxt;                                                    // (its successor is read at the top, behind the drain)
        nxt = -1;
    }
    if (prow >= 0) store_row(prow, pq, pscale, po);
}
