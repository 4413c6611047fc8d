// Item-list helpers shared by the SPARSE / DELTA probe kernels (lt_influence.hip) and the 3-layer path (lt_gcn3.hip):
// per-probe offsets into item lists, the membership bitmap with positions, the finite-difference tail.
#pragma once
#include "lt_rows.hip.h"

__device__ __forceinline__ int find_probe(const int32_t *__restrict__ off, int nb, int item) {
    int lo = 0, hi = nb;  // off[lo] <= item < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}

// position of `c` in the ascending list rows[0..cnt), or -1
__device__ __forceinline__ int find_row(const int32_t *__restrict__ rows, int cnt, int c) {
    int lo = 0, hi = cnt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = rows[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid;
    }
    return -1;
}


// ---- node ids out of range (include/linkteller_hip.h, lt_node_check) ----------------------------------------------------
// The lists are device memory the host side of the ABI cannot read; the reference raises IndexError (attacker.py:103, 226-229).
// The first kernel of a call that reads a list checks it: an id outside [0, n) becomes node 0 (no out-of-bounds access behind
// it) and raises err[which] in mapped host memory (a plain store: the flag only ever goes 0 -> 1 between two host reads).
__device__ __forceinline__ int checked_node(int v, int n, int32_t *__restrict__ err, int which) {
    if ((unsigned)v < (unsigned)n) return v;
    if (err) err[which] = 1;
    return 0;
}
// calls whose first reader is not one of the two blocks below (FULL / SPARSE: the perturbed-row GEMM gathers X[probes]; pair
// marks; the 3-layer path): both lists checked into the call's workspace by a launch of its own
static __global__ __launch_bounds__(256) void k_check_nodes(const int32_t *__restrict__ probes, int np, const int32_t *__restrict__ obs,
                                                            int no, int n, int32_t *__restrict__ probes_s, int32_t *__restrict__ obs_s,
                                                            int32_t *__restrict__ err) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < np) probes_s[i] = checked_node(probes[i], n, err, 0);
    else if (i - np < no) obs_s[i - np] = checked_node(obs[i - np], n, err, 1);
}

// bits[b][r >> 5] = { mask, base }: bit (r & 31) of mask = 1  <=>  r in R_v of probe b, and base = the position in
// R_v (the ascending CSC list of column v) of the lowest member of this word, so that ONE 8-byte load tells stage B
// both whether an entry of an observed row is affected by the probe and which item replaces it:
//     position(r) = base + popcount(mask & ((1 << (r & 31)) - 1))
// (without the bitmap -- huge graphs -- both questions are a binary search in R_v).  One block per probe.
// A call too large for a bitmap row per probe (bits == NULL) still gives one to its BIG probes -- the up to
// LT_BIG_SLOTS probes of the chunk whose R_v has more than LT_BIG_RV members (a hub probed): big_bits[slot], slot =
// big_slot[b] (or -1).  Searching such a list once per entry of an observed hub row is what made those pairs slow.
#define LT_BIG_RV 512
#define LT_BIG_SLOTS 64
static __device__ __forceinline__ void item_bits_block(const int bid, const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow,
                                                       const int32_t *__restrict__ probes, int nb, int words,
                                                       uint2 *__restrict__ bits, int32_t *__restrict__ off,
                                                       int2 *__restrict__ item_pr, uint2 *__restrict__ big_bits,
                                                       int32_t *__restrict__ big_slot, int32_t *__restrict__ big_count,
                                                       const int32_t *__restrict__ rowptr, const int32_t *__restrict__ observe, int n_obs,
                                                       int32_t *__restrict__ hub_obs, const float *__restrict__ tval = nullptr,
                                                       int2 *__restrict__ item_va = nullptr, int n_check = 0,
                                                       int32_t *__restrict__ err = nullptr, int32_t *__restrict__ probes_s = nullptr,
                                                       int32_t *__restrict__ obs_s = nullptr, uint2 *lds_row = nullptr) {
    // lds_row != NULL (the launch's dynamic LDS, >= words * 8 bytes: graphs whose bitmap row fits LT_IB_LDS_BYTES): the probe's bitmap
    // row is built in LDS and written out once -- two GLOBAL atomics per item made a probed hub's block (1 749 items on the
    // power-law co-headline graph) the long pole of the launch that carries these blocks (round 5)
    // One block more than probes (hub_obs != NULL or obs_s != NULL): it lists the observed nodes that are hub rows, hub_obs[0] =
    // how many, hub_obs[1 ...] = their positions j in `observe` (any order) -- stage B launches its hub blocks for those alone --
    // and writes the checked observed list (obs_s) for the kernels behind this one.
    if (bid == nb) {
        __shared__ int32_t s_n;
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n_obs; j += blockDim.x) {
            int u = observe[j];
            if (n_check > 0) u = checked_node(u, n_check, err, 1);
            if (obs_s) obs_s[j] = u;
            if (hub_obs && rowptr[u + 1] - rowptr[u] > LT_ROW_SEG) hub_obs[1 + atomicAdd(&s_n, 1)] = j;
        }
        __syncthreads();
        if (threadIdx.x == 0 && hub_obs) hub_obs[0] = s_n;
        return;
    }
    // One block per probe.  It also forms the probe's item offset off[b] = sum of |R_v| over the probes before it
    // (every block sums its own prefix: nb^2 / 2 four-byte loads in all, no scan kernel in front), the last block
    // writes the total off[nb].  bits == NULL: no bitmap (huge graphs); item_pr == NULL: no (probe, row) table.
    __shared__ int32_t red[4];
    __shared__ int32_t s_slot;
    const int b = bid;
    int v = probes[b];
    if (n_check > 0) v = checked_node(v, n_check, err, 0);
    if (probes_s && threadIdx.x == 0) probes_s[b] = v;
    const int t0 = tptr[v], t1 = tptr[v + 1];
    int part = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
        int vi = probes[i];
        if (n_check > 0) vi = checked_node(vi, n_check, nullptr, 0);
        part += tptr[vi + 1] - tptr[vi];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    if (threadIdx.x == 0) {
        int slot = -1;
        if (!bits && big_bits && t1 - t0 > LT_BIG_RV) {
            slot = atomicAdd(big_count, 1);
            if (slot >= LT_BIG_SLOTS) slot = -1;
        }
        s_slot = slot;
        if (big_slot) big_slot[b] = slot;
    }
    __syncthreads();
    uint2 *const mine_g = bits ? bits + (size_t)b * words : (s_slot >= 0 ? big_bits + (size_t)s_slot * words : nullptr);
    uint2 *mine = (mine_g && lds_row) ? lds_row : mine_g;
    if (mine)
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = make_uint2(0u, 0xffffffffu);
    __syncthreads();
    const int my_off = red[0] + red[1] + red[2] + red[3];
    if (threadIdx.x == 0) {
        off[b] = my_off;
        if (b == nb - 1) off[nb] = my_off + (t1 - t0);
    }
    int2 *items = item_pr ? item_pr + my_off : nullptr;
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int r = trow[t];
        if (mine) {
            atomicOr(&mine[r >> 5].x, 1u << (r & 31));
            atomicMin(&mine[r >> 5].y, (unsigned)(t - t0));
        }
        // item (off[b] + position in R_v) = (probe index, row): stage A reads it instead of searching `off`
        if (items) items[t - t0] = make_int2(b, r);
        // ... and (probe node, A_hat[r, v]): DELTA stage A would otherwise chase probes[b] -> tptr[v] -> tval[t] per item
        if (item_va) item_va[my_off + (t - t0)] = make_int2(v, __float_as_int(tval[t]));
    }
    if (mine_g && mine != mine_g) {
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine_g[i] = mine[i];
    }
}
#define LT_IB_LDS_BYTES (16 * 1024)      // bitmap rows of up to this many bytes (n <= 65536 nodes) are built in LDS
static inline unsigned lt_item_bits_smem(int words) { return (size_t)words * sizeof(uint2) <= LT_IB_LDS_BYTES ? (unsigned)(words * sizeof(uint2)) : 0u; }
static __global__ __launch_bounds__(256) void k_item_bits(const int32_t *__restrict__ tptr, const int32_t *__restrict__ trow,
                                                   const int32_t *__restrict__ probes, int nb, int words,
                                                   uint2 *__restrict__ bits, int32_t *__restrict__ off,
                                                   int2 *__restrict__ item_pr, uint2 *__restrict__ big_bits,
                                                   int32_t *__restrict__ big_slot, int32_t *__restrict__ big_count,
                                                   const int32_t *__restrict__ rowptr = nullptr,
                                                   const int32_t *__restrict__ observe = nullptr, int n_obs = 0,
                                                   int32_t *__restrict__ hub_obs = nullptr, const float *__restrict__ tval = nullptr,
                                                   int2 *__restrict__ item_va = nullptr, int n_check = 0,
                                                   int32_t *__restrict__ err = nullptr, int32_t *__restrict__ probes_s = nullptr,
                                                   int32_t *__restrict__ obs_s = nullptr, int lds_words = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ib_smem[];      // (lds_words * 8 bytes when the bitmap row fits)
    item_bits_block((int)blockIdx.x, tptr, trow, probes, nb, words, bits, off, item_pr, big_bits, big_slot, big_count, rowptr, observe,
                    n_obs, hub_obs, tval, item_va, n_check, err, probes_s, obs_s,
                    lds_words >= words ? reinterpret_cast<uint2 *>(ib_smem) : (uint2 *)nullptr);
}
// ---- the fused DELTA route, first half: a probe's INCIDENCE RECORD (round 4) -----------------------------------------------
// Everything about a probe v that depends on the graph alone is a property of NODE v, built once by lt_graph_create
// (lt_core.hip build_delta_records; lt_graph::dl_meta / dl_rec): the items (members r of R_v = the CSC column of v, with
// A_hat[r, v]) and, for every node u that holds a member, the entries of row u that are members -- as (A_hat[u, r], item << 16 |
// entry position in row u), the nodes ascending, a node's entries in entry order:
//     dl_meta[v] = (offset into dl_rec, items, touched nodes, incidences)
//     dl_rec + offset: items (r, A_hat[r, v]) [items] | nodes (u, start | count << 16) [touched] | entries [incidences]
// Per call and probe, delta_record_block resolves the record against the observed list -- for each observed position, is its
// node among the probe's touched nodes (a binary search in LDS), and where are its entries -- and writes what the finish
// kernel needs as one small table row (probes -> meta -> record -> search: dependent round trips and LDS passes the finish
// kernel then does not make; a kernel starts with cold caches, a trip is ~ 2 us):
//     [0] items  [1] short | long << 16 touched POSITIONS  [2] v  [3] offset of the node's entries in dl_rec
//     | items [maxc, the unused slots repeat the first] | touched positions (j, start | count << 16) [n_obs]: those of up to 4 entries from
//     the front, the longer ones (row v itself holds ALL of R_v) from the back, in no particular order
// Nothing here reads a layer, so these blocks ride in the launch that forms the pre-activation (k_spmm_f64), which hides them.
struct lt_df_inc { float a; int ik; };      // A_hat[u, r] and (item << 16 | entry position in row u)
static inline __host__ __device__ int lt_dl_rec_words(int maxc, int n_obs) { return (4 + 2 * maxc + 2 * n_obs + 3) & ~3; }
static __device__ __forceinline__ void delta_record_block(const int b, const lt_bits_job &J, unsigned char *smem) {
    int2 *sL = reinterpret_cast<int2 *>(smem);                  // [lcap] the probe's touched nodes (u, start | count << 16), u ascending
    __shared__ int32_t s_ntp, s_nlp;
    const int tid = threadIdx.x;
    // (the finish kernel reads the probe's node from this block's table row and never the observed list: checking here covers
    // the whole route)
    const int v = J.n > 0 ? checked_node(J.probes[b], J.n, J.err, 0) : J.probes[b];
    const int4 m = J.dl_meta[v];
    const int cnt = m.y, Tu = m.z;
    const int2 *__restrict__ src = reinterpret_cast<const int2 *>(J.dl_src + m.x);
    int32_t *R = J.dl_rec + (size_t)b * J.dl_rec_words;
    if (tid == 0) { s_ntp = 0; s_nlp = 0; }
    int2 *gi = reinterpret_cast<int2 *>(R + 4);
    // (every slot: the finish kernel loads them unseen -- and the rows they name; an unused slot repeats the probe's first item,
    // a row the block reads anyway.  Row 0 for all of them made every block of the launch hammer the same cache lines.)
    for (int i = tid; i < J.dl_maxc; i += 256) gi[i] = i < cnt ? src[i] : (cnt > 0 ? src[0] : make_int2(0, 0));
    for (int i = tid; i < Tu; i += 256) sL[i] = src[cnt + i];
    __syncthreads();
    int2 *gtp = reinterpret_cast<int2 *>(R + 4 + 2 * J.dl_maxc);
    constexpr int JP = 2;                                        // positions searched together (a step is one LDS trip for both)
    for (int j0 = 0; j0 < J.n_obs; j0 += JP * 256) {             // (block-uniform trips)
        int u[JP], pos[JP];
#pragma unroll
        for (int h = 0; h < JP; ++h) {
            const int j = j0 + tid + h * 256;
            u[h] = j < J.n_obs ? J.observe[j] : -1;
            if (b == 0 && j < J.n_obs && J.n > 0) (void)checked_node(u[h], J.n, J.err, 1);   // (an id out of range is found nowhere)
            pos[h] = 0;
        }
        for (int nrem = Tu; nrem > 1;) {                         // branch-free lower bound
            const int half = nrem >> 1;
#pragma unroll
            for (int h = 0; h < JP; ++h) pos[h] = sL[pos[h] + half - 1].x < u[h] ? pos[h] + half : pos[h];
            nrem -= half;
        }
#pragma unroll
        for (int h = 0; h < JP; ++h) {
            const int j = j0 + tid + h * 256;
            if (Tu > 0 && sL[pos[h]].x < u[h]) ++pos[h];
            if (j < J.n_obs && pos[h] < Tu) {
                const int2 le = sL[pos[h]];
                if (le.x == u[h]) {
                    if ((le.y >> 16) > 4) gtp[J.n_obs - 1 - atomicAdd(&s_nlp, 1)] = make_int2(j, le.y);
                    else gtp[atomicAdd(&s_ntp, 1)] = make_int2(j, le.y);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) *reinterpret_cast<int4 *>(R) = make_int4(cnt, s_ntp | (s_nlp << 16), v, m.x + 2 * (cnt + Tu));
}
static __global__ __launch_bounds__(256) void k_delta_records(const lt_bits_job job) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dr_smem[];
    delta_record_block((int)blockIdx.x, job, dr_smem);
}
// the same block as part of another launch (256 threads per block)
// lt_bits_job::zero_*: wave `w` of zero_blocks fills rows zero_row0 + w, + zero_blocks, ... of the caller's float64 matrix (pinned
// host memory as a rule) with +0.0 -- whole rows, 1-KiB stores, a bounded number in flight.  One wave per carrying block (the
// block's other waves leave at once).
static __device__ __forceinline__ void zero_rows_wave(const lt_bits_job &j, const int w) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, cols = j.zero_cols, cap = j.zero_inflight;
    const bool pair_ok = (reinterpret_cast<uintptr_t>(j.zero_dst) & 15) == 0 && (j.zero_ld & 1) == 0;
    for (int r = j.zero_row0 + w; r < j.zero_row0 + j.zero_rows; r += j.zero_blocks) {
        double *row = j.zero_dst + (long)r * j.zero_ld;
        for (int c = 2 * lane; c < cols; c += 128) {
            if (c + 1 < cols && pair_ok) *reinterpret_cast<double2 *>(row + c) = make_double2(0.0, 0.0);
            else { row[c] = 0.0; if (c + 1 < cols) row[c + 1] = 0.0; }
            if (cap <= 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (cap <= 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (cap <= 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (cap <= 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else asm volatile("" ::: "memory");
        }
    }
}

static __device__ __forceinline__ void item_bits_block(const int bid, const lt_bits_job &j, unsigned char *smem = nullptr) {
    if (j.dl_rec != nullptr) {      // (`smem`: the launch's dynamic LDS, j.smem_bytes)
        delta_record_block(bid, j, smem);
        return;
    }
    item_bits_block(bid, j.tptr, j.trow, j.probes, j.nb, j.words, j.bits, j.off, j.item_pr, j.big_bits, j.big_slot, j.big_count, j.rowptr,
                    j.observe, j.n_obs, j.hub_obs, j.tval, j.item_va, j.n, j.err, j.probes_s, j.obs_s,
                    (smem && j.smem_bytes >= (unsigned)j.words * sizeof(uint2) && j.smem_bytes > 0) ? reinterpret_cast<uint2 *>(smem) : (uint2 *)nullptr);
}
// position of column c in R_v from the probe's bitmap row, or -1
__device__ __forceinline__ int bits_pos(const uint2 *__restrict__ mb, int c) {
    const uint2 w = mb[c >> 5];
    const unsigned bit = 1u << (c & 31);
    return (w.x & bit) ? (int)(w.y + __popc(w.x & (bit - 1u))) : -1;
}

// ------------------------------------------------------------------------------------------------
// Pair marks (large calls): which (probe b, observed j) pairs does a probe reach at all?  Pair (b, j) is affected
// iff some entry r of row u_j is an item of probe b (r in R_v) -- a path u_j - r - v.  Testing that per pair costs
// deg(u_j) membership lookups for EVERY pair (2 M pairs x 31 entries at BASELINE configs[4], 98 % of them for
// nothing); the join over the middle node r costs (entries of the observed rows) + (items) + (paths):
//   k_pm_count / k_pm_alloc / k_pm_place   once per call: for every node r the list of observed j whose row holds r
//                                          (counts, a cursor-allocated slice of `list`, the members; the order inside a
//                                          slice is arbitrary -- marks are an OR)
//   k_pm_mark                              per probe chunk: item (b, r) -> set bit (b, j) for every j listed under r
// Observed hubs (rows of more than LT_ROW_SEG entries) are not listed: stage B serves them for every probe anyway
// (stageB_long_block), so an observed row contributes at most LT_ROW_SEG entries and slot k = j * LT_ROW_SEG + i needs
// no prefix sum.  Stage B then reads one bit per pair; the marked pairs are computed exactly as before.
// ------------------------------------------------------------------------------------------------
template <int PHASE>   // 0 count, 1 allocate, 2 place
static __global__ __launch_bounds__(256) void k_pm_lists(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const int32_t *__restrict__ observe, int n_obs,
                                                         int32_t *__restrict__ cnt, int32_t *__restrict__ start,
                                                         int32_t *__restrict__ rank, int32_t *__restrict__ list,
                                                         int32_t *__restrict__ cursor) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    const int j = (int)(k / LT_ROW_SEG), i = (int)(k % LT_ROW_SEG);
    bool on = j < n_obs;
    int r = 0;
    if (on) {
        const int u = observe[j];
        const int e0 = rowptr[u], d = rowptr[u + 1] - e0;
        on = d <= LT_ROW_SEG && i < d;
        if (on) r = col[e0 + i];
    }
    if (PHASE == 0) { if (on) rank[k] = atomicAdd(&cnt[r], 1); }
    else if (PHASE == 1) {
        // one slice of `list` per node: the slices' places come off ONE cursor -- a wave sums its claims (inclusive scan) and asks
        // once (round 5: one atomic per node on that single word was 50 us of serialised adds at BASELINE configs[4])
        const bool claim = on && rank[k] == 0;
        const int mine = claim ? cnt[r] : 0;
        int incl = mine;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int t = __shfl_up(incl, m, 64);
            if (lane >= m) incl += t;
        }
        // (round 6: ... and a BLOCK asks once -- returning adds on one word are served one after the other by that word's L2 channel,
        // ~6 ns each: the 8 K waves of a 4 096-node observed list still made this launch 51 us)
        __shared__ int s_tot[4];
        __shared__ int s_base;
        const int wid = threadIdx.x >> 6;
        if (lane == 63) s_tot[wid] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wid ? s_tot[w] : 0;
            all += s_tot[w];
        }
        if (threadIdx.x == 0) s_base = all > 0 ? atomicAdd(cursor, all) : 0;
        __syncthreads();
        if (claim) start[r] = s_base + before + incl - mine;
    } else if (on) list[start[r] + rank[k]] = j;
}

// 8 lanes per item: the item's list of observed nodes is a contiguous slice
static __global__ __launch_bounds__(256) void k_pm_mark(const int32_t *__restrict__ off, int nb, const int2 *__restrict__ item_pr,
                                                        const int32_t *__restrict__ cnt, const int32_t *__restrict__ start,
                                                        const int32_t *__restrict__ list, int n_obs,
                                                        unsigned *__restrict__ marks) {
    const int total = off[nb];
    const int q = threadIdx.x & 7;
    const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 3, gstride = (long)gridDim.x * 32;
    for (long item = g0; item < total; item += gstride) {
        const int2 pr = item_pr[item];
        const int c = cnt[pr.y];
        if (c == 0) continue;
        const int32_t *l = list + start[pr.y];
        const long base = (long)pr.x * n_obs;
        for (int t = q; t < c; t += 8) {
            const long p = base + l[t];
            atomicOr(&marks[p >> 5], 1u << (p & 31));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// shared tail: finite difference + L2 norm of one observed row          attacker.py:105-106,227-229
// ------------------------------------------------------------------------------------------------
// Vector form of the tail, for layers wider than one pass of these kernels (lt_influence_rows_vec): instead of the norm,
// the C differences of the pair go to vec[pair * C + c] UNSCALED -- SPARSE: (acc + b2) - base, the fp32 finite difference
// before the division by delta; DELTA (base == NULL): the propagated difference acc itself; an untouched pair: zeros --
// so that the caller can add the vectors of the hidden-layer slices (lt_wide_combine) before dividing and taking the norm.
template <int CP>
__device__ __forceinline__ void store_diff_vec(float *__restrict__ vec, long pair, int C, const float (&acc)[CP],
                                               const float *__restrict__ b2, const float *__restrict__ base, bool touched) {
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) vec[pair * C + c] = !touched ? 0.f : (base ? (acc[c] + b2[c]) - base[c] : acc[c]);
}

template <int CP>
__device__ __forceinline__ float diff_norm(const float (&acc)[CP], const float *__restrict__ b2,
                                           const float *__restrict__ base, int C, float delta) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (c < C) {
            const float o = acc[c] + b2[c];           // layers.py:34
            const float d = (o - base[c]) / delta;    // attacker.py:105-106
            ss = fmaf(d, d, ss);
        }
    return sqrtf(ss);
}

