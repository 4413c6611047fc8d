// The probe primitive: rows of the influence matrix.
//   reference: Attacker.get_gradient_eps_mat (attacker.py:100-108) + the inner loop of
//   link_prediction_attack_efficient (attacker.py:220-229):
//       pert = 0; pert[v] = X[v] * d;  grad = (model(X + pert, A) - model(X, A)) / d
//       influence_val[i][j] = || grad[test_nodes[j]] ||_2
//
// X + pert differs from X in row v only, so S1' = (X + pert) W1 differs from S1 = X W1 in row v
// only (K1-K3 of SURVEY.md 2.1 vanish).  Three evaluations of the same quantity:
//
//   FULL   per probe: s'_v = x'_v W1 (batched MFMA GEMM over the chunk's probes); stage A runs the
//          fused layer-1 (SpMM over ALL n rows with row v of S1 replaced, +b1, ReLU, .W2) for the
//          probe; stage B runs layer 2 on the observed rows, subtracts the baseline logits,
//          divides by d and takes the L2 norm.  P probes share each gathered S1 row in registers.
//   SPARSE the same arithmetic, restricted to rows that can differ from the baseline: layer 1 on
//          R_v = {r : A_hat[r,v] != 0}, layer 2 with those rows substituted.  Bit-identical to FULL.
//   DELTA  propagates dS1[v] = d * S1[v] itself: dZ1[r] = A_hat[r,v] dS1[v]; the ReLU difference is
//          evaluated piecewise-linearly (no subtraction of nearly equal numbers); layer 2 sums
//          A_hat[u,r] * dS2[r] over r in R_v.  Free of the fp32 cancellation noise of the
//          finite difference: agrees with an fp64 run of the reference.
#include <stdlib.h>

#include <type_traits>

#include "lt_rows.hip.h"
#include "lt_lanes.hip.h"
#include "lt_items.hip.h"

#define LT_BLOCK 256
#define LT_CHUNK_BUDGET ((size_t)1 << 30)  // bytes of per-probe scratch per chunk
#define LT_ITEM_GRID 2048                  // blocks of the grid-stride item kernels
#define LT_SB_AHEAD 24                     // FULL stage B: entries in flight per wave (a multiple of LT_L2_LANES)

// ------------------------------------------------------------------------------------------------
// FULL stage A, narrow hidden widths (LPR < 64): a lane group per (row, probe), pointer-select substitution
// ------------------------------------------------------------------------------------------------
template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_full_stageA(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S1, int Hp,
    const float *__restrict__ b1p, const float *__restrict__ W2p, int C,
    const int32_t *__restrict__ probes, int nbq, const float *__restrict__ Sp, float *__restrict__ S2p) {
    // One LPR-lane group per (row, column) pair, the column (probe, or nbq = the unperturbed layer) running
    // fastest: the 64 / LPR groups of a wave mostly share their row, so its S1 gathers are one request for the
    // whole wave instead of one per probe, and the groups' results are adjacent in S2p.
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int gl = lane & (LPR - 1);
    const int ncol = nbq + 1;
    const long gid = wave * RPW + lane / LPR;
    const int r = (int)(gid / ncol), b = (int)(gid % ncol);
    if (r >= n) return;
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    const f32x4 b1v = active ? ld4(b1p + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 acc = row_dot(col, val, rowptr[r], rowptr[r + 1], S1, Hp, coff, active, b < nbq ? probes[b] : -1,
                              Sp + (size_t)(b < nbq ? b : 0) * Hp, b1v);
    float part[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = 0.f;
    if (active) relu_w2_partial<CP>(acc, W2p + (size_t)coff * C, C, part);
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
    if (gl == 0) {
        float *dst = S2p + ((size_t)r * ncol + b) * C;   // [row][probe | baseline][class]
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) dst[c] = part[c];
    }
}

// ------------------------------------------------------------------------------------------------
// FULL stage A, wide hidden width (LPR = 64, 128 < Hp <= 256): a wave owns one row and P probes.
//  * the row's CSR entries are wave-uniform (scalar loads); each gathered S1 row feeds P independent
//    fmaf chains -> P x fewer L2 gather bytes per flop;
//  * blocks are one wave; the probe groups of a row are adjacent in the grid, so their gathers of that
//    row's S1 lines meet in L2;
//  * substituting the probe's own row is rare: the hot loop only asks "does this column equal one of my
//    P probes" (one vector compare per entry, probe ids sit one per lane, the answer is a wave mask in
//    SGPRs); the probes that were hit are recomputed after the epilogue, one single-probe chain each;
//  * the P*C (probe, class) partial sums are reduced together (lt_lanes.hip.h): the xor 32,16,8,4,2,1
//    butterfly of group_sum<64> with the live registers halving at every stage -- same pairings,
//    fp add commutes, so the bits equal the baseline kernel's -- and leave in one coalesced store.
// ------------------------------------------------------------------------------------------------
// fmaxf(x, 0) as exactly one v_max_f32: the accumulators pass through inline asm, so hipcc cannot prove
// them canonical and would put a v_max_f32 x, x, x in front of every fmaxf (same bits for every non-NaN x)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Epilogue of the wide stage-A kernel: h = relu(z) (the chains already hold z = A_hat S1' + b1), the lane's
// share of h . W2 for all P probes, the multi-value lane reduction, and the [row][probe][class] store.
template <int CP, int P>
__device__ __forceinline__ void stageA_epilogue(f32x4 (&acc)[P], int lane, bool active, int coff,
                                                const float *__restrict__ W2p, int C, int n, int nb,
                                                int r, int pb, float *__restrict__ S2p, unsigned skip,
                                                int base_p) {
    float w2[4 * CP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CP; ++c) w2[k * CP + c] = (active && c < C) ? W2p[(size_t)(coff + k) * C + c] : 0.f;

    float part[P * CP];
    if constexpr (CP == 2) {
        // two classes: the (c = 0, c = 1) pair rides in one packed register (v_pk_mul/v_pk_fma with the
        // hidden value broadcast through op_sel); per component the operations and their order are
        // those of relu_w2_partial, so the bits are unchanged
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 w0 = {w2[0], w2[1]}, w1 = {w2[2], w2[3]}, w2v = {w2[4], w2[5]}, w3 = {w2[6], w2[7]};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float h0 = relu1(acc[p].x), h1 = relu1(acc[p].y);
            const float h2 = relu1(acc[p].z), h3 = relu1(acc[p].w);
            f32x2 q = f32x2{h0, h0} * w0;
            q = __builtin_elementwise_fma(f32x2{h1, h1}, w1, q);
            q = __builtin_elementwise_fma(f32x2{h2, h2}, w2v, q);
            q = __builtin_elementwise_fma(f32x2{h3, h3}, w3, q);
            part[2 * p] = q.x;
            part[2 * p + 1] = q.y;
        }
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            // same operation order as relu_w2_partial
            const float h0 = fmaxf(acc[p].x, 0.f);
            const float h1 = fmaxf(acc[p].y, 0.f);
            const float h2 = fmaxf(acc[p].z, 0.f);
            const float h3 = fmaxf(acc[p].w, 0.f);
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                float q = h0 * w2[c];
                q = fmaf(h1, w2[CP + c], q);
                q = fmaf(h2, w2[2 * CP + c], q);
                q = fmaf(h3, w2[3 * CP + c], q);
                part[p * CP + c] = q;
            }
        }
    }
    // 64-lane totals of all P*CP values (lt_lanes.hip.h): in batches of up to 64 values, lane l ends with
    // the total of value q = batch * VB + (l >> log2(64 / VB)) = (probe q / CP, class q % CP), so one
    // batch leaves in one coalesced store of S2p[row][pb + p][c]
    constexpr int V = P * CP, VB = V < 64 ? V : 64;
#pragma unroll
    for (int j = 0; j < V / VB; ++j) {
        const float z = lane_totals<VB>(part + j * VB, lane);
        const int q = j * VB + lane_totals_owner<VB>(lane);
        const int p = q / CP, c = q % CP;
        const bool owner = (lane & (64 / VB - 1)) == 0;
        // S2p is [row][nb probes | baseline][class].  `skip`: probes (bits) whose entries the wave rewrites itself
        // (substituted recomputation); `base_p`: a probe of this wave whose chain met no substitution, i.e. IS the
        // unperturbed layer -- its totals also go to the baseline column (or -1)
        float *rowp = S2p + (size_t)r * (nb + 1) * C;
        if (owner && c < C && pb + p < nb && !((skip >> (p & 31)) & 1u)) rowp[(size_t)(pb + p) * C + c] = z;
        if (owner && c < C && p == base_p) rowp[(size_t)nb * C + c] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// The kernel (LDS ring, one wave per block).  The S1 rows of a row's
// entries are fetched by LDS-DMA (global_load_lds_dwordx4: one 1 KiB row per instruction, no VGPR
// destination) into a wave-private ring NB half-blocks (4 entries each) ahead of the FMAs, so
//   * loads in flight cost LDS, not registers (acc[P] + 4 staged rows: 92 VGPRs at P = 16, 156 at P = 32),
//   * the wait before each entry is a COUNTED s_waitcnt vmcnt(4*NB + 3 - k) -- never a drain,
//   * the (col, val) of a half-block arrive as one 4-dword scalar load each, a half-block early.
// Arithmetic is the same fmaf chain as every other layer-1 kernel (entries past the row end are skipped).
// ------------------------------------------------------------------------------------------------
#ifndef LT_RING_NB
#define LT_RING_NB 1   // half-blocks in flight behind the one being consumed (ring = 4*(NB+1) KiB per wave)
#endif
#define LT_RING_SLOTS (4 * (LT_RING_NB + 1))
#define LT_REDO_W 4     // substituted probes of a wave recomputed together
#define LT_REDO_AH 8    // entries in flight in that recomputation (16 for the single chain of a short row)
#ifndef LT_RING_ROWS
#define LT_RING_ROWS 1   // consecutive rows a wave walks (amortises wave launch)
#endif

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Long rows (more than LT_ROW_SEG entries: hubs) are not walked by the plain instantiation (MODE 0).  MODE 1
// takes one SEGMENT of a long row per block (same walk, over LT_ROW_SEG entries, the chains of a non-first
// segment starting from +0) and leaves the P (+1 baseline) segment sums in `lpart`; k_full_long_combine adds
// them in segment order -- row_dot's canonical order -- and runs the epilogue.  That costs P KiB of scratch per
// segment and probe group, fine for a graph with a few hubs; a graph with 10^5 long rows (R-MAT) uses MODE 2
// instead: one wave walks all the segments of its row one after the other and keeps the running sum in a
// second set of registers (so P <= 16 there).
template <int CP, int P, int MODE>   // 0: rows of up to LT_ROW_SEG entries; 1: one segment of a long row; 2: a whole long row
__global__ __launch_bounds__(64) void k_full_stageA_lds(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S1, int Hp,
    const float *__restrict__ b1p, const float *__restrict__ W2p, int C,
    const int32_t *__restrict__ probes, int nb, const float *__restrict__ Sp,
    float *__restrict__ S2p, int n_segblocks, const int32_t *__restrict__ seg_long,
    const int32_t *__restrict__ seg_begin, const int32_t *__restrict__ long_row,
    float *__restrict__ lpart, unsigned *__restrict__ lhit) {
    static_assert((P * CP) % 4 == 0, "P*CP must be a multiple of 4");
    __shared__ __attribute__((aligned(16))) float ring[LT_RING_SLOTS * 256];
    const int lane = threadIdx.x;
    const int groups = (nb + P - 1) / P;
    constexpr bool segmode = MODE == 1;   // separate instantiations: in one kernel the different tails cost 50-100 VGPRs of copies
    int r, pb, e0, e1, sg = 0;
    bool first_seg = true;
    if (segmode) {
        sg = blockIdx.x / groups;
        pb = (blockIdx.x % groups) * P;
        r = long_row[seg_long[sg]];
        e0 = seg_begin[sg];
        e1 = min(rowptr[r + 1], e0 + LT_ROW_SEG);
        first_seg = e0 == rowptr[r];
    } else if (MODE == 2) {
        pb = (blockIdx.x % groups) * P;
        r = long_row[blockIdx.x / groups];
        e0 = rowptr[r];
        e1 = rowptr[r + 1];
    } else {
        const int bid = blockIdx.x;
        pb = (bid % groups) * P;
        r = bid / groups;
        if (r >= n) return;
        e0 = rowptr[r];
        e1 = rowptr[r + 1];
        if (e1 - e0 > LT_ROW_SEG) return;   // a long row: MODE 1 / MODE 2 blocks take it
    }
  {
    const bool active = 4 * lane < Hp;
    const int coff = active ? 4 * lane : Hp - 4;
    int vprobe = (lane < P && pb + lane < nb) ? probes[pb + lane] : -1;
    asm volatile("" : "+v"(vprobe));   // retire this ordinary load before any LDS-DMA is in flight (hipcc would drain vmcnt(0) at its first use)

    // every chain starts from the bias (row_dot's `init`; +0 for the later segments of a long row); retired
    // before the DMAs for the same reason
    f32x4 b1v = first_seg ? ld4(b1p + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(b1v));
    f32x4 acc[P];   // set by the first entry walked: acc = fma(a_0, s_0, init) -- no initialisation pass
    unsigned long long hit = 0ull;
    unsigned lane_off = 4u * (unsigned)coff;
    // the chains of the entries [ws, we) (at most LT_ROW_SEG of them), started from `winit`, into acc[]
    auto walk = [&](const int ws, const int we, const f32x4 winit) {
        const int deg = we - ws;
        const int nh = (deg + 3) >> 2;

        auto load_cols = [&](int h, int (&c)[4], int eb) {
            const int32_t *cp = col + eb + 4 * h;     // may run past the row: stays inside the padded array
    #pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = cp[k];
        };
        auto load_vals = [&](int h, float (&a)[4], int eb) {
            const float *vp_ = val + eb + 4 * h;
    #pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = vp_[k];   // entries past the row end are never used (fmas stops at cnt)
        };
        auto issue = [&](int h, const int (&c)[4]) {
            const int base = (h % (LT_RING_NB + 1)) * 4;
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                // wave-level hit mask in SGPRs (v_cmp + s_or): bit l = lane l's probe sits on this column
                hit |= (4 * h + k < deg) ? __ballot(vprobe == c[k]) : 0ull;
                // uniform row base + 32-bit unsigned lane offset -> the saddr form of global_load_lds (no VALU)
                const char *rowp = reinterpret_cast<const char *>(S1 + (size_t)c[k] * Hp);
                // two empty asm fences: the first keeps hipcc from re-associating the address into
                // (S1 + lane_off) + row, the second re-defines lane_off in this basic block so that instruction
                // selection sees base + zext(32-bit offset); without them the add is a 64-bit VALU op per entry
                asm("" : "+s"(rowp));
                asm("" : "+v"(lane_off));
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(rowp + lane_off),
                                                 (lds_ptr_t)(ring + (base + k) * 256), 16, 0, 0);
            }
        };
        // half-block h out of the ring into registers; AFTER = number of half-blocks issued after it
        auto fetch = [&](int h, f32x4 (&s)[4], auto after_tag) {
            constexpr int AFTER = decltype(after_tag)::value;
            const float *slot = ring + (h % (LT_RING_NB + 1)) * 4 * 256 + 4 * lane;
            wait_vmcnt<4 * AFTER + 3>(); s[0] = *reinterpret_cast<const f32x4 *>(slot);
            wait_vmcnt<4 * AFTER + 2>(); s[1] = *reinterpret_cast<const f32x4 *>(slot + 256);
            wait_vmcnt<4 * AFTER + 1>(); s[2] = *reinterpret_cast<const f32x4 *>(slot + 512);
            wait_vmcnt<4 * AFTER + 0>(); s[3] = *reinterpret_cast<const f32x4 *>(slot + 768);
        };
        // the first `cnt` (>= 1) entries of a fetched half-block into all P chains; FIRST: the row's first
        // half-block, whose entry 0 starts every chain from the bias
        auto fmas = [&](const float (&a)[4], const f32x4 (&s)[4], int cnt, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if constexpr (FIRST) {
                // acc[p] = fma(a_0, s_0, b1) for every probe.  The P results are equal by construction (FULL mode
                // is P independent recomputations of the same row), and hipcc would compute one and copy it P
                // times; volatile asm keeps them P separate FMAs -- and P opaque values, so the chains that
                // continue from them stay separate too.
                const double ap = __builtin_bit_cast(double, f32x2_t{a[0], a[0]});
                const double slo = __builtin_bit_cast(double, f32x2_t{s[0].x, s[0].y});
                const double shi = __builtin_bit_cast(double, f32x2_t{s[0].z, s[0].w});
                const double blo = __builtin_bit_cast(double, f32x2_t{winit.x, winit.y});
                const double bhi = __builtin_bit_cast(double, f32x2_t{winit.z, winit.w});
    #pragma unroll
                for (int p = 0; p < P; ++p) {
                    double lo, hi;
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(ap), "v"(slo), "v"(blo));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(ap), "v"(shi), "v"(bhi));
                    const f32x2_t l2 = __builtin_bit_cast(f32x2_t, lo), h2 = __builtin_bit_cast(f32x2_t, hi);
                    acc[p] = f32x4{l2.x, l2.y, h2.x, h2.y};
                }
            }
    #pragma unroll
            for (int k = FIRST ? 1 : 0; k < 4; ++k) {
                if (k >= cnt) break;   // wave-uniform: entries past the row end are skipped (last half-block only)
    #pragma unroll
                for (int p = 0; p < P; ++p) acc[p] = fma4(a[k], s[k], acc[p]);
            }
        };
        static_assert(LT_RING_NB == 1, "the loop below is written for one half-block in flight behind the consumed one");
        if (nh == 0) {
    #pragma unroll
            for (int p = 0; p < P; ++p) {   // empty row: z = b1 (opaque copies, see fmas)
                f32x4 t = winit;
                asm volatile("" : "+v"(t));
                acc[p] = t;
            }
        } else {
            int cI[4], cN[4], cF[4];
            float aC[4], aN[4];
            f32x4 s[4];
            load_cols(0, cI, ws);
            load_vals(0, aC, ws);
            issue(0, cI);
            load_cols(1, cN, ws);   // one half-block of look-ahead (past a short row: inside the padding, unused)
            // Half-block h+1 goes in flight, half-block h is consumed.  The (col, val) scalar loads of the NEXT
            // iteration are issued once the first ring read of this one has landed (the empty asm ties their
            // address to it), i.e. in front of a block of FMAs that covers their latency; issued any earlier
            // they would share the lgkmcnt(0) that guards the ring reads.
            auto front = [&](int h) {
                issue(h + 1, cN);
                fetch(h, s, std::integral_constant<int, 1>{});
                int eb = ws;
                asm volatile("" : "+s"(eb) : "v"(s[0].x));
                load_cols(h + 2, cF, eb);
                load_vals(h + 1, aN, eb);
            };
            auto back = [&]() {
                // pin the look-ahead columns in SGPRs here (hipcc otherwise sinks the loads to the top of the
                // next iteration, right in front of their use)
                asm volatile("" : "+s"(cF[0]), "+s"(cF[1]), "+s"(cF[2]), "+s"(cF[3]));
    #pragma unroll
                for (int k = 0; k < 4; ++k) { aC[k] = aN[k]; cN[k] = cF[k]; }
            };
            // the row's first half-block starts the chains (one copy of that code for long and short rows)
            const bool more = nh > 1;
            if (more) front(0);
            else fetch(0, s, std::integral_constant<int, 0>{});
            fmas(aC, s, more ? 4 : deg, std::true_type{});
            if (more) {
                back();
                int h = 1;
                for (; h + 1 < nh; ++h) {
                    front(h);
                    fmas(aC, s, 4, std::false_type{});
                    back();
                }
                fetch(h, s, std::integral_constant<int, 0>{});
                fmas(aC, s, deg - 4 * h, std::false_type{});
            }
        }

    };
    f32x4 tot[MODE == 2 ? P : 1];   // MODE 2: sum of the segment sums so far
    if constexpr (MODE != 2) {
        walk(e0, e1, b1v);
    } else {
        for (int ws = e0;; ws += LT_ROW_SEG) {
            const int we = min(e1, ws + LT_ROW_SEG);
            walk(ws, we, ws == e0 ? b1v : f32x4{0.f, 0.f, 0.f, 0.f});
            if (ws == e0) {
#pragma unroll
                for (int p = 0; p < P; ++p) tot[p] = acc[p];
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    tot[p].x += acc[p].x; tot[p].y += acc[p].y; tot[p].z += acc[p].z; tot[p].w += acc[p].w;
                }
            }
            if (we >= e1) break;
        }
    }

    const unsigned hitmask = (unsigned)hit;   // bit p: probe pb+p sits on one of this row's columns (probe ids sit in lanes < P <= 32)
    // A column of this row is one of my probes (about P*deg/n of the waves): the chains above used the
    // UNSUBSTITUTED S1 row for all P probes -- right for every probe but the one(s) sitting on that column.
    // The epilogue leaves those out, and the wave recomputes them below, one single-probe chain each (the chain
    // every other kernel uses), with its accumulators dead and the row's S1 lines still warm in L1/L2.
    // Keeping the select path out of the hot loop is worth 50 VGPRs.
    // group 0 also delivers the baseline column: any of its probes that met no substitution
    const int nvalid = min(nb - pb, P);
    const unsigned valid = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    const unsigned clean = ~hitmask & valid;
    const int base_p = (pb == 0 && clean != 0u) ? __builtin_ctz(clean) : -1;
    const bool base_redo = pb == 0 && clean == 0u;   // every probe of group 0 sits on this row: recompute it plainly
    // segment sums go to lpart; uniform slot base + 32-bit lane offset = the saddr store form
    float *slot_base = segmode ? lpart + ((size_t)(sg * groups + pb / P) * (P + 1)) * Hp : nullptr;
    if constexpr (MODE == 0) {
        stageA_epilogue<CP, P>(acc, lane, active, coff, W2p, C, n, nb, r, pb, S2p, hitmask, base_p);
    } else if constexpr (MODE == 2) {
        stageA_epilogue<CP, P>(tot, lane, active, coff, W2p, C, n, nb, r, pb, S2p, hitmask, base_p);
    } else {
        // The chains that met no substitution hold the same bits (P recomputations of the unperturbed segment), so ONE of
        // them is stored -- slot P -- with the mask of the probes that did meet one (their own chains follow below, into
        // their own slots); k_full_long_combine picks per (segment, probe).  P times less scratch than a slot per probe:
        // 105 MB -> 3 MB per step on the power-law graph.
        const int keep = clean != 0u ? __builtin_ctz(clean) : -1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p == keep) {   // wave-uniform
                char *sp = reinterpret_cast<char *>(slot_base + (size_t)P * Hp);
                asm("" : "+s"(sp));
                *reinterpret_cast<f32x4 *>(sp + lane_off) = acc[p];
            }
        }
        if (lane == 0) lhit[sg * groups + pb / P] = hitmask & valid;
    }
    if (__builtin_expect(hitmask != 0u || base_redo, 0)) {
        // The probes that sit on one of the columns walked (and, if no chain of group 0 was clean, the unperturbed one)
        // get their own single chain: over the row, or over this segment of it.  On long rows up to LT_REDO_W of them walk together
        // -- every entry's S1 row is loaded once and serves them all, LT_REDO_AH entries in flight -- because on a hub
        // segment, where a wave meets 0.9 substitutions on average and some meet five, one latency-bound pass per probe
        // was the kernel's tail (133 us for a 60 us walk).
        constexpr int W = MODE == 0 ? 1 : LT_REDO_W, AH = LT_REDO_AH;   // short rows: a wave rarely meets two
        const f32x4 b1r = first_seg ? ld4(b1p + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned m = hitmask;
        bool do_base = base_redo;
        while (m || do_base) {
            int pj[W], vj[W];   // probe slot (-1: the unperturbed chain, -2: none) and its node
            f32x4 own[W], z[W], zt[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                pj[j] = -2;
                vj[j] = -1;
                if (m) {
                    pj[j] = __builtin_ctz(m);
                    m &= m - 1;
                    vj[j] = probes[pb + pj[j]];
                } else if (do_base) {
                    pj[j] = -1;
                    do_base = false;
                }
                own[j] = ld4(Sp + (size_t)(pb + (pj[j] < 0 ? 0 : pj[j])) * Hp + coff);
            }
            // MODE 0 / 1: e1 - e0 <= LT_ROW_SEG, one segment chain (row_dot of a short row is the same thing);
            // MODE 2: the whole long row, segment by segment, the sums added in segment order (row_dot)
            if constexpr (MODE == 0) {
                zt[0] = e1 > e0 ? seg_chain_clamped<16>(col, val, e0, e1, S1, Hp, coff, vj[0],
                                                        Sp + (size_t)(pb + (pj[0] < 0 ? 0 : pj[0])) * Hp, b1r)
                                : b1r;
            } else
            for (int ws = e0; ws == e0 || ws < e1; ws += LT_ROW_SEG) {
                const int we = min(e1, ws + LT_ROW_SEG);
#pragma unroll
                for (int j = 0; j < W; ++j) z[j] = ws == e0 ? b1r : f32x4{0.f, 0.f, 0.f, 0.f};
                for (int e = ws; e < we; e += AH) {
                    int c[AH];
                    float a[AH];
                    f32x4 sv[AH];
#pragma unroll
                    for (int k = 0; k < AH; ++k) {
                        const int ee = min(e + k, we - 1);   // past the end: the last entry again, unused
                        c[k] = col[ee];
                        a[k] = val[ee];
                    }
#pragma unroll
                    for (int k = 0; k < AH; ++k) sv[k] = ld4(S1 + (size_t)c[k] * Hp + coff);
#pragma unroll
                    for (int k = 0; k < AH; ++k) {
                        if (e + k >= we) break;   // wave-uniform
#pragma unroll
                        for (int j = 0; j < W; ++j) z[j] = fma4(a[k], c[k] == vj[j] ? own[j] : sv[k], z[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if (ws == e0) zt[j] = z[j];
                    else { zt[j].x += z[j].x; zt[j].y += z[j].y; zt[j].z += z[j].z; zt[j].w += z[j].w; }
                }
                if (MODE != 2) break;
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (pj[j] == -2) continue;   // wave-uniform
                if (segmode) {
                    char *sp = reinterpret_cast<char *>(slot_base + (size_t)(pj[j] < 0 ? P : pj[j]) * Hp);
                    *reinterpret_cast<f32x4 *>(sp + lane_off) = zt[j];
                } else {
                    float part[CP];
#pragma unroll
                    for (int c = 0; c < CP; ++c) part[c] = 0.f;
                    if (active) relu_w2_partial<CP>(zt[j], W2p + (size_t)coff * C, C, part);
#pragma unroll
                    for (int c = 0; c < CP; ++c) part[c] = group_sum<64>(part[c]);
                    if (lane == 0) {
                        float *dst = S2p + ((size_t)r * (nb + 1) + (pj[j] < 0 ? nb : pb + pj[j])) * C;
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) dst[c] = part[c];
                    }
                }
            }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// shared tail: finite difference + L2 norm of one observed row          attacker.py:105-106,227-229
// ------------------------------------------------------------------------------------------------
// FULL stage A, long rows: a wave = (long row, probe) -- or (long row, baseline) for probe index nb.  Adds the
// segment sums the SEG instantiation of k_full_stageA_lds left in `lpart`, in segment order (row_dot's
// canonical order), then the single-probe tail every other kernel uses (relu_w2_partial + group_sum<64>).
// One wave per probe rather than per probe group: a hub row has a dozen segments to read one after the other,
// and n_long * nb short waves hide that latency where n_long * groups long ones did not (114 -> 15 us).
// ------------------------------------------------------------------------------------------------
template <int CP, int P>
__global__ __launch_bounds__(LT_BLOCK) void k_full_long_combine(
    int Hp, const float *__restrict__ W2p, int C, int nb, int n_long, const int32_t *__restrict__ long_row,
    const int32_t *__restrict__ long_segptr, const float *__restrict__ lpart, const unsigned *__restrict__ lhit,
    float *__restrict__ S2p) {
    const int lane = threadIdx.x & 63;
    const long wid = ((long)blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int li = (int)(wid / (nb + 1)), b = (int)(wid % (nb + 1));
    if (li >= n_long) return;
    const int groups = (nb + P - 1) / P;
    const int g = b < nb ? b / P : 0, sl = b < nb ? b % P : P;
    const int r = long_row[li];
    const int s0 = long_segptr[li], s1 = long_segptr[li + 1];
    const bool active = 4 * lane < Hp;
    const int coff = active ? 4 * lane : Hp - 4;
    // slot P of a (segment, group) holds the unperturbed segment sum; a probe that sits on one of the segment's columns
    // (bit of lhit) has its own.  The masks of 64 segments are fetched at once, one per lane: `mine` = the segments (bits)
    // where this wave's probe has a slot of its own.
    auto masks = [&](int sb) {
        const int sx = sb + lane;
        const unsigned m = (b < nb && sx < s1) ? lhit[sx * groups + g] : 0u;
        return __ballot((m >> (sl & 31)) & 1u);
    };
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int sb = s0; sb < s1; sb += 64) {
        const unsigned long long mine = masks(sb);
        const int se = min(s1, sb + 64);
        auto slot = [&](int s) {
            const int own = ((mine >> (s - sb)) & 1ull) ? sl : P;
            return *reinterpret_cast<const f32x4 *>(lpart + ((size_t)(s * groups + g) * (P + 1) + own) * Hp + coff);
        };
        int s = sb;
        if (sb == s0) z = slot(s++);   // the first segment's sum starts the total (it holds the bias)
        for (; s + 8 <= se; s += 8) {   // 8 segment sums in flight, added in segment order
            f32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = slot(s + k);
#pragma unroll
            for (int k = 0; k < 8; ++k) { z.x += t[k].x; z.y += t[k].y; z.z += t[k].z; z.w += t[k].w; }
        }
        for (; s < se; ++s) {
            const f32x4 t = slot(s);
            z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w;
        }
    }
    float part[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = 0.f;
    if (active) relu_w2_partial<CP>(z, W2p + (size_t)coff * C, C, part);
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = group_sum<64>(part[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) S2p[((size_t)r * (nb + 1) + b) * C + c] = part[c];
    }
}

// ------------------------------------------------------------------------------------------------
// FULL stage B: a wave = one observed node x 64 probes (lane = probe).  S2p is [row][probe | baseline][class],
// so every CSR entry of the observed row is one coalesced 64 x C x 4-byte load; (col, val) are
// wave-uniform scalars.  The sum is formed exactly as row2_dot forms it -- 8 interleaved partial
// chains (entry e goes to chain (e - e0) & 7) combined as ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7)), the
// xor 4,2,1 butterfly -- so the bits equal the baseline's layer-2 kernel.  The baseline logits of the
// observed node come from the same walk over the baseline column (wave-uniform loads): FULL mode needs
// neither k_layer1 nor k_layer2.
// WPB waves per block share one (observed node, 64 probes): wave w walks the entries e0 + w, e0 + w + WPB, ...
// i.e. the chains w, w + WPB, ... (WPB divides 8) and the chain sums meet in LDS.  WPB = 4 when the graph has
// hub rows (an observed hub is 10^3 entries), 1 otherwise.
template <int CP, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_full_stageB(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S2p, int C,
    const float *__restrict__ b2,
    const int32_t *__restrict__ observe, int n_obs, int nb, float delta, float *__restrict__ out,
    long ldo) {
    constexpr int NCH = LT_L2_LANES / WPB;   // chains per wave
    const int lane = threadIdx.x & 63;
    // wave index as a scalar: the (col, val) and baseline-column loads below stay scalar loads
    const int w = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pblocks = (nb + 63) >> 6;
    const int j = blockIdx.x / pblocks;
    const int b = (blockIdx.x % pblocks) * 64 + lane;
    const int u = observe[j];
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    const bool live = b < nb;
    const size_t rstride = (size_t)(nb + 1) * C;
    const float *T = S2p + (size_t)(live ? b : 0) * C;
    const float *Tb = S2p + (size_t)nb * C;
    float part[NCH][CP], pbase[NCH][CP];
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int c = 0; c < CP; ++c) part[q][c] = pbase[q][c] = 0.f;
    // LT_SB_AHEAD entries are loaded before their FMAs run (in entry order, so every chain keeps its order): an
    // observed hub with 10^3 entries is 10^3 / 8 dependent round trips otherwise
    constexpr int AHEAD = LT_SB_AHEAD / WPB < NCH ? NCH : LT_SB_AHEAD / WPB;   // a multiple of NCH
    static_assert(AHEAD % NCH == 0, "entries in flight per wave must cover whole rounds of its chains");
    for (int e = e0 + w; e < e1; e += AHEAD * WPB) {
        float a[AHEAD], tv[AHEAD][CP], tbv[AHEAD][CP];
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            const int ee = e + q * WPB;
            const bool in = ee < e1;
            a[q] = in ? val[ee] : 0.f;
            const size_t ro = (size_t)(in ? col[ee] : 0) * rstride;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                tv[q][c] = (in && c < C) ? T[ro + c] : 0.f;
                tbv[q][c] = (in && c < C) ? Tb[ro + c] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < AHEAD; ++q)
            if (e + q * WPB < e1) {
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) {
                        part[q % NCH][c] = fmaf(a[q], tv[q][c], part[q % NCH][c]);
                        pbase[q % NCH][c] = fmaf(a[q], tbv[q][c], pbase[q % NCH][c]);
                    }
            }
    }
    // chain k of the row = local chain k / WPB of wave k % WPB
    float ch[LT_L2_LANES][CP], chb[LT_L2_LANES][CP];
    if constexpr (WPB == 1) {
#pragma unroll
        for (int k = 0; k < LT_L2_LANES; ++k)
#pragma unroll
            for (int c = 0; c < CP; ++c) { ch[k][c] = part[k][c]; chb[k][c] = pbase[k][c]; }
    } else {
        __shared__ float xch[2][LT_L2_LANES][CP][64];
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                xch[0][w + q * WPB][c][lane] = part[q][c];
                xch[1][w + q * WPB][c][lane] = pbase[q][c];
            }
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int k = 0; k < LT_L2_LANES; ++k)
#pragma unroll
            for (int c = 0; c < CP; ++c) { ch[k][c] = xch[0][k][c][lane]; chb[k][c] = xch[1][k][c][lane]; }
    }
    float acc[CP], base[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const float t0 = ch[0][c] + ch[4][c], t1 = ch[1][c] + ch[5][c];
        const float t2 = ch[2][c] + ch[6][c], t3 = ch[3][c] + ch[7][c];
        acc[c] = (t0 + t2) + (t1 + t3);
        const float u0 = chb[0][c] + chb[4][c], u1 = chb[1][c] + chb[5][c];
        const float u2 = chb[2][c] + chb[6][c], u3 = chb[3][c] + chb[7][c];
        base[c] = c < C ? ((u0 + u2) + (u1 + u3)) + b2[c] : 0.f;   // = OUT[u] of k_layer2, bit for bit
    }
    if (live) out[(long)b * ldo + j] = diff_norm<CP>(acc, b2, base, C, delta);
}

// ------------------------------------------------------------------------------------------------
// SPARSE / DELTA: item lists.  Probe b owns items [off[b], off[b+1]): one per r in R_v (CSC column v)
// ------------------------------------------------------------------------------------------------
// SPARSE stage A: for every item (b, r in R_v) the layer-1 row with S1[v] replaced -> S2x[item, :]
// DELTA  stage A: for every item the layer-1 *change*                              -> S2x[item, :]
template <int LPR, int CP, int DELTA>  // 0: sparse recompute, 1: delta (fp32 Z1), 2: delta (fp64 Z1)
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageA(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ tval,
    const float *__restrict__ S1, const float *__restrict__ Z1, const double *__restrict__ Z1d,
    const double *__restrict__ S1d, int Hp,
    const float *__restrict__ b1p, const float *__restrict__ W2p, int C,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ Sp, float delta, float *__restrict__ S2x, int n_long,
    const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_segptr,
    const float *__restrict__ seg_part, const int2 *__restrict__ item_pr, const double *__restrict__ Spd,
    const double *__restrict__ crefv, const float *__restrict__ S1x) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    const int total = off[nb];
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * (LT_BLOCK / 64);
    for (int base = wave0 * RPW; base < total; base += nwaves * RPW) {
        const int item = base + lane / LPR;
        const bool live = item < total;  // group-uniform; dead groups still join the shuffles
        float part[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = 0.f;
        if (live) {
            // (probe index, row) of the item: written by k_item_bits (one load instead of a search in `off`)
            const int2 pr = item_pr[item];
            const int b = pr.x, r = pr.y;
            const int v = probes[b];
            const int t = tptr[v] + (item - off[b]);
            if (active) {
                const f32x4 b1v = ld4(b1p + coff);
                if (DELTA) {
                    const float arv = tval[t];
                    f32x4 s;
                    if (DELTA == 2) {   // the probe's S1 row off the fp64 product (the fp32 one is not even computed for this mode)
                        // (aggregate-first route: the probes' own product rows Spd[b], there is no S1d)
                        const double *sp = Spd ? Spd + (size_t)b * Hp + coff : S1d + (size_t)v * Hp + coff;
                        if (false) {
                        } else if (crefv && !Spd)     // deferred cref: S1d holds the rows without the reference vector's product
                            s = f32x4{(float)(sp[0] + crefv[coff]), (float)(sp[1] + crefv[coff + 1]), (float)(sp[2] + crefv[coff + 2]),
                                      (float)(sp[3] + crefv[coff + 3])};
                        else
                        s = f32x4{(float)sp[0], (float)sp[1], (float)sp[2], (float)sp[3]};
                    } else {
                        s = ld4(S1 + (size_t)v * Hp + coff);
                    }
                    float dh[4];
                    if (DELTA == 2) {
                        // kink test on the fp64-accumulated pre-activation (see lt_fp64.hip)
                        const double *zp = Z1d + (size_t)r * Hp + coff;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float dz = arv * (delta * s[k]);
                            const double z = zp[k], z1 = z + (double)dz;
                            dh[k] = z > 0.0 ? (z1 > 0.0 ? dz : (float)(-z)) : (z1 > 0.0 ? (float)z1 : 0.f);
                        }
                    } else {
                        const f32x4 z = ld4(Z1 + (size_t)r * Hp + coff);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float dz = arv * (delta * s[k]);
                            const float z1 = z[k] + dz;
                            dh[k] = z[k] > 0.f ? (z1 > 0.f ? dz : -z[k]) : (z1 > 0.f ? z1 : 0.f);
                        }
                    }
                    const float *w2 = W2p + (size_t)coff * C;
#pragma unroll
                    for (int c = 0; c < CP; ++c)
                        if (c < C) {
                            float p = dh[0] * w2[c];
                            p = fmaf(dh[1], w2[C + c], p);
                            p = fmaf(dh[2], w2[2 * C + c], p);
                            p = fmaf(dh[3], w2[3 * C + c], p);
                            part[c] = p;
                        }
                } else {
                    const int e0 = rowptr[r], e1 = rowptr[r + 1];
                    f32x4 acc;
                    if (e1 - e0 > LT_ROW_SEG && seg_part != nullptr) {
                        // a hub row: only the segment that holds column v differs from the baseline, whose
                        // segment sums k_layer1_seg left in seg_part -- recompute that one, re-add in order
                        int lo = 0, hi = n_long - 1;                 // r's index among the long rows
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (long_row[mid] < r) lo = mid + 1; else hi = mid; }
                        const int sb = long_segptr[lo], ns = long_segptr[lo + 1] - sb;
                        int pl = e0, ph = e1 - 1;                   // position of v in the (ascending) columns of r
                        while (pl < ph) { const int mid = (pl + ph) >> 1; if (col[mid] < v) pl = mid + 1; else ph = mid; }
                        const int sstar = (pl - e0) / LT_ROW_SEG;
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                        for (int s = 0; s < ns; ++s) {
                            f32x4 t;
                            if (s == sstar) {
                                const int es = e0 + s * LT_ROW_SEG;
                                t = seg_chain(col, val, es, min(e1, es + LT_ROW_SEG), S1, Hp, coff, true, v,
                                              Sp + (size_t)b * Hp, s == 0 ? b1v : f32x4{0.f, 0.f, 0.f, 0.f});
                            } else {
                                t = ld4(seg_part + (size_t)(sb + s) * Hp + coff);
                            }
                            if (s == 0) acc = t;
                            else { acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
                        }
                    } else {
                        acc = row_dot(col, val, e0, e1, S1, Hp, coff, true, v, Sp + (size_t)b * Hp, b1v);
                    }
                    relu_w2_partial<CP>(acc, W2p + (size_t)coff * C, C, part);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
        if (live && gl == 0) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) S2x[(size_t)item * C + c] = part[c];
        }
    }
}

// DELTA stage A on the fp64 pre-activation, the form every `delta` call takes (k_item_stageA<.., 2> computes the same values):
// an item needs (probe node v, row r, A_hat[r, v]) -- one 16-byte pair of loads from the tables k_item_bits left, instead of
// the chain item_pr -> probes[b] -> tptr[v] -> tval[t] -- then the probe's product row and the row's pre-activation, both
// addressed from the tables.  TWO items per lane group are in flight (all loads unconditional, indices clamped: hipcc drains
// the queue in front of a guarded load); W2 and the deferred reference product of the lane's 4 hidden columns stay in
// registers across the items.  Same operations in the same order as k_item_stageA: the bits are its bits.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// SX / ZF: the probe's product row from S1x (fp32) / the pre-activation from Z1x (fp32) -- compile-time, so that no load sits
// behind a branch
template <int LPR, int CP, bool SX, bool ZF>
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageA_d2(
    const double *__restrict__ Z1d, const double *__restrict__ S1d, const float *__restrict__ S1x,
    const double *__restrict__ Spd, const double *__restrict__ crefv, int Hp, const float *__restrict__ W2p, int C,
    int nb, const int32_t *__restrict__ off, float delta, float *__restrict__ S2x, const int2 *__restrict__ item_pr,
    const int2 *__restrict__ item_va, const float *__restrict__ Z1x,        // Z1x != NULL: the pre-activation in fp32 (k_spmm_f64)
    const double *__restrict__ S1qs) {                                      // with S1x: the scales of its fixed-point rows (lt_fp64.hip)
    constexpr int RPW = 64 / LPR, U = 2;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    const bool active = 4 * gl < Hp;
    const int coff = active ? 4 * gl : 0;
    const int total = off[nb];
    const int wave0 = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int stride = gridDim.x * (LT_BLOCK / 64) * RPW;
    float w2[4][CP];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CP; ++c) w2[k][c] = c < C ? W2p[(size_t)(coff + k) * C + c] : 0.f;
    double cr[4] = {0.0, 0.0, 0.0, 0.0};
    const bool add_cref = crefv != nullptr && Spd == nullptr;
    if (add_cref) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cr[k] = crefv[coff + k];
    }
    for (int base = wave0 * RPW; base < total; base += U * stride) {
        int item[U];
        int2 pr[U], va[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            item[u] = base + u * stride + lane / LPR;
            const int ic = min(item[u], total - 1);        // past the end: the last item again, never stored
            pr[u] = item_pr[ic];
            va[u] = item_va[ic];
        }
        f64x4 z[U], sd[U];
        f32x4 sx[U];
        double sq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (ZF) {
                const f32x4 zf = ld4(Z1x + (size_t)pr[u].y * Hp + coff);
                z[u] = f64x4{(double)zf[0], (double)zf[1], (double)zf[2], (double)zf[3]};
            } else {
                z[u] = *reinterpret_cast<const f64x4 *>(Z1d + (size_t)pr[u].y * Hp + coff);
            }
            if constexpr (SX) {
                sx[u] = ld4(S1x + (size_t)va[u].x * Hp + coff);
                sq[u] = S1qs[va[u].x];
            }
            else sd[u] = *reinterpret_cast<const f64x4 *>((Spd ? Spd + (size_t)pr[u].x * Hp : S1d + (size_t)va[u].x * Hp) + coff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float arv = __int_as_float(va[u].y);
            float part[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) part[c] = 0.f;
            if (active) {
                float dh[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // the probe's S1 row off the fp64 product (fp32 storage and / or deferred reference product: lt_fp64.hip)
                    typedef int qx4 __attribute__((ext_vector_type(4)));
                    const float sk = SX ? (float)((double)__builtin_bit_cast(qx4, sx[u])[k] * sq[u] + cr[k])
                                        : (add_cref ? (float)(sd[u][k] + cr[k]) : (float)sd[u][k]);
                    // kink test on the fp64-accumulated pre-activation
                    const float dz = arv * (delta * sk);
                    const double zz = z[u][k], z1 = zz + (double)dz;
                    dh[k] = zz > 0.0 ? (z1 > 0.0 ? dz : (float)(-zz)) : (z1 > 0.0 ? (float)z1 : 0.f);
                }
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) {
                        float p = dh[0] * w2[0][c];
                        p = fmaf(dh[1], w2[1][c], p);
                        p = fmaf(dh[2], w2[2][c], p);
                        p = fmaf(dh[3], w2[3][c], p);
                        part[c] = p;
                    }
            }
#pragma unroll
            for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
            if (item[u] < total && gl == 0) {
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) S2x[(size_t)item[u] * C + c] = part[c];
            }
        }
    }
}

// SPARSE / DELTA stage B for an OBSERVED HUB (a row of more than LT_ROW_SEG entries).  A block (the first blocks of
// k_item_stageB's launch) takes one observed hub and 32 probes, 8 chain lanes per probe.  A pair (hub u, probe v) is
// affected through the members of row(u) /\ R_v, and which entries those are is found from the SHORT side:
//   light probe (R_v short against the row -- see `heavy` below): its 8 lanes look every r of R_v up in the
//       hub's sorted columns (branch-free lower bound, LT_SBL_NS searches in flight per lane) -- |R_v| log d loads
//       instead of one membership test per ENTRY of the row (a 43 075-entry hub x 512 probes of ~ 30 neighbours each
//       took 7.5 ms of lane-serial lookups that way).  DELTA adds the members it finds straight into the chains and
//       never walks the row; SPARSE keeps them as a short sorted list and walks the row (it must: the whole chain is
//       re-summed) comparing against the next member -- no lookups in the walk.
//       The list holds LT_SBL_MC members at a time and is refilled as the walk passes them.
//   heavy probe (R_v long against the row): one membership test per entry as before -- through the probe's bitmap
//       row when it has one (all probes of a small call, the big probes of a large one: k_item_bits).
// The row's (col, val, baseline S2 row) are fetched into LDS once per chunk by all 256 threads and shared by the 32
// probes; the walk happens only if some probe of the block needs it.  Same chains in every route (entry e -> chain
// (e - e0) & 7, k-ordered, non-members skipped in DELTA), same butterfly, same tail: the bits of k_item_stageB.
#define LT_SBL_CHUNK 1024
#define LT_SBL_UN 8
#define LT_SBL_UN_DELTA 32
#define LT_SBL_NS 8        // light probes: searches in flight per lane
#define LT_SBL_MC 128      // SPARSE: members of a light probe kept in LDS at a time (> 64 + a search round: see fill)
// WIDE (DELTA): 32 membership tests in flight per lane instead of 8 -- 211 VGPRs.  Not launched any more: at twitch size the plain
// form is as fast (14.2 vs 15.0 us) and, at 76 VGPRs, rides in front of k_item_stageB_rows' launch; at BASELINE configs[4] (5000 hub
// blocks) the wide form was slower (0.76 -> 1.28 ms)
template <int CP, bool DELTA, bool SHORT, bool WIDE = false>   // SHORT: the short-side search (without it every probe tests every entry, no member lists in LDS)
__device__ __forceinline__ void stageB_long_block(
    int bid, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, const float *__restrict__ OUT,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs, float delta,
    float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words,
    const uint2 *__restrict__ big_bits, const int32_t *__restrict__ big_slot, const int32_t *__restrict__ hub_obs, int hub_cap,
    float *__restrict__ vec = nullptr) {
    constexpr int CHUNK = CP <= 2 ? LT_SBL_CHUNK : (CP <= 4 ? LT_SBL_CHUNK / 2 : LT_SBL_CHUNK / 4);   // LDS: <= 16 KB
    constexpr int GROUPS = LT_BLOCK / LT_L2_LANES;
    // entries of the walk whose membership tests are in flight per lane: DELTA keeps nothing else per entry (no baseline
    // row), and an observed hub of 1 700 entries is 27 dependent rounds at 8
    constexpr int UN = (DELTA && WIDE) ? LT_SBL_UN_DELTA : LT_SBL_UN;
    __shared__ int sc[CHUNK];
    __shared__ float sv[CHUNK];
    __shared__ float sT[(DELTA && !(DELTA && SHORT)) ? 1 : CHUNK][CP];     // SPARSE: the baseline rows of the chunk; DELTA + SHORT: the members' item rows (below)
    __shared__ int2 smem[(DELTA || !SHORT) ? 1 : GROUPS][(DELTA || !SHORT) ? 1 : LT_SBL_MC];   // SPARSE, light probes: (entry - e0, position in R_v)
    // DELTA, heavy probes with a bitmap row (round 5): the WHOLE block finds the members of a chunk (4 entries per thread, compacted in
    // entry order), the probe's 8 chain lanes then add them in that order -- 8 lanes testing every entry themselves made the pair
    // (a probed hub, the 49 489-entry observed hub) of BASELINE configs[4] 773 dependent rounds: 0.7 ms, the whole launch
    // (SHORT only: without the short-side search every probe walks the row -- the twitch-size launch, where the plain walk wins)
    constexpr bool COOP = DELTA && SHORT;
    __shared__ int2 s_mem[COOP ? CHUNK : 1];                   // (entry in the chunk, position in R_v) of the members, ascending
    __shared__ const uint2 *s_mb[COOP ? GROUPS : 1];
    __shared__ const float *s_items[COOP ? GROUPS : 1];
    __shared__ int s_wc[COOP ? (CHUNK / LT_BLOCK) * (LT_BLOCK / 64) : 1];
    __shared__ unsigned s_hmask, s_bmask;
    __shared__ const int32_t *s_rv[COOP ? GROUPS : 1];
    __shared__ int s_cnt[COOP ? GROUPS : 1];
    const int pblocks = (nb + GROUPS - 1) / GROUPS;
    // the launch has blocks for hub_cap = min(n_obs, hub rows of the graph) observed hubs; hub_obs (k_item_bits) lists the
    // positions there are -- MORE than hub_cap when observe_nodes repeats a hub (a star graph observed twice): a block then
    // serves slot, slot + hub_cap, ... (block-uniform trips; usually one or none)
    for (int slot = bid / pblocks; slot < hub_obs[0]; slot += hub_cap) {
    const int j = hub_obs[1 + slot];
    const int u = observe[j];
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    const int d = e1 - e0;
    const int tid = threadIdx.x;
    const int q = tid & (LT_L2_LANES - 1);
    const int grp = tid / LT_L2_LANES;
    const int b = (bid % pblocks) * GROUPS + grp;
    const bool live = b < nb;
    const int v = probes[live ? b : 0];
    const int32_t *rv = trow + tptr[v];
    const int cnt = tptr[v + 1] - tptr[v];
    const float *items = S2x + (size_t)off[live ? b : 0] * C;
    const uint2 *mb = nullptr;
    if (bits) mb = bits + (size_t)(live ? b : 0) * words;
    else if (big_slot && live && big_slot[b] >= 0) mb = big_bits + (size_t)big_slot[b] * words;
    auto pos = [&](int c) { return mb ? bits_pos(mb, c) : find_row(rv, cnt, c); };
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    bool touch = false;
    // ---- light probes: the members of row(u) /\ R_v from the R_v side ---------------------------------------------
    // (group-uniform) per-entry tests cost d / 64 rounds of one load (a miss, ~ 2x an L2 hit) with a bitmap row, of
    // log |R_v| loads without; the search from the R_v side costs |R_v| / (8 NS) rounds of log d loads
    const int lg_d = 32 - __clz(d), lg_c = 32 - __clz(cnt > 1 ? cnt : 1);
    // with a bitmap row, in dependent rounds: the walk is d / (8 lanes * UN) of them, the search (lg d + 2) per 64 members
    // COOP (the whole block finds a heavy probe's members, below): a pass over the row costs ~2 us per CHUNK entries whatever the
    // probe; the search from the R_v side costs, per round of 64 members, lg(sample) LDS steps (~0.3 us together) and samp_sh
    // DEPENDENT global loads (~0.7 us each) -- a row of up to CHUNK entries sits in LDS whole and is never worth a pass.
    // (BASELINE configs[4]: probes of 2 442 and 2 314 neighbours against the 49 489-entry observed hub counted as light under the
    // round-4 rule "cnt lg d > d": 39 rounds x 16 global steps each, 0.7 ms, the whole launch)
    int samp_sh = 0;
    if constexpr (COOP) { while ((d >> samp_sh) > CHUNK) ++samp_sh; }
    const bool heavy = live && (!SHORT || (mb ? (COOP ? (long)((cnt + 63) / 64) * (3 + 7 * samp_sh) > 20L * ((d + CHUNK - 1) / CHUNK)
                                                      : ((DELTA && WIDE) ? (long)((cnt + 63) / 64) * (lg_d + 2) > (d + LT_L2_LANES * UN - 1) / (LT_L2_LANES * UN)
                                                                         : (long)cnt * lg_d > d))
                                              : 2L * cnt * lg_d > (long)d * lg_c));
    // COOP, a third way (round 5): the WHOLE BLOCK searches a big probe's R_v in the row -- CHUNK members per pass, a handful of
    // two-level searches per thread in flight, the finds compacted in R_v (= entry) order and added by the probe's chain lanes.
    // What a round of the probe's own 8-lane search costs is its dependent trips under a chip full of such blocks, ~10 us whatever
    // the row (tools/stageb_lab2.py, BASELINE configs[4]: the eight probes of 714 .. 2 442 neighbours against the 138 observed hubs
    // of under 1 000 entries -- rows that sit in LDS whole -- were 0.68 of the launch's 0.78 ms, 12 .. 39 rounds each; against the
    // 49 489-entry hub 39 rounds x 16 global steps, 0.7 ms); a block pass over CHUNK members ~20 us, one big probe after the other; a
    // pass over the row (needs a bitmap row) ~5 us per CHUNK entries.  So: up to four rounds a probe searches for itself (the 32
    // probes of a block side by side), beyond that the cheaper of the two block forms.
    bool bsearch = false;
    bool heavy_ = heavy;
    if constexpr (COOP) {
        if (live) {
            if ((cnt + 63) / 64 > 4) {
                const long c_block = (long)((cnt + CHUNK - 1) / CHUNK) * (20 + 7 * samp_sh);
                const long c_walk = mb ? 5L * ((d + CHUNK - 1) / CHUNK) : (1L << 40);
                bsearch = c_block <= c_walk;
                heavy_ = !bsearch;            // (mb != NULL then: the walk was the cheaper one)
            } else heavy_ = false;
        }
    }
    const bool light = live && !heavy_ && !bsearch;
    const bool coop = COOP && heavy_ && mb != nullptr;          // (group-uniform)
    if (COOP) {
        if (tid == 0) { s_hmask = 0u; s_bmask = 0u; }
        __syncthreads();
        if (coop && q == 0) { atomicOr(&s_hmask, 1u << grp); s_mb[grp] = mb; s_items[grp] = items; }
        if (bsearch && q == 0) { atomicOr(&s_bmask, 1u << grp); s_rv[grp] = rv; s_cnt[grp] = cnt; s_items[grp] = items; }
    }
    const int32_t *cu = col + e0;
    int n_samp = 0;
    if constexpr (COOP) {       // the search sample of this observed hub (block-uniform; sc is restaged by the walk afterwards)
        n_samp = d >> samp_sh;
        for (int i = tid; i < n_samp; i += LT_BLOCK) sc[i] = cu[((i + 1) << samp_sh) - 1];
        __syncthreads();
    }
    const int gl0 = (tid & 63) & ~(LT_L2_LANES - 1);    // first lane of the group inside its wave
    int nmem = 0;       // SPARSE: members in the list (group-uniform)
    int r_next = 0;     // next position of R_v to look up (group-uniform)
    // one round: R_v[r_next + (0 .. 8 * NS)) looked up in the hub's columns, the finds handed round the group in
    // ascending order of R_v (= ascending entry).  DELTA: chain (e & 7) adds its own; SPARSE: appended to the list.
    auto search_round = [&]() {
        int fe[LT_SBL_NS], key[LT_SBL_NS], base[LT_SBL_NS];
#pragma unroll
        for (int k = 0; k < LT_SBL_NS; ++k) {
            const int idx = r_next + k * LT_L2_LANES + q;
            key[k] = idx < cnt ? rv[idx] : 0x7fffffff;
            base[k] = 0;
        }
        if constexpr (COOP) {
            // two levels (round 5): the last column of every 2^samp_sh-entry stretch of the row sits in LDS (sc, staged once per
            // observed hub below) -- the upper levels of every search cost LDS trips, only the last samp_sh steps dependent
            // global loads (49 489 entries: 10 + 6 instead of 16; a row of up to CHUNK entries is searched in LDS alone)
            for (int nrem = n_samp; nrem > 1;) {
                const int half = nrem >> 1;
#pragma unroll
                for (int k = 0; k < LT_SBL_NS; ++k) base[k] = sc[base[k] + half - 1] < key[k] ? base[k] + half : base[k];
                nrem -= half;
            }
#pragma unroll
            for (int k = 0; k < LT_SBL_NS; ++k) {
                if (n_samp > 0 && sc[base[k]] < key[k]) ++base[k];
                base[k] <<= samp_sh;                       // first entry of the stretch that holds the lower bound (or d)
            }
            for (int nrem = 1 << samp_sh; nrem > 1;) {     // (a stretch cut short by the row's end: entries past it read as the last)
                const int half = nrem >> 1;
#pragma unroll
                for (int k = 0; k < LT_SBL_NS; ++k) base[k] = cu[min(base[k] + half - 1, d - 1)] < key[k] ? base[k] + half : base[k];
                nrem -= half;
            }
            if (samp_sh == 0) {         // the row sits in LDS whole: no global trip at all
#pragma unroll
                for (int k = 0; k < LT_SBL_NS; ++k) fe[k] = (base[k] < d && sc[min(base[k], d - 1)] == key[k]) ? base[k] : -1;
            } else {
#pragma unroll
                for (int k = 0; k < LT_SBL_NS; ++k) {
                    const int cb_ = cu[min(base[k], d - 1)];
                    if (base[k] < d && cb_ < key[k]) ++base[k];
                    fe[k] = (base[k] < d && cu[min(base[k], d - 1)] == key[k]) ? base[k] : -1;
                }
            }
        } else {
        // branch-free lower bound over cu[0, d): LT_SBL_NS independent searches per lane, one load each per step
        for (int nrem = d; nrem > 1;) {
            const int half = nrem >> 1;
#pragma unroll
            for (int k = 0; k < LT_SBL_NS; ++k) base[k] = cu[base[k] + half - 1] < key[k] ? base[k] + half : base[k];
            nrem -= half;
        }
#pragma unroll
        for (int k = 0; k < LT_SBL_NS; ++k) {
            if (cu[base[k]] < key[k]) ++base[k];
            fe[k] = (base[k] < d && cu[base[k]] == key[k]) ? base[k] : -1;
        }
        }
#pragma unroll
        for (int k = 0; k < LT_SBL_NS; ++k)
#pragma unroll
            for (int t = 0; t < LT_L2_LANES; ++t) {
                const int e = __shfl(fe[k], gl0 + t, 64);
                if (e < 0) continue;                       // group-uniform
                const int p = r_next + k * LT_L2_LANES + t;   // position in R_v
                if (DELTA) {
                    touch = true;
                    if ((e & (LT_L2_LANES - 1)) == q) {
                        const float a = val[e0 + e];
                        const float *tr = items + (size_t)p * C;
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) acc[c] = fmaf(a, tr[c], acc[c]);
                    }
                } else {
                    if (q == 0) smem[grp][nmem] = make_int2(e, p);
                    ++nmem;
                }
            }
        r_next += LT_L2_LANES * LT_SBL_NS;
        if (!DELTA) __builtin_amdgcn_wave_barrier();    // lane 0's list writes stay ahead of the group's reads
    };
    // SPARSE: (re)fill the member list -- drop the members every lane has passed (entry < base_e), search on until the
    // list holds more than LT_SBL_MC - 8 * LT_SBL_NS members (then it reaches past the 64 entries the group handles next) or R_v
    // is exhausted
    auto fill = [&](int base_e) {
        int f = 0;
        while (f < nmem && smem[grp][f].x < base_e) ++f;
        if (f > 0) {
            for (int i = q; i < nmem - f; i += LT_L2_LANES) {     // lockstep: the reads of a trip precede its writes
                const int2 m = smem[grp][f + i];
                __builtin_amdgcn_wave_barrier();
                smem[grp][i] = m;
            }
            __builtin_amdgcn_wave_barrier();
            nmem -= f;
        }
        while (r_next < cnt && nmem <= LT_SBL_MC - LT_L2_LANES * LT_SBL_NS) search_round();
    };
    if (light) {
        if (DELTA) while (r_next < cnt) search_round();
        else fill(0);
    }
    // ---- big probes searched by the whole block (the sample is still in sc; sv / sT / s_mem are free until the walk) ----------
    if constexpr (COOP) {
        constexpr int SL = CHUNK / LT_BLOCK, NW = LT_BLOCK / 64;
        __syncthreads();                                        // (the groups' own searches are done: s_bmask, s_rv, s_cnt are set)
        for (unsigned bm_ = s_bmask; bm_ != 0u; bm_ &= bm_ - 1u) {              // (block-uniform)
            const int g = __ffs((int)bm_) - 1;
            const int32_t *rvg = s_rv[g];
            const int cg = s_cnt[g];
            const float *itg = s_items[g];
            for (int p0_ = 0; p0_ < cg; p0_ += CHUNK) {
                int key[SL], base[SL], fe[SL];
#pragma unroll
                for (int k = 0; k < SL; ++k) {
                    const int idx = p0_ + k * LT_BLOCK + tid;
                    key[k] = idx < cg ? rvg[idx] : 0x7fffffff;
                    base[k] = 0;
                }
                for (int nrem = n_samp; nrem > 1;) {
                    const int half = nrem >> 1;
#pragma unroll
                    for (int k = 0; k < SL; ++k) base[k] = sc[base[k] + half - 1] < key[k] ? base[k] + half : base[k];
                    nrem -= half;
                }
#pragma unroll
                for (int k = 0; k < SL; ++k) {
                    if (n_samp > 0 && sc[base[k]] < key[k]) ++base[k];
                    base[k] <<= samp_sh;
                }
                for (int nrem = 1 << samp_sh; nrem > 1;) {
                    const int half = nrem >> 1;
#pragma unroll
                    for (int k = 0; k < SL; ++k) base[k] = cu[min(base[k] + half - 1, d - 1)] < key[k] ? base[k] + half : base[k];
                    nrem -= half;
                }
                unsigned long long bmk[SL];
#pragma unroll
                for (int k = 0; k < SL; ++k) {
                    if (samp_sh == 0) {
                        fe[k] = (base[k] < d && sc[min(base[k], d - 1)] == key[k]) ? base[k] : -1;
                    } else {
                        const int cb_ = cu[min(base[k], d - 1)];
                        if (base[k] < d && cb_ < key[k]) ++base[k];
                        fe[k] = (base[k] < d && cu[min(base[k], d - 1)] == key[k]) ? base[k] : -1;
                    }
                    bmk[k] = __ballot(fe[k] >= 0);
                    if ((tid & 63) == 0) s_wc[k * NW + (tid >> 6)] = __popcll(bmk[k]);
                }
                __syncthreads();
                int nm = 0;
#pragma unroll
                for (int k = 0; k < SL; ++k) {
                    int before = 0;
#pragma unroll
                    for (int w_ = 0; w_ < NW; ++w_) {
                        const int cw = s_wc[k * NW + w_];
                        before += w_ < (tid >> 6) ? cw : 0;
                        nm += cw;
                    }
                    if (fe[k] >= 0) {
                        int slot = before + __popcll(bmk[k] & ((1ull << (tid & 63)) - 1ull));
#pragma unroll
                        for (int k2 = 0; k2 < SL; ++k2)
                            if (k2 < k)
#pragma unroll
                                for (int w_ = 0; w_ < NW; ++w_) slot += s_wc[k2 * NW + w_];
                        s_mem[slot] = make_int2(fe[k], p0_ + k * LT_BLOCK + tid);     // (entry, position in R_v): ascending in both
                    }
                }
                __syncthreads();
                for (int m = tid; m < nm; m += LT_BLOCK) {      // the members' values and item rows, one trip for the block
                    const int2 me_ = s_mem[m];
                    sv[m] = val[e0 + me_.x];
                    const float *t = itg + (size_t)me_.y * C;
#pragma unroll
                    for (int c = 0; c < CP; ++c) sT[m][c] = c < C ? t[c] : 0.f;
                }
                __syncthreads();
                if (grp == g && nm > 0) {                       // the probe's own lanes: chain (entry & 7), entry order
                    touch = true;
                    int m = 0;
                    for (; m + 8 <= nm; m += 8) {
                        int ex[8];
                        float a8[8], t8[8][CP];
#pragma unroll
                        for (int x = 0; x < 8; ++x) {
                            ex[x] = s_mem[m + x].x;
                            a8[x] = sv[m + x];
#pragma unroll
                            for (int c = 0; c < CP; ++c) t8[x][c] = sT[m + x][c];
                        }
#pragma unroll
                        for (int x = 0; x < 8; ++x)
                            if ((ex[x] & (LT_L2_LANES - 1)) == q) {
#pragma unroll
                                for (int c = 0; c < CP; ++c)
                                    if (c < C) acc[c] = fmaf(a8[x], t8[x][c], acc[c]);
                            }
                    }
                    for (; m < nm; ++m)
                        if ((s_mem[m].x & (LT_L2_LANES - 1)) == q) {
#pragma unroll
                            for (int c = 0; c < CP; ++c)
                                if (c < C) acc[c] = fmaf(sv[m], sT[m][c], acc[c]);
                        }
                }
                __syncthreads();
            }
        }
    }
    // ---- the walk: heavy probes test every entry, SPARSE light probes with members re-sum the row against their list
    const bool walk = live && (heavy_ || (!DELTA && nmem > 0));
    if (__syncthreads_or(walk ? 1 : 0)) {
        // SPARSE light (all group-uniform): mg = first listed member not yet behind the group, next_e = its entry,
        // last_e = the entry of the last listed member
        int mg = 0;
        int next_e = (!DELTA && nmem > 0) ? smem[grp][0].x : 0x7fffffff;
        int last_e = (!DELTA && nmem > 0) ? smem[grp][nmem - 1].x : -1;
        for (int cb = e0; cb < e1; cb += CHUNK) {
            const int nc = min(CHUNK, e1 - cb);
            __syncthreads();
            for (int i = tid; i < nc; i += LT_BLOCK) {
                const int cc = col[cb + i];
                sc[i] = cc;
                sv[i] = val[cb + i];
                if (!DELTA) {
#pragma unroll
                    for (int c = 0; c < CP; ++c) sT[i][c] = c < C ? S2[(size_t)cc * C + c] : 0.f;
                }
            }
            __syncthreads();
            if constexpr (COOP) {
                constexpr int SL = CHUNK / LT_BLOCK, NW = LT_BLOCK / 64;
                for (unsigned hm = s_hmask; hm != 0u; hm &= hm - 1u) {          // (block-uniform)
                    const int g = __ffs((int)hm) - 1;
                    const uint2 *mbg = s_mb[g];
                    uint2 wv[SL];
                    int cc[SL];
#pragma unroll
                    for (int k = 0; k < SL; ++k) {                              // unconditional loads (past the end: the last entry again)
                        cc[k] = sc[min(k * LT_BLOCK + tid, nc - 1)];
                        wv[k] = mbg[cc[k] >> 5];
                    }
                    unsigned long long bm[SL];
                    bool mem[SL];
#pragma unroll
                    for (int k = 0; k < SL; ++k) {
                        mem[k] = k * LT_BLOCK + tid < nc && (wv[k].x & (1u << (cc[k] & 31))) != 0u;
                        bm[k] = __ballot(mem[k]);
                        if ((tid & 63) == 0) s_wc[k * NW + (tid >> 6)] = __popcll(bm[k]);
                    }
                    __syncthreads();
                    int nm = 0;
#pragma unroll
                    for (int k = 0; k < SL; ++k) {
                        int before = 0;
#pragma unroll
                        for (int w_ = 0; w_ < NW; ++w_) {
                            const int cw = s_wc[k * NW + w_];
                            before += w_ < (tid >> 6) ? cw : 0;
                            nm += cw;
                        }
                        if (mem[k]) {
                            const unsigned bit = 1u << (cc[k] & 31);
                            int slot = before + __popcll(bm[k] & ((1ull << (tid & 63)) - 1ull));
#pragma unroll
                            for (int k2 = 0; k2 < SL; ++k2)
                                if (k2 < k)
#pragma unroll
                                    for (int w_ = 0; w_ < NW; ++w_) slot += s_wc[k2 * NW + w_];
                            s_mem[slot] = make_int2(k * LT_BLOCK + tid, (int)(wv[k].y + __popc(wv[k].x & (bit - 1u))));
                        }
                    }
                    __syncthreads();
                    // the members' item rows, fetched by the whole block in one trip (one dependent global load per member in the
                    // chain lanes' loop made a pair of 1 000 common neighbours a millisecond)
                    {
                        const float *itg = s_items[g];
                        for (int m = tid; m < nm; m += LT_BLOCK) {
                            const float *t = itg + (size_t)s_mem[m].y * C;
#pragma unroll
                            for (int c = 0; c < CP; ++c) sT[m][c] = c < C ? t[c] : 0.f;
                        }
                    }
                    __syncthreads();
                    if (grp == g && nm > 0) {                                  // the probe's own lanes: chain (entry & 7), entry order
                        touch = true;
                        int m = 0;
                        for (; m + 4 <= nm; m += 4) {
                            int ex[4];
                            float a4[4], t4[4][CP];
#pragma unroll
                            for (int x = 0; x < 4; ++x) ex[x] = s_mem[m + x].x;
#pragma unroll
                            for (int x = 0; x < 4; ++x) {
                                a4[x] = sv[ex[x]];
#pragma unroll
                                for (int c = 0; c < CP; ++c) t4[x][c] = sT[m + x][c];
                            }
#pragma unroll
                            for (int x = 0; x < 4; ++x)
                                if ((ex[x] & (LT_L2_LANES - 1)) == q) {
#pragma unroll
                                    for (int c = 0; c < CP; ++c)
                                        if (c < C) acc[c] = fmaf(a4[x], t4[x][c], acc[c]);
                                }
                        }
                        for (; m < nm; ++m) {
                            const int ex = s_mem[m].x;
                            if ((ex & (LT_L2_LANES - 1)) == q) {
                                const float a = sv[ex];
#pragma unroll
                                for (int c = 0; c < CP; ++c)
                                    if (c < C) acc[c] = fmaf(a, sT[m][c], acc[c]);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            if (!walk || coop) continue;
            for (int i = q; i < nc; i += LT_L2_LANES * UN) {
                int mp[UN];
                float a_[UN], tb[DELTA ? 1 : UN][CP];
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int ii = i + k * LT_L2_LANES;
                    mp[k] = -1;
                    a_[k] = ii < nc ? sv[ii] : 0.f;
#pragma unroll
                    for (int c = 0; c < CP; ++c)
                        if (!DELTA) tb[k][c] = ii < nc ? sT[ii][c] : 0.f;
                }
                if (heavy_ && mb) {
                    // the membership words of UN entries in flight together: unconditional loads (past the chunk end: its
                    // last entry again) -- hipcc drains the queue in front of every load it finds behind a branch
                    uint2 wv[UN];
                    int cv[UN];
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        cv[k] = sc[min(i + k * LT_L2_LANES, nc - 1)];
                        wv[k] = mb[cv[k] >> 5];
                    }
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        const unsigned bit = 1u << (cv[k] & 31);
                        if (i + k * LT_L2_LANES < nc && (wv[k].x & bit)) mp[k] = (int)(wv[k].y + __popc(wv[k].x & (bit - 1u)));
                    }
                } else if (heavy_) {
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        const int ii = i + k * LT_L2_LANES;
                        if (ii < nc) mp[k] = pos(sc[ii]);
                    }
                } else if (!DELTA) {
                    const int base_e = cb - e0 + (i - q);       // the group's next 64 entries start here
                    const int end_e = base_e + LT_L2_LANES * UN;
                    if (r_next < cnt && last_e < end_e) {       // the list may end inside them: refill it
                        fill(base_e);
                        mg = 0;
                        next_e = nmem > 0 ? smem[grp][0].x : 0x7fffffff;
                        last_e = nmem > 0 ? smem[grp][nmem - 1].x : -1;
                    }
                    while (next_e < base_e) { ++mg; next_e = mg < nmem ? smem[grp][mg].x : 0x7fffffff; }
                    // the members among the 64 (group-uniform trips, usually none): entry base_e + 8 k + q is lane q's k-th
                    while (next_e < end_e) {
                        const int rel = next_e - base_e, p = smem[grp][mg].y;
                        const bool mine = (rel & (LT_L2_LANES - 1)) == q;
#pragma unroll
                        for (int k = 0; k < UN; ++k)
                            if (mine && (rel >> 3) == k) mp[k] = p;
                        ++mg;
                        next_e = mg < nmem ? smem[grp][mg].x : 0x7fffffff;
                    }
                }
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int ii = i + k * LT_L2_LANES;
                    if (ii >= nc) break;
                    const float a = a_[k];
                    if (mp[k] >= 0) {
                        touch = true;
                        const float *t = items + (size_t)mp[k] * C;
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) acc[c] = fmaf(a, t[c], acc[c]);
                    } else if (!DELTA) {
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) acc[c] = fmaf(a, tb[DELTA ? 0 : k][c], acc[c]);
                    }
                }
            }
        }
    }
    int t = touch ? 1 : 0;
#pragma unroll
    for (int m = LT_L2_LANES / 2; m >= 1; m >>= 1) t |= __shfl_xor(t, m, 64);
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = group_sum<LT_L2_LANES>(acc[c]);
    float res = 0.f;
    if (t) {
        if (DELTA) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) {
                    const float d_ = acc[c] / delta;
                    ss = fmaf(d_, d_, ss);
                }
            res = sqrtf(ss);
        } else {
            res = diff_norm<CP>(acc, b2, OUT + (size_t)u * C, C, delta);
        }
    }
    if (live && q == 0) {
        out[(long)b * ldo + j] = res;
        if (vec) store_diff_vec<CP>(vec, (long)b * ldo + j, C, acc, b2, DELTA ? (const float *)nullptr : OUT + (size_t)u * C, t != 0);
    }
    __syncthreads();        // (the next slot restages sc / sv / sT)
    }
}

// The observed hubs of SPARSE in a launch of their own: stageB_long_block's member lists take 48 KB of LDS per block, which
// as part of k_item_stageB capped every block of that launch -- the pairs too -- at three per CU (28 -> 55 us at twitch
// size, hubs or not).  DELTA's hub blocks need 8 KB and stay in front of the pair launch.
template <int CP, bool DELTA, bool SHORT, bool WIDE = false>
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageB_hubs(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, const float *__restrict__ OUT,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs, float delta,
    float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words,
    const uint2 *__restrict__ big_bits, const int32_t *__restrict__ big_slot, const int32_t *__restrict__ hub_obs,
    float *__restrict__ vec) {
    stageB_long_block<CP, DELTA, SHORT, WIDE>((int)blockIdx.x, rowptr, col, val, tptr, trow, S2, C, b2, OUT, probes, nb, off, S2x,
                                        observe, n_obs, delta, out, ldo, bits, words, big_bits, big_slot, hub_obs,
                                        (int)gridDim.x / ((nb + LT_BLOCK / LT_L2_LANES - 1) / (LT_BLOCK / LT_L2_LANES)), vec);
}

// ---- DELTA at twitch size: stage A and stage B of ONE probe in one block (round 4; "delta_fused") ----------------------------
// k_item_stageA_d2 + k_item_stageB_rows cost 12.8 + 13.4 us of a 65 us step: stage B tests every entry of every observed row
// against every probe -- 4.75 M (probe, entry) tests at n_test = 500 for the 7 % of the pairs a probe touches at all; those
// tests, not latency or cache lines, are what it costs (a per-probe block that still tested every entry took the same 18 us
// whether it made 24 or 6 dependent round trips and whether its loads were coalesced: ~20 instructions per test).
// Here the touched pairs are ENUMERATED from the probe's side instead, ~ 360 incidences per probe instead of 9 500 tests, and
// everything about them that does not depend on the layers is ready when this kernel starts: a node's incidence record --
// its items and, per touched node, the member entries in entry order -- is built with the graph (lt_core.hip
// build_delta_records), and delta_record_block (lt_items.hip.h), riding in the launch that forms the pre-activation, has
// matched it against the observed list: the table row of probe b names the observed POSITIONS the probe touches and where
// their entries are.  This kernel, one block per probe, is then two dependent round trips: the table row (items, touched
// positions), then the items' pre-activation rows with the positions' entries; stage A over the items (k_item_stageA_d2's
// arithmetic, statement by statement); and per touched position the sum row2_dot would form: entry k feeds chain k & 7, a chain is
// a k-ordered fmaf sequence from +0, the chains are added by the xor 4, 2, 1 butterfly, then d / delta and the fmaf sum of squares.
// So the matrix has the bits of the three-launch route (`delta_fused` = 0; tests/test_gpu_round4.py compares them).  Graphs
// with incidence records: no hub rows, n <= 65534, at most LT_DL_MAX_T incidences per node.
#ifdef LT_DF_TRACE      // tools/df_trace.py: phase stamps of every wave of k_delta_probe_finish on the constant 100 MHz clock
#define LT_DF_TRACE_BLOCKS 4096
__device__ unsigned long long g_df_trace[LT_DF_TRACE_BLOCKS * 4 * 8];
extern "C" int lt_debug_df_trace(unsigned long long *host_out, int n_words) {
    LT_HIP(hipDeviceSynchronize());
    LT_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_df_trace), (size_t)n_words * sizeof(unsigned long long)));
    return LT_OK;
}
#define DF_STAMP(k_)                                                                                                   \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < LT_DF_TRACE_BLOCKS)                                                \
            g_df_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k_)] = wall_clock64();                     \
    } while (0)
#else
#define DF_STAMP(k_)
#endif
#define LT_DF_U 4          // items in flight per lane group in stage A
#define LT_DF_LDS_MAX (144 * 1024)   // dynamic LDS the kernel may be given
template <int LPR, int CP, bool SX, bool ZF>
__global__ __launch_bounds__(LT_BLOCK) void k_delta_probe_finish(
    const double *__restrict__ Z1d, const double *__restrict__ S1d, const float *__restrict__ S1x,
    const double *__restrict__ crefv, const double *__restrict__ S1qs, const float *__restrict__ Z1x, int Hp,
    const float *__restrict__ W2p, int C, const int32_t *__restrict__ rec, int rec_words, int maxc,
    const int32_t *__restrict__ dl_src, int n_obs, float delta, float *__restrict__ out, long ldo,
    double *__restrict__ out64 = nullptr, long ld64 = 0, int out64_sparse = 0) {      // (rows [0, out64_sparse) of out64 hold zeros already)
    extern __shared__ __attribute__((aligned(16))) unsigned char df_smem[];
    float *sS2 = reinterpret_cast<float *>(df_smem);             // [maxc][C] the items' layer-2 differences
    // (launched with 256, 128 or 64 threads: a call of more probes than the chip holds 4-wave blocks for -- ~ 90 VGPRs, 5 waves per
    // SIMD, 1280 blocks -- takes narrower blocks, so that its probes still wait for their round trips side by side)
    constexpr int RPW = 64 / LPR;
    const int NT = (int)blockDim.x, WAVES = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    DF_STAMP(0);
    const int32_t *R = rec + (size_t)b * rec_words;
    const int2 *gItems = reinterpret_cast<const int2 *>(R + 4);
    const int2 *gTp = reinterpret_cast<const int2 *>(R + 4 + 2 * maxc);
    // ---- round trip 1: the header, and unseen (their counts are in it) the items, this thread's touched positions, this
    // wave's long position
    const int4 hdr = *reinterpret_cast<const int4 *>(R);
    constexpr int TPR = 2;                                    // short positions per thread held in registers (beyond: a later trip)
    int2 tp[TPR];
#pragma unroll
    for (int h = 0; h < TPR; ++h) tp[h] = gTp[min(tid + h * NT, n_obs - 1)];
    int2 ltp = gTp[max(n_obs - 1 - wid, 0)];
    float *orow = out + (long)b * ldo;
    for (int j = tid; j < n_obs; j += NT) orow[j] = 0.f;    // (the barrier below orders these before the positions' results)
    // rows [0, out64_sparse) of out64 were zero-filled over PCIe by blocks of the launch that formed the product rows: their
    // probes' blocks write the touched positions only -- 7 % of the row at twitch size
    double *const srow = (out64 && b < out64_sparse) ? out64 + (long)b * ld64 : (double *)nullptr;
    auto put = [&](int pos, float v) __attribute__((always_inline)) {
        orow[pos] = v;
        if (srow) srow[pos] = (double)v;
    };
    const int cnt = hdr.x, n_short = hdr.y & 0xffff, n_long = (int)((unsigned)hdr.y >> 16), v = hdr.z;
    const lt_df_inc *ent = reinterpret_cast<const lt_df_inc *>(dl_src + hdr.w);
    // per touched position: up to 4 entries by one thread (a select per (entry, chain) pair, nothing but registers)
    auto short_answer = [&](const lt_df_inc (&e)[4], const int c_) -> float {
        float a[4], tv[4][CP];
        int k[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            a[x] = e[x].a;
            k[x] = x < c_ ? (e[x].ik & (LT_L2_LANES - 1)) : -1;
            const float *t = sS2 + (size_t)(e[x].ik >> 16) * C;
#pragma unroll
            for (int c = 0; c < CP; ++c) tv[x][c] = c < C ? t[c] : 0.f;
        }
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float ch[LT_L2_LANES];
#pragma unroll
            for (int qq = 0; qq < LT_L2_LANES; ++qq) {
                float acc = 0.f;
#pragma unroll
                for (int x = 0; x < 4; ++x) acc = k[x] == qq ? fmaf(a[x], tv[x][c], acc) : acc;
                ch[qq] = acc;
            }
            // lane 0 of group_sum<8>: x += xor 4; x += xor 2; x += xor 1
            const float o = ((ch[0] + ch[4]) + (ch[2] + ch[6])) + ((ch[1] + ch[5]) + (ch[3] + ch[7]));
            if (c < C) {
                const float dd = o / delta;
                ss = fmaf(dd, dd, ss);
            }
        }
        return sqrtf(ss);
    };
    // a position of more than 4 entries by a wave: lane y holds entry y (64 at a time), the wave walks them in order handing
    // each to lane (c, qq) = (l >> 3, l & 7) of its chain; the 8-lane butterfly is row2_dot's own
    auto long_first = [&](const int2 lt_, const int y0) -> lt_df_inc {      // (the loads: before stage A, for the wave's first position)
        const int st = lt_.y & 0xffff, c_ = lt_.y >> 16;
        return ent[st + min(y0 + lane, c_ - 1)];
    };
    auto long_answer = [&](const int2 lt_, lt_df_inc mine) {
        const int st = lt_.y & 0xffff, c_ = lt_.y >> 16;
        const int qq = lane & (LT_L2_LANES - 1), cl = lane >> 3;
        float acc = 0.f;
        for (int y0 = 0; y0 < c_; y0 += 64) {
            if (y0 > 0) mine = ent[st + min(y0 + lane, c_ - 1)];
            float tv[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) tv[c] = c < C ? sS2[(size_t)(mine.ik >> 16) * C + c] : 0.f;
            const int qm = mine.ik & (LT_L2_LANES - 1);
            const bool valid = y0 + lane < c_;
            // the entries of this lane's chain, as a lane mask: walked in ascending order (a chain has ~ count / 8 of them)
            unsigned long long mk = 0ull;
#pragma unroll
            for (int q_ = 0; q_ < LT_L2_LANES; ++q_) {
                const unsigned long long bq = __ballot(valid && qm == q_);
                mk = qq == q_ ? bq : mk;
            }
            if (cl >= C) mk = 0ull;
            while (__ballot(mk != 0ull)) {                        // (wave-uniform trips: every lane stays a readable source)
                const bool on = mk != 0ull;
                const int y = on ? __ffsll((long long)mk) - 1 : 0;
                mk &= mk - 1ull;                                  // (0 stays 0)
                const float ay = __shfl(mine.a, y, 64);
                float ty = 0.f;
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    const float tc = __shfl(tv[c], y, 64);
                    ty = cl == c ? tc : ty;
                }
                acc = on ? fmaf(ay, ty, acc) : acc;
            }
        }
        const float o = group_sum<LT_L2_LANES>(acc);
        float ss = 0.f;
        for (int cc = 0; cc < C; ++cc) {
            const float oc = __shfl(o, cc * LT_L2_LANES, 64);
            const float dd = oc / delta;
            ss = fmaf(dd, dd, ss);
        }
        if (lane == 0) put(lt_.x, sqrtf(ss));
    };
    lt_df_inc te[TPR][4];
    lt_df_inc lmine = {0.f, 0};
    // ---- stage A: this probe's items (k_item_stageA_d2, statement by statement) ----
    {
        const int gl = lane & (LPR - 1);
        const bool active = 4 * gl < Hp;
        const int coff = active ? 4 * gl : 0;
        float w2[4][CP];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < CP; ++c) w2[k][c] = c < C ? W2p[(size_t)(coff + k) * C + c] : 0.f;
        constexpr int U = LT_DF_U;      // (wider first trips -- 8, 12 items per lane group -- measured slower: 5.3 -> 8.1 us to the barrier)
        const int stride = WAVES * RPW;
        // (the first trip's items were asked for before the header was back: the table row holds maxc item slots, the unused
        // ones naming row 0 -- loaded, never stored)
        int base = wid * RPW + lane / LPR;
        int2 itm0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) itm0[u] = gItems[min(base + u * stride, maxc - 1)];
        // ---- round trip 2 (with the items' rows below): the probe's S1 row, the positions' entries
        // delta * S1[v, coff + k]: the probe's S1 row off the fp64 product (fixed-point rows and / or deferred reference product)
        float ds[4];
        {
            double cr[4] = {0.0, 0.0, 0.0, 0.0};
            if (crefv != nullptr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) cr[k] = crefv[coff + k];
            }
            if constexpr (SX) {
                typedef int qx4 __attribute__((ext_vector_type(4)));
                const f32x4 sx = ld4(S1x + (size_t)v * Hp + coff);
                const double sq = S1qs[v];
#pragma unroll
                for (int k = 0; k < 4; ++k) ds[k] = delta * (float)((double)__builtin_bit_cast(qx4, sx)[k] * sq + cr[k]);
            } else {
                const f64x4 sd = *reinterpret_cast<const f64x4 *>(S1d + (size_t)v * Hp + coff);
#pragma unroll
                for (int k = 0; k < 4; ++k) ds[k] = delta * (crefv != nullptr ? (float)(sd[k] + cr[k]) : (float)sd[k]);
            }
        }
        bool first = true;
        do {
            int it[U];
            int2 itm[U];
            typename std::conditional<ZF, f32x4, f64x4>::type z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                it[u] = base + u * stride;
                itm[u] = itm0[u];
                const int r = itm[u].x;
                // (a slot without an item: every lane asks for the row's first 16 bytes -- one cache line instead of a row; a
                // branch around the load would make hipcc drain the queue in front of it, one round trip per item)
                const int co = it[u] < cnt ? coff : 0;
                if constexpr (ZF) z[u] = ld4(Z1x + (size_t)r * Hp + co);
                else z[u] = *reinterpret_cast<const f64x4 *>(Z1d + (size_t)r * Hp + co);
            }
            // (a probe of more items than one trip takes: the next trip's items travel with this trip's rows)
#pragma unroll
            for (int u = 0; u < U; ++u) itm0[u] = gItems[min(it[u] + U * stride, maxc - 1)];
            if (first) {
#pragma unroll
                for (int h = 0; h < TPR; ++h) {
                    const int st = tp[h].y & 0xffff, c_ = tp[h].y >> 16;
                    const bool have = tid + h * NT < n_short;
#pragma unroll
                    for (int x = 0; x < 4; ++x) te[h][x] = ent[have ? st + min(x, c_ - 1) : 0];
                }
                if (wid < n_long) lmine = long_first(ltp, 0);
            }
            first = false;
            // four items at a time: their lane-group sums run the butterfly in lockstep (one item after the other, each of its
            // steps waited for the one before: 6 dependent LDS-crossbar trips per item at LPR = 64); the addends of a sum and
            // their order are group_sum's
            constexpr int Q = 4;
            static_assert(U % Q == 0, "stage A works in quads");
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += Q) {
                if (__ballot(it[u0] < cnt) == 0ull) break;            // (wave-uniform; the later slots are further out still)
                float part[Q][CP];
#pragma unroll
                for (int uu = 0; uu < Q; ++uu) {
                    const int u = u0 + uu;
                    const float arv = __int_as_float(itm[u].y);
#pragma unroll
                    for (int c = 0; c < CP; ++c) part[uu][c] = 0.f;
                    if (active) {
                        float dh[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float dz = arv * ds[k];
                            const double zz = (double)z[u][k], z1 = zz + (double)dz;
                            dh[k] = zz > 0.0 ? (z1 > 0.0 ? dz : (float)(-zz)) : (z1 > 0.0 ? (float)z1 : 0.f);
                        }
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) {
                                float p = dh[0] * w2[0][c];
                                p = fmaf(dh[1], w2[1][c], p);
                                p = fmaf(dh[2], w2[2][c], p);
                                p = fmaf(dh[3], w2[3][c], p);
                                part[uu][c] = p;
                            }
                    }
                }
#pragma unroll
                for (int m = LPR / 2; m >= 1; m >>= 1)
#pragma unroll
                    for (int uu = 0; uu < Q; ++uu)
#pragma unroll
                        for (int c = 0; c < CP; ++c) part[uu][c] += __shfl_xor(part[uu][c], m, 64);
#pragma unroll
                for (int uu = 0; uu < Q; ++uu) {
                    const int u = u0 + uu;
                    if (it[u] < cnt && gl == 0) {
#pragma unroll
                        for (int c = 0; c < CP; ++c)
                            if (c < C) sS2[(size_t)it[u] * C + c] = part[uu][c];
                    }
                }
            }
            base += U * stride;
        } while (base < cnt);
    }
    DF_STAMP(1);
    __syncthreads();
    DF_STAMP(2);
    // ---- the touched positions ----
#pragma unroll
    for (int h = 0; h < TPR; ++h)
        if (tid + h * NT < n_short) put(tp[h].x, short_answer(te[h], tp[h].y >> 16));
    DF_STAMP(3);
    if (wid < n_long) long_answer(ltp, lmine);
    DF_STAMP(4);
    // (beyond the registers: a trip of their own each)
    for (int x = tid + TPR * NT; x < n_short; x += NT) {
        const int2 t_ = gTp[x];
        const int st = t_.y & 0xffff, c_ = t_.y >> 16;
        lt_df_inc e[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) e[y] = ent[st + min(y, c_ - 1)];
        put(t_.x, short_answer(e, c_));
    }
    // The long positions beyond each wave's first (a clique of k nodes among the probes is k of them per member): 8 lanes per
    // position, 8 positions per wave side by side.  Lane q of a group holds entries q, q + 8, ... of a 64-entry stretch and the
    // group walks the stretch in entry order, every entry handed round by an 8-lane shuffle to the lane of its chain: the
    // addends and their order are long_answer's (chain k & 7, entry order), the butterfly and the tail row2_dot's.
    // (one wave per position, as above, is ~ 1 us each: 56 of them took a block 14 us.)
    {
        const int q = lane & (LT_L2_LANES - 1);
        const int NG = NT / LT_L2_LANES;                            // groups per block
        // (wave-uniform trips: the groups of a wave whose position is past the end ride along unpredicated, stores excepted)
        for (int s0 = WAVES + wid * (64 / LT_L2_LANES); s0 < n_long; s0 += NG) {
            const int s_ = s0 + lane / LT_L2_LANES;
            const bool have = s_ < n_long;
            const int2 t_ = gTp[n_obs - 1 - min(s_, n_long - 1)];
            const int st = t_.y & 0xffff, c_ = have ? (t_.y >> 16) : 0;
            float acc[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[c] = 0.f;
            for (int y0 = 0; __ballot(y0 < c_) != 0ull; y0 += 64) {
                lt_df_inc e[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) e[m] = ent[st + min(y0 + q + 8 * m, max(c_ - 1, 0))];
                float tv[8][CP];
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int c = 0; c < CP; ++c) tv[m][c] = c < C ? sS2[(size_t)(e[m].ik >> 16) * C + c] : 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (__ballot(y0 + 8 * m < c_) == 0ull) break;          // (wave-uniform)
#pragma unroll
                    for (int src = 0; src < LT_L2_LANES; ++src) {
                        const float a_ = __shfl(e[m].a, src, LT_L2_LANES);
                        const int k_ = __shfl(e[m].ik, src, LT_L2_LANES);
                        const bool mine_ = y0 + 8 * m + src < c_ && (k_ & (LT_L2_LANES - 1)) == q;
#pragma unroll
                        for (int c = 0; c < CP; ++c) {
                            const float t = __shfl(tv[m][c], src, LT_L2_LANES);
                            acc[c] = mine_ ? fmaf(a_, t, acc[c]) : acc[c];
                        }
                    }
                }
            }
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                const float o = group_sum<LT_L2_LANES>(acc[c]);
                if (c < C) {
                    const float dd = o / delta;
                    ss = fmaf(dd, dd, ss);
                }
            }
            if (have && q == 0) put(t_.x, sqrtf(ss));
        }
    }
    DF_STAMP(5);
    // out64 != NULL (lt_influence_rows_f64): the block widens its own finished row into the caller's float64 matrix -- pinned host
    // memory through its device-side alias: the rows cross PCIe while the other probes' blocks still compute, instead of in a
    // launch of their own behind this one (lt_export_rows_f64).  The barrier drains every wave's stores (hipcc emits vmcnt(0) in
    // front of it); the row is read back past this CU's L1 (an earlier call's export may have left lines of it there).
    if (out64 && !srow) {
        __syncthreads();
        double *drow = out64 + (long)b * ld64;
        const bool pair_ok = (reinterpret_cast<uintptr_t>(drow) & 15) == 0;
        const int half = (n_obs + 1) >> 1;
        for (int k = tid; k < half; k += NT) {
            const int c = 2 * k;
            const float v0 = __builtin_nontemporal_load(orow + c);
            if (c + 1 < n_obs) {
                const float v1 = __builtin_nontemporal_load(orow + c + 1);
                if (pair_ok) *reinterpret_cast<double2 *>(drow + c) = make_double2((double)v0, (double)v1);
                else { drow[c] = (double)v0; drow[c + 1] = (double)v1; }
            } else drow[c] = (double)v0;
        }
    }
}

// beyond the default 64 KB of dynamic LDS a kernel is told, once, that it may take most of a CU's 160 KB
template <int LPR, int CP, bool SX, bool ZF>
static int df_allow_big_lds() {
    // (the attribute belongs to the device's copy of the code object: remembered per device, not per process)
    static unsigned long long done = 0ull;
    int dev = 0;
    LT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
        LT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_delta_probe_finish<LPR, CP, SX, ZF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LT_DF_LDS_MAX));
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
    return LT_OK;
}

// one (probe, observed node) pair of SPARSE / DELTA stage B by its 8-lane group (k_item_stageB: gid from the grid; k_item_stageB_list:
// from the list of marked pairs)
template <int CP, bool DELTA>
__device__ __forceinline__ void item_stageB_pair(const long gid, const int q,
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, const float *__restrict__ OUT,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs, float delta,
    float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words, int skip_long,
    const unsigned *__restrict__ marks, const uint2 *__restrict__ big_bits, const int32_t *__restrict__ big_slot,
    float *__restrict__ vec) {
    const int b = (int)(gid / n_obs), j = (int)(gid % n_obs);
    const int u = observe[j];
    const int v = probes[b];
    const int32_t *rv = trow + tptr[v];
    const int cnt = tptr[v + 1] - tptr[v];
    const float *items = S2x + (size_t)off[b] * C;
    const int e0 = rowptr[u], e1 = rowptr[u + 1];
    if (skip_long && e1 - e0 > LT_ROW_SEG) return;   // an observed hub: one of the first blocks

    // does row u touch R_v at all?  (otherwise the perturbed logits ARE the baseline logits)
    const uint2 *mb = bits ? bits + (size_t)b * words
                           : ((big_slot && big_slot[b] >= 0) ? big_bits + (size_t)big_slot[b] * words : nullptr);
    // position of column c in R_v (-1: not a member): the bitmap when there is one, the search otherwise
    auto pos = [&](int c) { return mb ? bits_pos(mb, c) : find_row(rv, cnt, c); };
    int t;
    if (marks) {   // the join over the middle nodes has answered it (k_pm_mark): one bit per pair
        t = (marks[gid >> 5] >> (gid & 31)) & 1u;
    } else {
        bool touch = false;
        for (int e = e0 + q; e < e1; e += LT_L2_LANES) touch |= pos(col[e]) >= 0;
        // 8-lane any(): xor butterfly on an int
        t = touch ? 1 : 0;
#pragma unroll
        for (int m = LT_L2_LANES / 2; m >= 1; m >>= 1) t |= __shfl_xor(t, m, 64);
    }
    float res = 0.f;
    float acc[CP];
    if (t) {  // group-uniform
        if (DELTA) {
            // d_out[c] = sum over e with col[e] in R_v of val[e] * dS2[item(col[e]), c]
            row2_dot<CP>(col, val, e0, e1, q, C,
                         [&](int c, int) {
                             const int p = pos(c);
                             return p >= 0 ? items + (size_t)p * C : (const float *)nullptr;
                         },
                         acc);
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) {
                    const float d = acc[c] / delta;
                    ss = fmaf(d, d, ss);
                }
            res = sqrtf(ss);
        } else {
            row2_dot<CP>(col, val, e0, e1, q, C,
                         [&](int c, int) {
                             const int p = pos(c);
                             return p >= 0 ? items + (size_t)p * C : S2 + (size_t)c * C;
                         },
                         acc);
            res = diff_norm<CP>(acc, b2, OUT + (size_t)u * C, C, delta);
        }
    }
    if (q == 0) {
        out[(long)b * ldo + j] = res;
        if (vec) store_diff_vec<CP>(vec, (long)b * ldo + j, C, acc, b2, DELTA ? (const float *)nullptr : OUT + (size_t)u * C, t != 0);
    }
}


// SPARSE / DELTA stage B: 8 lanes per (probe, observed node).
template <int CP, bool DELTA>
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageB(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, const float *__restrict__ OUT,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs, float delta,
    float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words, int long_blocks, int skip_long,
    const unsigned *__restrict__ marks, const uint2 *__restrict__ big_bits, const int32_t *__restrict__ big_slot,
    int hub_short, const int32_t *__restrict__ hub_obs, float *__restrict__ vec) {
    // long_blocks > 0 (the graph has hub rows): the first long_blocks blocks serve the observed hubs
    // (stageB_long_block: most of them find a plain row and exit); skip_long: the pairs below leave those rows alone
    // (SPARSE with the short-side search: k_item_stageB_hubs has them, long_blocks = 0 here)
    // (DELTA: always k_item_stageB_hubs -- its hub block keeps 32 membership tests in flight per lane, 211 VGPRs, which as
    // part of this kernel cost the pairs of a large call their occupancy: 0.88 -> 1.25 ms at BASELINE configs[4])
    if constexpr (!DELTA) {
        if ((int)blockIdx.x < long_blocks) {
            stageB_long_block<CP, DELTA, false>((int)blockIdx.x, rowptr, col, val, tptr, trow, S2, C, b2, OUT, probes, nb, off,
                                                S2x, observe, n_obs, delta, out, ldo, bits, words, big_bits, big_slot, hub_obs,
                                                long_blocks / ((nb + LT_BLOCK / LT_L2_LANES - 1) / (LT_BLOCK / LT_L2_LANES)), vec);
            return;
        }
    }
    const long gid = ((long)(blockIdx.x - long_blocks) * LT_BLOCK + threadIdx.x) / LT_L2_LANES;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    if (gid >= (long)nb * n_obs) return;
    item_stageB_pair<CP, DELTA>(gid, q, rowptr, col, val, tptr, trow, S2, C, b2, OUT, probes, nb, off, S2x, observe, n_obs, delta, out, ldo,
                                bits, words, skip_long, marks, big_bits, big_slot, vec);
}

// The same over a LIST of pairs (round 6): calls that find their affected pairs by the join over the middle nodes (pair marks) leave
// 90+ % of their pairs unmarked, and a wave of k_item_stageB holds 8 pairs of which one marked pair makes the whole wave wait for its
// chains -- 44 % of the waves of a `balanced-full` chunk, 342 us.  k_pm_compact turns the marks into the list of marked pairs, `out`
// is zero-filled by a memset, and the marked pairs are walked 8 to a wave: the same per-pair code, the same bits.
static __global__ __launch_bounds__(256) void k_pm_compact(const unsigned *__restrict__ marks, long n_words, int32_t *__restrict__ pairs,
                                                           int32_t *__restrict__ n_pairs) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (long w0 = (long)blockIdx.x * 256; w0 < n_words; w0 += (long)gridDim.x * 256) {       // (block-uniform)
        const long w = w0 + threadIdx.x;
        unsigned m = w < n_words ? marks[w] : 0u;
        const int cnt = __popc(m);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wid] = incl;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            before += k < wid ? s_wave[k] : 0;
            all += s_wave[k];
        }
        if (all > 0 && threadIdx.x == 0) s_base = atomicAdd(n_pairs, all);
        __syncthreads();
        if (cnt > 0) {
            int pos = s_base + before + incl - cnt;
            while (m) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1u;
                pairs[pos++] = (int32_t)(w * 32 + bit);
            }
        }
        __syncthreads();
    }
}
template <int CP, bool DELTA>
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageB_list(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int32_t *__restrict__ tptr,
    const int32_t *__restrict__ trow, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, const float *__restrict__ OUT,
    const int32_t *__restrict__ probes, int nb, const int32_t *__restrict__ off,
    const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs, float delta,
    float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words, int skip_long,
    const unsigned *__restrict__ marks, const uint2 *__restrict__ big_bits, const int32_t *__restrict__ big_slot,
    float *__restrict__ vec, const int32_t *__restrict__ pairs, const int32_t *__restrict__ n_pairs) {
    const int total = *n_pairs;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    const int groups = (int)gridDim.x * (LT_BLOCK / LT_L2_LANES);
    for (int idx = ((int)blockIdx.x * LT_BLOCK + threadIdx.x) / LT_L2_LANES; idx < total; idx += groups)
        item_stageB_pair<CP, DELTA>((long)pairs[idx], q, rowptr, col, val, tptr, trow, S2, C, b2, OUT, probes, nb, off, S2x, observe, n_obs,
                                    delta, out, ldo, bits, words, skip_long, marks, big_bits, big_slot, vec);
}

// SPARSE / DELTA stage B, calls with a bitmap row per probe (twitch size): one block per (observed node, slice of the
// probes) instead of one 8-lane group per pair.  k_item_stageB pays the chain observe[j] -> rowptr[u] -> col[e] -> bitmap
// for every one of the 250 K pairs of a 500 x 500 call although 93 % of them turn out untouched; here the observed row
// is staged in LDS once and each 8-lane group then tests LT_SB_UNR probes at a time against it (independent bitmap loads,
// the only global traffic of an untouched pair).  The touched pairs run row2_dot over the staged row: same chains
// (entry e -> chain (e - e0) & 7, k-ordered), same butterfly, same tail -- the bits of k_item_stageB.
#define LT_SB_UNR 4
#define LT_SB_SHORT 32    // rows up to this many entries test their probes one after the other (DELTA)
#define LT_SB_EB 4        // DELTA, longer rows: entries per lane whose tests and items are in flight together
#define LT_SB_PASS 2048   // probes per pass of a block (its touched pairs are listed in LDS)
template <int CP, bool DELTA>
__global__ __launch_bounds__(LT_BLOCK) void k_item_stageB_rows(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ S2, int C, const float *__restrict__ b2, const float *__restrict__ OUT, int nb,
    const int32_t *__restrict__ off, const float *__restrict__ S2x, const int32_t *__restrict__ observe, int n_obs,
    float delta, float *__restrict__ out, long ldo, const uint2 *__restrict__ bits, int words, int psplit,
    int long_blocks = 0, const int32_t *__restrict__ tptr = nullptr, const int32_t *__restrict__ trow = nullptr,
    const int32_t *__restrict__ probes = nullptr, const int32_t *__restrict__ hub_obs = nullptr) {
    // DELTA, long_blocks > 0 (a graph with hub rows): the FIRST long_blocks blocks serve the observed hubs (stageB_long_block, every
    // probe walking the staged row through its bitmap: 76 VGPRs against this kernel's 90, 8 KB of LDS) -- the two kernels are
    // independent and each is one generation of latency-bound blocks: side by side in one launch, not one after the other
    // (15.0 + 19.7 us at twitch size with a power-law graph; a second stream for the hubs cost more than it hid)
    int bx = (int)blockIdx.x;
    if constexpr (DELTA) {
        if (bx < long_blocks) {
            stageB_long_block<CP, true, false, false>(bx, rowptr, col, val, tptr, trow, S2, C, b2, OUT, probes, nb, off, S2x, observe,
                                                      n_obs, delta, out, ldo, bits, words, (const uint2 *)nullptr,
                                                      (const int32_t *)nullptr, hub_obs,
                                                      long_blocks / ((nb + LT_BLOCK / LT_L2_LANES - 1) / (LT_BLOCK / LT_L2_LANES)));
            return;
        }
        bx -= long_blocks;
    }
    __shared__ int32_t scol[LT_ROW_SEG];
    __shared__ float sval[LT_ROW_SEG];
    const int j = bx / psplit, part = bx % psplit;
    const int u = observe[j];
    const int e0 = rowptr[u], d = rowptr[u + 1] - e0;
    if (d > LT_ROW_SEG) return;          // an observed hub: stageB_long_block (the hub blocks of k_item_stageB's launch)
    for (int i = threadIdx.x; i < d; i += LT_BLOCK) { scol[i] = col[e0 + i]; sval[i] = val[e0 + i]; }
    __syncthreads();
    const int q = threadIdx.x & (LT_L2_LANES - 1), grp = threadIdx.x / LT_L2_LANES;
    constexpr int GROUPS = LT_BLOCK / LT_L2_LANES;
    const int per = (nb + psplit - 1) / psplit;
    const int b_begin = part * per, b_end = min(nb, b_begin + per);
    const int32_t *lc = scol - e0;       // row2_dot indexes its arrays with the CSR entry number
    const float *lv = sval - e0;
    // Two phases per pass of up to LT_SB_PASS probes: (1) every group tests LT_SB_UNR probes at a time against the staged row
    // (independent bitmap loads) -- untouched pairs get their 0 at once, touched ones go onto a block-wide LDS list; (2) the
    // groups take the listed pairs round-robin, so the few touched pairs of a block (7 % at twitch size) run side by side
    // instead of one after the other inside the group that found them.  The list order is arbitrary: pairs are independent.
    if constexpr (DELTA) {
        // DELTA: only the members of R_v in the row contribute, and the bitmap word that answers "is this column a member"
        // also gives its position in R_v -- so the lane that tests an entry (lane q tests exactly the entries of chain q)
        // adds the member's term at once: no second phase, no look-up.  Same FMAs in the same per-chain order, then the same
        // 8-lane butterfly as row2_dot: the bits of k_item_stageB.
        for (int b0 = b_begin + grp; b0 < b_end; b0 += GROUPS * LT_SB_UNR) {
            float acc[LT_SB_UNR][CP];
            int t[LT_SB_UNR];
            const uint2 *mb[LT_SB_UNR];
            const float *items[LT_SB_UNR];
#pragma unroll
            for (int k = 0; k < LT_SB_UNR; ++k) {
                const int b = min(b0 + k * GROUPS, b_end - 1);   // past the end: the last probe again, never stored
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[k][c] = 0.f;
                t[k] = 0;
                mb[k] = bits + (size_t)b * words;
                items[k] = S2x + (size_t)off[b] * C;
            }
            if (d <= LT_SB_SHORT) {
                // short rows (a lane holds one to four entries): entry by entry (the batched form below on them too: 19.8 -> 27.6 us on
                // the power-law graph)
                // (round 5: the membership words of the group's LT_SB_UNR probes for an entry go out TOGETHER -- one trip instead of one
                // per probe: 0.0735 -> 0.0719 ms for the power-law step; the members' items stay behind their branch, they are rare.  Per
                // probe the entries come in the same order: same bits.  The lane's up to four entries in one trip as well: slower, 0.081.)
                for (int e = q; e < d; e += LT_L2_LANES) {
                    const int c = scol[e];
                    const unsigned bit = 1u << (c & 31);
                    const float a = sval[e];
                    uint2 w[LT_SB_UNR];
#pragma unroll
                    for (int k = 0; k < LT_SB_UNR; ++k) w[k] = mb[k][c >> 5];
#pragma unroll
                    for (int k = 0; k < LT_SB_UNR; ++k)
                        if (w[k].x & bit) {
                            const float *it = items[k] + (size_t)(w[k].y + __popc(w[k].x & (bit - 1u))) * C;
#pragma unroll
                            for (int cc = 0; cc < CP; ++cc)
                                if (cc < C) acc[k][cc] = fmaf(a, it[cc], acc[k][cc]);
                            t[k] = 1;
                        }
                }
            } else {
                // longer rows (a lane holds up to 16 entries of its chain): LT_SB_EB of them x LT_SB_UNR probes per trip -- all
                // their membership words first, then all the members' items (a non-member asks for its probe's first item:
                // a load behind a branch would wait for every load in front of it), then the FMAs in entry order.  A row of
                // 128 entries is 4 + 4 round trips per probe group instead of 16 + one per member.
                for (int e0_ = q; e0_ < d; e0_ += LT_L2_LANES * LT_SB_EB) {
                    float a[LT_SB_EB];
                    unsigned bit[LT_SB_EB];
                    uint2 w[LT_SB_EB][LT_SB_UNR];
#pragma unroll
                    for (int x = 0; x < LT_SB_EB; ++x) {
                        const int e = e0_ + x * LT_L2_LANES;
                        const int c = scol[e < d ? e : q];
                        a[x] = sval[e < d ? e : q];
                        bit[x] = e < d ? 1u << (c & 31) : 0u;            // (past the end: no bit, never a member)
#pragma unroll
                        for (int k = 0; k < LT_SB_UNR; ++k) w[x][k] = mb[k][c >> 5];
                    }
                    float iv[LT_SB_EB][LT_SB_UNR][CP];
#pragma unroll
                    for (int x = 0; x < LT_SB_EB; ++x)
#pragma unroll
                        for (int k = 0; k < LT_SB_UNR; ++k) {
                            const bool hit = (w[x][k].x & bit[x]) != 0u;
                            const float *it = items[k] + (hit ? (size_t)(w[x][k].y + __popc(w[x][k].x & (bit[x] - 1u))) * C : (size_t)0);
#pragma unroll
                            for (int cc = 0; cc < CP; ++cc) iv[x][k][cc] = cc < C ? it[cc] : 0.f;
                        }
#pragma unroll
                    for (int x = 0; x < LT_SB_EB; ++x)
#pragma unroll
                        for (int k = 0; k < LT_SB_UNR; ++k)
                            if (w[x][k].x & bit[x]) {
#pragma unroll
                                for (int cc = 0; cc < CP; ++cc)
                                    if (cc < C) acc[k][cc] = fmaf(a[x], iv[x][k][cc], acc[k][cc]);
                                t[k] = 1;
                            }
                }
            }
#pragma unroll
            for (int k = 0; k < LT_SB_UNR; ++k)
#pragma unroll
                for (int m = LT_L2_LANES / 2; m >= 1; m >>= 1) t[k] |= __shfl_xor(t[k], m, 64);
#pragma unroll
            for (int k = 0; k < LT_SB_UNR; ++k) {
                const int b = b0 + k * GROUPS;
                if (b >= b_end) continue;     // group-uniform
                float res = 0.f;
                if (t[k]) {                   // group-uniform
                    float ss = 0.f;
#pragma unroll
                    for (int cc = 0; cc < CP; ++cc) {
                        const float o = group_sum<LT_L2_LANES>(acc[k][cc]);
                        if (cc < C) {
                            const float dd = o / delta;
                            ss = fmaf(dd, dd, ss);
                        }
                    }
                    res = sqrtf(ss);
                }
                if (q == 0) out[(long)b * ldo + j] = res;
            }
        }
        return;
    }
    __shared__ int32_t s_list[LT_SB_PASS];
    __shared__ int32_t s_cnt;
    for (int p0 = b_begin; p0 < b_end; p0 += LT_SB_PASS) {
        const int p1 = min(b_end, p0 + LT_SB_PASS);
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (int b0 = p0 + grp; b0 < p1; b0 += GROUPS * LT_SB_UNR) {
            int t[LT_SB_UNR];
#pragma unroll
            for (int k = 0; k < LT_SB_UNR; ++k) {
                const int b = b0 + k * GROUPS;
                unsigned hit = 0u;
                if (b < p1) {
                    const uint2 *mb = bits + (size_t)b * words;
                    for (int e = q; e < d; e += LT_L2_LANES) {
                        const int c = scol[e];
                        hit |= (mb[c >> 5].x >> (c & 31)) & 1u;
                    }
                }
                t[k] = (int)hit;
            }
#pragma unroll
            for (int k = 0; k < LT_SB_UNR; ++k)
#pragma unroll
                for (int m = LT_L2_LANES / 2; m >= 1; m >>= 1) t[k] |= __shfl_xor(t[k], m, 64);
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < LT_SB_UNR; ++k) {
                    const int b = b0 + k * GROUPS;
                    if (b >= p1) continue;
                    if (t[k]) s_list[atomicAdd(&s_cnt, 1)] = b;
                    else out[(long)b * ldo + j] = 0.f;
                }
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int idx = grp; idx < cnt; idx += GROUPS) {
            const int b = s_list[idx];
            const uint2 *mb = bits + (size_t)b * words;
            const float *items = S2x + (size_t)off[b] * C;
            float acc[CP];
            float res;
            if (DELTA) {
                row2_dot<CP>(lc, lv, e0, e0 + d, q, C,
                             [&](int c, int) {
                                 const int p = bits_pos(mb, c);
                                 return p >= 0 ? items + (size_t)p * C : (const float *)nullptr;
                             },
                             acc);
                float ss = 0.f;
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) {
                        const float dd = acc[c] / delta;
                        ss = fmaf(dd, dd, ss);
                    }
                res = sqrtf(ss);
            } else {
                row2_dot<CP>(lc, lv, e0, e0 + d, q, C,
                             [&](int c, int) {
                                 const int p = bits_pos(mb, c);
                                 return p >= 0 ? items + (size_t)p * C : S2 + (size_t)c * C;
                             },
                             acc);
                res = diff_norm<CP>(acc, b2, OUT + (size_t)u * C, C, delta);
            }
            if (q == 0) out[(long)b * ldo + j] = res;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Probes per wave of the wide FULL kernel (8, 16 or 32): a tuning knob only -- results are bit-identical.
// Measured cost of a (row, group) wave ~ 3.4 + 0.63 * P (arbitrary units; twitch-RU, 500 probes: P = 8 /
// 16 / 32 -> 528 / 396 / 376 us), so the choice minimises ceil(nb / P) * (3.4 + 0.63 P); lt_set_tuning("full_p") pins it.
static int full_probes_per_wave(int nb) {
    if (lt_tune().full_p) return lt_tune().full_p;
    int best = 16;
    double best_cost = 1e30;
    for (int P = 8; P <= 32; P *= 2) {
        const double cost = (double)((nb + P - 1) / P) * (3.4 + 0.63 * P);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = P; }
    }
    return best;
}

// "overlap" = 0 keeps the hub-row kernels on the caller's stream (A/B knob)
static bool overlap_enabled() { return lt_tune().overlap != 0; }

// bytes of per-probe scratch per chunk ("chunk_budget_bytes": tests force multi-chunk calls)
static size_t chunk_budget() { return (size_t)lt_tune().chunk_budget; }

// measured on twitch-RU, 500 probes: H = 16: 158 us (narrow) vs 257; H = 32: 359 vs 264
static int wide_min_hp() { return lt_tune().wide_min_hp; }

// Long rows go segment-parallel when the segment sums of ONE probe fit LT_LONG_PAR_BYTES (a graph with a few
// hubs); a graph where they do not (10^5 long rows) walks each long row in one wave per <= 16 probes.
// "long_par" pins the choice (tests run both paths on small graphs).
#define LT_LONG_PAR_BYTES ((size_t)4 << 20)
static bool long_rows_parallel(const lt_graph *g, int Hp) {
    if (lt_tune().long_par >= 0) return lt_tune().long_par == 1;
    return (size_t)g->p_n_seg * Hp * sizeof(float) * 9 / 8 <= LT_LONG_PAR_BYTES;
}

// K-slice of the perturbed-row GEMM.  "probe_kslice" = 0 (default): the slicing of the baseline X*W1, so that a probe
// row S1'[v] = (x_v + x_v d) W1 and its baseline S1[v] are summed in the SAME order and their rounding errors largely
// cancel in the finite difference -- as they do in the reference, whose two torch.mm calls share one summation order;
// > 0: that many columns per slice (256 was the round-1 setting: more workgroups for few probes, uncorrelated rounding).
static int probe_kslice(const lt_baseline *b) {
    const int k = lt_tune().probe_kslice;
    return k > 0 ? (k + 15) / 16 * 16 : lt_gemm_pick_kslice(b->n, b->H, b->F);
}

// the fused DELTA route's sizes (delta_record_block + k_delta_probe_finish): whether the graph qualifies at all
struct df_geom { int maxc, rec_words; size_t record_smem, finish_smem; bool ok; };
static df_geom df_geometry(const lt_graph *g, int C, int n_obs) {
    df_geom d = {};
    d.maxc = g->max_col_nnz > 0 ? g->max_col_nnz : 1;
    d.rec_words = lt_dl_rec_words(d.maxc, n_obs);
    d.record_smem = ((size_t)g->dl_max_tu + 1) * sizeof(int2);
    d.finish_smem = (size_t)d.maxc * C * sizeof(float) + 16;
    d.ok = g->dl_meta != nullptr && n_obs >= 1 && n_obs <= 65534 && d.finish_smem <= (size_t)LT_DF_LDS_MAX &&
           d.record_smem <= (size_t)64 * 1024;
    return d;
}

struct infl_ws {
    float *Sp, *S2p;       // FULL / SPARSE: S1 rows of the perturbed probes; FULL: per-probe S2
    float *lpart;          // FULL: segment sums of the long rows [segment][group][P + 1][Hp]
    unsigned *lhit;        // FULL: which probes of a (segment, group) wrote a slot of their own
    int32_t *hub_obs;      // SPARSE / DELTA: [0] = observed hubs, [1 ...] = their positions in observe_nodes
    float *slabs;          // FULL / SPARSE: split-K partials of the perturbed-row GEMM
    float *S2x;            // SPARSE / DELTA: per-item values
    double *Spd;           // DELTA, aggregate-first route: fp64 product rows of the chunk's probes [chunk, Hp]
    int32_t *dl_rec;       // DELTA, fused route: the probes' incidence records [chunk][rec_words] (delta_lists_block)
    int32_t *off;          // SPARSE / DELTA: item offsets [chunk + 1]
    int2 *item_pr;         // SPARSE / DELTA: (probe index, row) of every item
    int2 *item_va;         // DELTA: (probe node, A_hat[row, probe node] as bits) of every item
    int32_t *pm_cnt, *pm_start, *pm_rank, *pm_list;   // SPARSE / DELTA pair marks: per-node lists of observed nodes (lt_items.hip.h)
    unsigned *pm_marks;    // SPARSE / DELTA: one bit per (probe of the chunk, observed node)
    int32_t *pm_pairs, *pm_npairs;   // the marked pairs of a chunk as a list (k_pm_compact) + its length; NULL: beyond LT_PM_LIST_MAX_BYTES
    uint2 *big_bits;       // SPARSE / DELTA without `bits`: bitmap rows of the chunk's big probes [LT_BIG_SLOTS][ceil(n / 32)]
    int32_t *big_slot;     // [chunk + 1] slot of each probe (-1: none) + the slot counter
    uint2 *bits;           // SPARSE / DELTA: membership bitmap + positions of R_v per probe [chunk][ceil(n / 32)], or NULL (huge graphs)
    int32_t *probes_s, *obs_s;   // the call's lists with every id checked against [0, n) (lt_items.hip.h checked_node) [n_probe] / [n_obs]
    size_t bytes;
    int chunk;
};

static infl_ws carve_infl(void *base, const lt_baseline *b, int n_probe, int n_obs, int mode) {
    infl_ws w = {};
    const size_t n = (size_t)b->n, C = (size_t)b->C, Hp = (size_t)b->Hp, F = (size_t)b->F;
    const size_t maxc = (size_t)(b->g->max_col_nnz > 0 ? b->g->max_col_nnz : 1);
    size_t per_probe = 0;
    const size_t splitk = ((F + probe_kslice(b) - 1) / probe_kslice(b)) * (size_t)b->H;
    const size_t nseg = long_rows_parallel(b->g, b->Hp) ? (size_t)b->g->p_n_seg : 0;
    if (mode == LT_MODE_FULL) per_probe = (n * C + Hp + splitk) * sizeof(float) + nseg * Hp * sizeof(float) * 9 / 8;
    else if (mode == LT_MODE_SPARSE) per_probe = (maxc * C + Hp + splitk) * sizeof(float) + sizeof(int32_t) + maxc * sizeof(int2);
    else per_probe = maxc * C * sizeof(float) + sizeof(int32_t) + maxc * sizeof(int2);
    size_t chunk = chunk_budget() / (per_probe ? per_probe : 1);
    if (chunk < 1) chunk = 1;
    if (chunk > 65534) chunk = 65534;  // grid.y (FULL, narrow kernel: probes + 1)
    if (chunk > (size_t)(n_probe > 0 ? n_probe : 1)) chunk = (size_t)(n_probe > 0 ? n_probe : 1);
    w.chunk = (int)chunk;
    size_t offb = 0;
    char *p = (char *)base;
    auto take = [&](size_t bytes) {
        void *q = p ? (void *)(p + offb) : nullptr;
        offb += lt_align_up(bytes ? bytes : 1, 256);
        return q;
    };
    w.probes_s = (int32_t *)take((size_t)(n_probe > 0 ? n_probe : 1) * sizeof(int32_t));
    w.obs_s = (int32_t *)take((size_t)(n_obs > 0 ? n_obs : 1) * sizeof(int32_t));
    if (mode == LT_MODE_FULL || mode == LT_MODE_SPARSE) {
        w.Sp = (float *)take(chunk * Hp * sizeof(float));
        w.slabs = (float *)take(lt_gemm_splitk_slab_bytes((int)chunk, b->H, b->F, probe_kslice(b)));
    }
    if (mode == LT_MODE_FULL) {
        w.S2p = (float *)take((chunk + 1) * n * C * sizeof(float));   // + the baseline column
        // groups * (P + 1) slots per segment: at most ceil(chunk / 8) * 9 (P = 8), or chunk / 32 * 33 + 33
        w.lpart = (float *)take(nseg * (((chunk + 7) / 8) * 9 + 33) * Hp * sizeof(float));
        w.lhit = (unsigned *)take(nseg * ((chunk + 7) / 8 + 1) * sizeof(unsigned));   // per (segment, group): probes with a slot of their own
    }
    if (mode == LT_MODE_DELTA) w.Spd = (double *)take(chunk * Hp * sizeof(double));   // aggregate-first: X[probes] W1 in fp64
    {   // the fused route's records (k_delta_probe_finish), when the graph qualifies and they stay modest
        const df_geom dg = df_geometry(b->g, (int)C, n_obs);
        if (mode == LT_MODE_DELTA && dg.ok && chunk * (size_t)dg.rec_words * sizeof(int32_t) <= ((size_t)1 << 30))
            w.dl_rec = (int32_t *)take(chunk * (size_t)dg.rec_words * sizeof(int32_t));
    }
    if (mode != LT_MODE_FULL) {
        w.hub_obs = (int32_t *)take(((size_t)n_obs + 1) * sizeof(int32_t));   // the observed nodes that are hub rows (k_item_bits)
        w.S2x = (float *)take(chunk * maxc * C * sizeof(float));
        w.off = (int32_t *)take((chunk + 1) * sizeof(int32_t));
        w.item_pr = (int2 *)take(chunk * maxc * sizeof(int2));
        if (mode == LT_MODE_DELTA) w.item_va = (int2 *)take(chunk * maxc * sizeof(int2));
        const size_t bw = (n + 31) / 32;
        // ("item_bits" = 0 forces the search path a huge graph takes: tests)
        const bool no_bits = lt_tune().item_bits == 0;
        w.bits = (!no_bits && (long long)(chunk * bw * sizeof(uint2)) <= lt_tune().bits_max_bytes) ? (uint2 *)take(chunk * bw * sizeof(uint2)) : nullptr;
        if (!w.bits && !no_bits) {
            w.big_bits = (uint2 *)take((size_t)LT_BIG_SLOTS * bw * sizeof(uint2));
            w.big_slot = (int32_t *)take((chunk + 1) * sizeof(int32_t));
        }
        if (lt_tune().pair_marks >= 0) {
            const size_t slots = (size_t)(n_obs > 0 ? n_obs : 1) * LT_ROW_SEG;
            w.pm_cnt = (int32_t *)take((n + 1) * sizeof(int32_t));     // [n] counts + the list cursor
            w.pm_start = (int32_t *)take(n * sizeof(int32_t));
            w.pm_rank = (int32_t *)take(slots * sizeof(int32_t));
            w.pm_list = (int32_t *)take(slots * sizeof(int32_t));
            w.pm_marks = (unsigned *)take((chunk * (size_t)(n_obs > 0 ? n_obs : 1) + 31) / 32 * sizeof(unsigned));
            // (the list of marked pairs, 4 bytes per pair of a chunk in the worst case: kept to calls where that stays modest)
            const size_t pl = chunk * (size_t)(n_obs > 0 ? n_obs : 1) * sizeof(int32_t);
            if (lt_tune().pair_list != 0 && pl <= ((size_t)256 << 20) && chunk * (size_t)(n_obs > 0 ? n_obs : 1) < 2147483647ull) {
                w.pm_npairs = (int32_t *)take(sizeof(int32_t));
                w.pm_pairs = (int32_t *)take(pl);
            }
        }
    }
    w.bytes = offb;
    return w;
}

extern "C" size_t lt_influence_workspace_bytes(const lt_baseline *b, int32_t n_probe, int32_t n_obs,
                                               int32_t mode) {
    if (!b || n_probe < 0 || n_obs < 0 || mode < LT_MODE_FULL || mode > LT_MODE_DELTA) return 0;
    return carve_infl(nullptr, b, n_probe, n_obs, mode).bytes;
}

static int influence_rows_impl(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                               const int32_t *observe_nodes, int32_t n_obs, float delta,
                               int32_t mode, float *out, int64_t ldo, void *workspace,
                               size_t workspace_bytes, void *stream, float *vec, double *dst64 = nullptr, int64_t ldd = 0);

// lt_influence_rows + the finished rows as float64 in dst (device memory, or pinned host memory: the reference's influence_val,
// attacker.py:216-229): the fused DELTA route's blocks write their own rows there, every other route ends with the launch of
// lt_export_rows_f64.
extern "C" int lt_influence_rows_f64(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                                     const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode, float *out,
                                     int64_t ldo, double *dst, int64_t ldd, void *workspace, size_t workspace_bytes, void *stream) {
    LT_REQUIRE(dst != nullptr || n_probe == 0 || n_obs == 0, "lt_influence_rows_f64: dst is NULL");
    LT_REQUIRE(ldd >= n_obs, "lt_influence_rows_f64: ldd smaller than the row");
    double *dev = nullptr;
    if (n_probe > 0 && n_obs > 0) {
        const int rc = lt_export_resolve(dst, &dev, "lt_influence_rows_f64");
        if (rc) return rc;
    }
    return influence_rows_impl(b, probe_nodes, n_probe, observe_nodes, n_obs, delta, mode, out, ldo, workspace, workspace_bytes,
                               stream, nullptr, dev, ldd);
}

extern "C" int lt_influence_rows(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                                 const int32_t *observe_nodes, int32_t n_obs, float delta,
                                 int32_t mode, float *out, int64_t ldo, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return influence_rows_impl(b, probe_nodes, n_probe, observe_nodes, n_obs, delta, mode, out, ldo, workspace, workspace_bytes,
                               stream, nullptr);
}

// The same call with the pairs' difference VECTORS next to the norms: vec[(i * ldo + j) * C + c], unscaled (see
// store_diff_vec in lt_items.hip.h).  For models wider than one pass of these kernels the caller runs one such call per slice
// of the hidden layer (W1[:, s], b1[s], W2[s, t]) and per slice of <= 8 classes and joins them with lt_wide_combine.
// LT_MODE_SPARSE and LT_MODE_DELTA only (FULL names the same quantity as SPARSE, bit for bit).
extern "C" int lt_influence_rows_vec(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                                     const int32_t *observe_nodes, int32_t n_obs, float delta,
                                     int32_t mode, float *out, int64_t ldo, float *vec, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    LT_REQUIRE(vec != nullptr, "lt_influence_rows_vec: vec is NULL");
    LT_REQUIRE(mode == LT_MODE_SPARSE || mode == LT_MODE_DELTA, "lt_influence_rows_vec: mode %d has no vector form (use SPARSE / DELTA)", mode);
    return influence_rows_impl(b, probe_nodes, n_probe, observe_nodes, n_obs, delta, mode, out, ldo, workspace, workspace_bytes,
                               stream, vec);
}

static int influence_rows_impl(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                               const int32_t *observe_nodes, int32_t n_obs, float delta,
                               int32_t mode, float *out, int64_t ldo, void *workspace,
                               size_t workspace_bytes, void *stream, float *vec, double *dst64, int64_t ldd) {
    lt_prof_call prof_call_;
    int32_t exported_rows = 0;      // (dst64) rows the fused route's blocks wrote themselves: the chunks are in probe order
    LT_REQUIRE(b != nullptr, "lt_influence_rows: baseline is NULL");
    LT_REQUIRE(n_probe >= 0 && n_obs >= 0, "lt_influence_rows: negative count");
    LT_REQUIRE(mode >= LT_MODE_FULL && mode <= LT_MODE_DELTA, "lt_influence_rows: unknown mode %d", mode);
    LT_REQUIRE(delta != 0.f && delta == delta, "lt_influence_rows: delta must be a non-zero number");
    if (n_probe == 0 || n_obs == 0) return LT_OK;
    LT_REQUIRE(probe_nodes && observe_nodes && out, "lt_influence_rows: NULL pointer");
    LT_REQUIRE(ldo >= n_obs, "lt_influence_rows: ldo=%lld < n_obs=%d", (long long)ldo, n_obs);
    LT_REQUIRE(b->n > 0, "lt_influence_rows: empty graph");
    { const int rc = lt_node_err_pending(); if (rc) return rc; }     // an earlier call's list held an id out of range
    const size_t need = lt_influence_workspace_bytes(b, n_probe, n_obs, mode);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 256))
        return lt_set_error(LT_ERR_WORKSPACE, "lt_influence_rows: workspace needs %zu bytes, 256-byte aligned", need);

    hipStream_t st = (hipStream_t)stream;
    const lt_graph *g = b->g;
    const infl_ws w = carve_infl(workspace, b, n_probe, n_obs, mode);
    const int lpr = lt_lpr_for(b->Hp), cp = lt_cp_for(b->C), C = b->C, Hp = b->Hp, n = b->n;
    // SPARSE / DELTA read the baseline activations (Z1, S2, OUT; DELTA the fp64 Z1 when enabled); FULL forms
    // what it needs of them itself
    // what the mode reads of the baseline (recomputed here if lt_baseline_refresh marked it stale): FULL the product
    // S1 = X*W1 only (its stage A yields the unperturbed layer itself); SPARSE S1 and the fp32 layers; DELTA with the
    // fp64 pre-activation enabled nothing in fp32 at all -- the probe's own S1 row is read off the fp64 product --
    // and without it S1 and the fp32 layers
    const bool delta64 = mode == LT_MODE_DELTA && b->Z1d != nullptr;

    // SPARSE / DELTA, large calls: the per-node lists of observed nodes, once per call (lt_items.hip.h "pair marks")
    // (with a membership bitmap the per-pair scan is one load per entry and the join pays from ~ 4 M pairs on -- measured
    // at twitch-RU size, tools/marks_ab.py; without one -- large graphs -- it always does)
    // DELTA at twitch size, graphs without hub rows: stage A + stage B of a probe in one block (k_delta_probe_block), no item
    // tables, no bitmap rows, no pair marks -- when the pre-activation is formed on all rows anyway and the block's tables fit
    // LDS ("delta_fused" = 0 keeps the item kernels; the matrices are bit-identical).  Decided once per call.
    const df_geom dg = df_geometry(g, C, n_obs);
    const bool fused = delta64 && vec == nullptr && lt_tune().delta_fused != 0 && dg.ok && w.dl_rec != nullptr &&
                       !lt_fp64_agg_active(b) && !lt_fp64_on_demand(b, n_probe);
    const bool use_marks = !fused && mode != LT_MODE_FULL && w.pm_cnt != nullptr &&
                           (w.bits == nullptr || (long long)(n_probe < w.chunk ? n_probe : w.chunk) * n_obs >= lt_tune().pair_marks);
    // observed hubs (stageB_long_block): members found from the short side, or every entry tested against every probe.
    // With a bitmap row per probe (twitch size) the per-entry test is one cached load and wins; without one it is a search
    // per entry and loses by 10x.  "hub_short_side" pins the choice (tests).
    const bool hub_short = mode != LT_MODE_FULL && (lt_tune().hub_short_side >= 0 ? lt_tune().hub_short_side != 0 : w.bits == nullptr);
    // stage B per observed row (k_item_stageB_rows) instead of per pair: calls with a bitmap row per probe and no pair marks
    // ("stageb_rows" = 0 keeps the per-pair kernel; results are bit-identical)
    // (the vector form is written by the per-pair kernel and the hub blocks only)
    const bool rows_route = mode != LT_MODE_FULL && w.bits != nullptr && !use_marks && lt_tune().stageb_rows != 0 && vec == nullptr;
    // probes of a chunk split over `psplit` blocks per observed node so that the launch fills the chip
    int psplit = 1;
    {
        const int chunk_nb = n_probe < w.chunk ? n_probe : w.chunk;
        while ((long)n_obs * psplit < 2048 && psplit * 32 * LT_SB_UNR < chunk_nb) psplit *= 2;
    }
    // What the mode reads of the baseline, recomputed now if lt_baseline_refresh marked it stale.  The fused route OFFERS the record
    // blocks of its first chunk to the launch that forms the fp64 product rows, should that launch happen (a refreshed baseline on
    // the feature-rows route): nothing in them reads a layer, and that launch has CU slots to spare.
    lt_bits_job cj0 = {};
    bool recs_rode = false, items_rode = false;
    int sparse_rows = 0;            // rows of dst64 zero-filled ahead of the probes' blocks
    {
        int32_t *const node_err0 = lt_node_err_dev();
        const bool offer = fused && dg.record_smem <= (size_t)16 * 1024;
        if (offer) {
            const int nb0 = n_probe < w.chunk ? n_probe : w.chunk;
            cj0.probes = probe_nodes; cj0.nb = nb0; cj0.nblocks = nb0; cj0.dl_rec = w.dl_rec; cj0.dl_meta = g->dl_meta; cj0.dl_src = g->dl_rec;
            cj0.dl_maxc = dg.maxc; cj0.dl_rec_words = dg.rec_words; cj0.observe = observe_nodes; cj0.n_obs = n_obs;
            cj0.n = n; cj0.err = node_err0; cj0.smem_bytes = (unsigned)dg.record_smem;
            // lt_influence_rows_f64: the first rows of the float64 matrix are zero-filled by blocks of that same launch -- np.zeros
            // (attacker.py:216) crossing PCIe while the product rows are formed; their probes' blocks then send the touched positions
            // only (7 % of a row at twitch size), the other probes' blocks widen their whole rows as before
            if (dst64 != nullptr && lt_tune().export_sparse != 0 && n_obs > 0) {
                cj0.zero_dst = dst64; cj0.zero_ld = (long)ldd; cj0.zero_cols = n_obs;
                cj0.zero_rows = (int)(((long long)n_probe * lt_tune().export_zero_share + 99) / 100);
                cj0.zero_blocks = cj0.zero_rows > 0 ? lt_tune().export_zero_blocks : 0;
                cj0.zero_inflight = lt_tune().export_zero_inflight;
            }
            lt_fp64_offer_job(&cj0);
        }
        // ... and the item route of a graph with per-probe bitmap rows its first chunk's item tables (k_item_bits' blocks), the same way
        const bool offer_items = !fused && delta64 && !use_marks && w.bits != nullptr && !lt_fp64_agg_active(b);
        if (offer_items) {
            const int nb0 = n_probe < w.chunk ? n_probe : w.chunk;
            const bool hubs0 = g->p_n_long > 0;
            cj0 = lt_bits_job{g->tptr, g->trow, probe_nodes, nb0, (n + 31) / 32, w.bits, w.off, w.item_pr, w.big_bits, w.big_slot,
                              (int32_t *)nullptr, g->rowptr, observe_nodes, n_obs, hubs0 ? w.hub_obs : (int32_t *)nullptr, nb0 + 1, g->tval,
                              w.item_va};
            cj0.n = n; cj0.err = node_err0; cj0.probes_s = w.probes_s; cj0.obs_s = w.obs_s;      // (DELTA without pair marks: the inline check)
            cj0.smem_bytes = lt_item_bits_smem((n + 31) / 32);
            lt_fp64_offer_job(&cj0);
        }
        int rc = lt_baseline_ensure_padding(b, st);
        if (!rc) rc = mode == LT_MODE_FULL ? lt_baseline_ensure_s1(b, st) : lt_baseline_ensure_layers(b, mode == LT_MODE_DELTA, st, !delta64);
        if (offer) recs_rode = lt_fp64_offer_taken();
        else if (offer_items) items_rode = lt_fp64_offer_taken();
        sparse_rows = recs_rode ? cj0.zero_rows : 0;
        if (rc) return rc;
    }
    // Node ids (lt_items.hip.h checked_node): DELTA calls without pair marks check their lists in the first blocks that read them
    // (the record blocks / k_item_bits -- both usually ride in the pre-activation's launch: no launch, no round trip added);
    // every other call -- FULL / SPARSE start with a GEMM that gathers X[probes], pair marks with the observed rows -- by a
    // launch of its own, ~2 us in front of steps of 0.15 ms and more.  Behind the check every kernel reads the checked lists.
    int32_t *const node_err = lt_node_err_dev();
    const bool inline_check = mode == LT_MODE_DELTA && !use_marks;
    if (!inline_check) {
        const long tot = (long)n_probe + n_obs;
        hipLaunchKernelGGL(k_check_nodes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, probe_nodes, n_probe, observe_nodes,
                           n_obs, n, w.probes_s, w.obs_s, node_err);
        LT_CHECK_LAUNCH();
        probe_nodes = w.probes_s;
        observe_nodes = w.obs_s;
    }
    if (use_marks) {
        LT_REQUIRE((long)n_obs * LT_ROW_SEG / 256 + 1 < 2147483647L, "lt_influence_rows: n_obs=%d exceeds the grid limit", n_obs);
        const unsigned gl = (unsigned)(((long)n_obs * LT_ROW_SEG + 255) / 256);
        LT_HIP(hipMemsetAsync(w.pm_cnt, 0, ((size_t)n + 1) * sizeof(int32_t), st));
#define LT_PM_ARGS g->rowptr, g->col, observe_nodes, n_obs, w.pm_cnt, w.pm_start, w.pm_rank, w.pm_list, w.pm_cnt + n
        hipLaunchKernelGGL(k_pm_lists<0>, dim3(gl), dim3(256), 0, st, LT_PM_ARGS);
        hipLaunchKernelGGL(k_pm_lists<1>, dim3(gl), dim3(256), 0, st, LT_PM_ARGS);
        hipLaunchKernelGGL(k_pm_lists<2>, dim3(gl), dim3(256), 0, st, LT_PM_ARGS);
#undef LT_PM_ARGS
        LT_CHECK_LAUNCH();
    }

    for (int p0 = 0; p0 < n_probe; p0 += w.chunk) {
        const int nb = (n_probe - p0) < w.chunk ? (n_probe - p0) : w.chunk;
        const int32_t *probes = probe_nodes + p0;
        float *orow = out + (int64_t)p0 * ldo;
        float *vrow = vec ? vec + (int64_t)p0 * ldo * b->C : (float *)nullptr;
        const long pairs = (long)nb * n_obs;
        LT_REQUIRE(mode == LT_MODE_FULL || ((pairs * LT_L2_LANES + LT_BLOCK - 1) / LT_BLOCK < 2147483647L &&
                                            ((nb + 31) / 32) * (long)n_obs < 2147483647L),
                   "lt_influence_rows: %d probes x %d observed nodes per chunk exceed the grid limit (lower chunk_budget_bytes)",
                   nb, n_obs);
        const unsigned gridB = (unsigned)((pairs * LT_L2_LANES + LT_BLOCK - 1) / LT_BLOCK);
        // SPARSE / DELTA on a graph with hub rows: blocks for the observed hubs ride in front of stage B's launch
        // hub blocks: 32 probes x one observed hub each, for at most as many observed hubs as the graph has hub rows
        const long long_blocks = (mode != LT_MODE_FULL && g->p_n_long > 0) ? ((nb + 31) / 32) * (long)(n_obs < g->p_n_long ? n_obs : g->p_n_long) : 0;
        LT_REQUIRE(gridB + long_blocks < 2147483647L, "lt_influence_rows: stage-B grid limit");

        if (mode != LT_MODE_DELTA) {
            // perturbed rows and their S1 rows: Sp = (X[v] + X[v]*d) W1            attacker.py:101-105
            if (Hp != b->H) LT_HIP(hipMemsetAsync(w.Sp, 0, (size_t)nb * Hp * sizeof(float), st));
            // (the GEMM gathers row probes[i] of X and perturbs it while loading: no Xp buffer, no extra kernel)
            int rc = lt_launch_gemm_splitk(b->X, b->ldx, b->W1, b->H, w.Sp, Hp, nb, b->H, b->F, probe_kslice(b), w.slabs, st,
                                           probes, delta);
            if (rc) return rc;
        }

        if (mode == LT_MODE_FULL) {
            { lt_prof_scope prof_(LT_K_FULL_A, st);
            // the batched (P probes per wave) kernel also serves narrower layers with part of its lanes idle: it
            // still beats one chain per (row, probe) as soon as the layer has a few dozen columns
            if (Hp >= wide_min_hp()) {
                const int P = full_probes_per_wave(nb);
                const long rblocks = (n + LT_RING_ROWS - 1) / LT_RING_ROWS;
                const int rgroups = (nb + P - 1) / P;
                LT_REQUIRE((rblocks + g->p_n_seg) * rgroups < 2147483647L, "lt_influence_rows: n * probe groups exceeds the grid limit");
                dim3 gridr((unsigned)(rblocks * rgroups));
                // hub rows: segment-parallel (MODE 1 + combine) when their segment sums fit a modest scratch, else
                // one wave per (row, <= 16 probes) walking the segments in turn (MODE 2).  Either way on a side
                // stream next to the plain rows: alone those few long waves would leave most of the chip idle.
                const bool par = g->p_n_seg > 0 && long_rows_parallel(g, Hp);
                const int n_segblocks = par ? g->p_n_seg * rgroups : 0;
                const int PL = P == 8 ? 8 : 16, lgroups = (nb + PL - 1) / PL;
                hipStream_t ls = st;
                if (g->p_n_long > 0 && overlap_enabled() && b->side) {   // stream + events made by lt_baseline_create
                    ls = b->side;
                    LT_HIP(hipEventRecord(b->ev_fork, st));
                    LT_HIP(hipStreamWaitEvent(ls, b->ev_fork, 0));
                }
#define LT_RING_ARGS n, g->rowptr, g->col, g->val, b->S1, Hp, b->b1p, b->W2p, C, probes, nb, w.Sp, w.S2p
#define LT_RING_LAUNCH(P_)                                                                                    \
    do {                                                                                                      \
        if (n_segblocks > 0) {                                                                                \
            LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_stageA_lds<CP_, P_, 1>), dim3(n_segblocks),         \
                                                   dim3(64), 0, ls, LT_RING_ARGS, n_segblocks,                \
                                                   g->p_seg_long, g->p_seg_begin, g->p_long_row, w.lpart,     \
                                                   w.lhit));                                                  \
            LT_CHECK_LAUNCH();                                                                                \
            LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_long_combine<CP_, P_>),                             \
                                                   dim3((unsigned)(((long)g->p_n_long * (nb + 1) + 3) / 4)),  \
                                                   dim3(LT_BLOCK), 0, ls, Hp, b->W2p, C, nb, g->p_n_long,     \
                                                   g->p_long_row, g->p_long_segptr, w.lpart, w.lhit, w.S2p)); \
            LT_CHECK_LAUNCH();                                                                                \
        }                                                                                                     \
        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_stageA_lds<CP_, P_, 0>), gridr, dim3(64), 0, st,        \
                                               LT_RING_ARGS, 0, (const int32_t *)nullptr,                     \
                                               (const int32_t *)nullptr, (const int32_t *)nullptr,            \
                                               (float *)nullptr, (unsigned *)nullptr));                       \
    } while (0)
#define LT_LONG_LAUNCH(P_)                                                                                    \
    do {                                                                                                      \
        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_stageA_lds<CP_, P_, 2>),                                \
                                               dim3((unsigned)((long)g->p_n_long * lgroups)), dim3(64), 0,    \
                                               ls, LT_RING_ARGS, 0, (const int32_t *)nullptr,                 \
                                               (const int32_t *)nullptr, g->p_long_row, (float *)nullptr,     \
                                               (unsigned *)nullptr));                                         \
        LT_CHECK_LAUNCH();                                                                                    \
    } while (0)
                if (g->p_n_long > 0 && !par) {
                    LT_REQUIRE((long)g->p_n_long * lgroups < 2147483647L, "lt_influence_rows: grid limit (long rows)");
                    if (PL == 8) LT_LONG_LAUNCH(8);
                    else LT_LONG_LAUNCH(16);
                }
                if (P == 8) LT_RING_LAUNCH(8);
                else if (P == 32) LT_RING_LAUNCH(32);
                else LT_RING_LAUNCH(16);
#undef LT_RING_LAUNCH
#undef LT_LONG_LAUNCH
#undef LT_RING_ARGS
                if (ls != st) {
                    LT_HIP(hipEventRecord(b->ev_join, ls));
                    LT_HIP(hipStreamWaitEvent(st, b->ev_join, 0));
                }
            } else {
                const long gpb = (LT_BLOCK / 64) * (64 / lpr);               // (row, column) groups per block
                const long blocks = ((long)n * (nb + 1) + gpb - 1) / gpb;    // columns = probes + the baseline
                LT_REQUIRE(blocks < 2147483647L, "lt_influence_rows: n * probes exceeds the grid limit");
                LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
                    hipLaunchKernelGGL((k_full_stageA<LPR_, CP_>), dim3((unsigned)blocks), dim3(LT_BLOCK), 0, st, n,
                                       g->rowptr, g->col, g->val, b->S1, Hp, b->b1p, b->W2p, C, probes, nb,
                                       w.Sp, w.S2p)));
            } }
            LT_CHECK_LAUNCH();
            { lt_prof_scope prof_(LT_K_FULL_B, st);
            const unsigned gridB2 = (unsigned)(((nb + 63) / 64) * (long)n_obs);
            if (g->p_n_long > 0) {
                LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_stageB<CP_, 4>), dim3(gridB2), dim3(256), 0, st, n,
                                                       g->rowptr, g->col, g->val, w.S2p, C, b->b2,
                                                       observe_nodes, n_obs, nb, delta, orow, (long)ldo));
            } else {
                LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_full_stageB<CP_, 1>), dim3(gridB2), dim3(64), 0, st, n,
                                                       g->rowptr, g->col, g->val, w.S2p, C, b->b2,
                                                       observe_nodes, n_obs, nb, delta, orow, (long)ldo));
            } }
            LT_CHECK_LAUNCH();
        } else {
            const int words = (n + 31) / 32;
            // item offsets, the (probe, row) table of the items and the membership bitmap, one block per probe
            if (w.big_slot) LT_HIP(hipMemsetAsync(w.big_slot + w.chunk, 0, sizeof(int32_t), st));
            const unsigned *marks = nullptr;
            bool pair_list = false;        // this chunk's marked pairs are a list (k_pm_compact)
            // DELTA on an S1d route whose pre-activation is still to be formed, all rows at once: the item tables ride in that
            // launch (lt_fp64_prepare_rows); otherwise -- and always in SPARSE -- a launch of their own
            {
                if (fused) {
                    // the records of the chunk's probes: as extra blocks of the launch that forms the pre-activation when there is
                    // one (nothing in it depends on them, and it hides them), else a launch of their own
                    {
                        lt_bits_job cj = {};
                        cj.probes = probes; cj.nb = nb; cj.nblocks = nb; cj.dl_rec = w.dl_rec; cj.dl_meta = g->dl_meta; cj.dl_src = g->dl_rec;
                        cj.dl_maxc = dg.maxc; cj.dl_rec_words = dg.rec_words; cj.observe = observe_nodes; cj.n_obs = n_obs;
                        cj.n = n; cj.err = node_err;
                        cj.smem_bytes = (unsigned)dg.record_smem;
                        bool rode = p0 == 0 && recs_rode;      // (chunk 0's records went along with the product rows' launch)
                        if (p0 == 0 && recs_rode) {
                            // (the next rows of lt_influence_rows_f64's matrix are zero-filled under this launch)
                            lt_bits_job zj = {};
                            bool z_rode = false;
                            if (cj0.zero_blocks > 0 && lt_tune().export_zero_share2 > 0 && cj0.zero_rows < n_probe) {
                                zj.zero_dst = cj0.zero_dst; zj.zero_ld = cj0.zero_ld; zj.zero_cols = cj0.zero_cols; zj.zero_row0 = cj0.zero_rows;
                                const long long want = ((long long)n_probe * lt_tune().export_zero_share2 + 99) / 100;
                                zj.zero_rows = (int)std::min<long long>(want, n_probe - cj0.zero_rows);
                                zj.zero_blocks = cj0.zero_blocks; zj.zero_inflight = cj0.zero_inflight;
                            }
                            int rc = lt_fp64_prepare_rows(b, nullptr, 0, nullptr, n_probe, st, zj.zero_blocks > 0 ? &zj : nullptr,
                                                          zj.zero_blocks > 0 ? &z_rode : nullptr);
                            if (rc) return rc;
                            if (z_rode) sparse_rows += zj.zero_rows;
                        } else if (p0 == 0) {
                            // (a launch's dynamic LDS is given to ALL its blocks: beyond 16 KB of node list the records get a launch
                            // of their own rather than cost the row blocks their occupancy)
                            const bool ride = dg.record_smem <= (size_t)16 * 1024;
                            int rc = lt_fp64_prepare_rows(b, nullptr, 0, nullptr, n_probe, st, ride ? &cj : nullptr, ride ? &rode : nullptr);
                            if (rc) return rc;
                        }
                        if (!rode) {
                            lt_prof_scope prof_(LT_K_ITEM_BITS, st);
                            hipLaunchKernelGGL(k_delta_records, dim3((unsigned)nb), dim3(256), dg.record_smem, st, cj);
                            LT_CHECK_LAUNCH();
                        }
                    }
                    lt_prof_scope prof_(LT_K_ITEM_B, st);
                    const float *sxp = b->s1_f32 ? b->S1x : (const float *)nullptr;
                    const float *zxp = b->z1x_valid ? b->Z1x : (const float *)nullptr;
                    const double *crp = b->cref_deferred ? b->fd_cref : (const double *)nullptr;
                    // (beyond the default 64 KB of dynamic LDS the kernel is told once per instantiation that it may take more)
                    const unsigned df_threads = nb > 2560 ? 64u : (nb > 1280 ? 128u : (unsigned)LT_BLOCK);
#define LT_DF_LAUNCH(SX_, ZF_)                                                                                                        \
    LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,                                                                                           \
        if (dg.finish_smem > (size_t)64 * 1024) { const int rc_ = df_allow_big_lds<LPR_, CP_, SX_, ZF_>(); if (rc_) return rc_; }    \
        hipLaunchKernelGGL((k_delta_probe_finish<LPR_, CP_, SX_, ZF_>), dim3((unsigned)nb), dim3(df_threads), dg.finish_smem, st,      \
                           b->Z1d, b->S1d, sxp, crp, b->S1qs, zxp, Hp, b->W2p, C, w.dl_rec, dg.rec_words, dg.maxc, g->dl_rec,        \
                           n_obs, delta, orow, (long)ldo, drow64, (long)ldd, sparse_rows - p0)))
                    double *const drow64 = (dst64 && exported_rows == p0) ? dst64 + (int64_t)p0 * ldd : (double *)nullptr;
                    if (drow64) exported_rows = p0 + nb;
                    if (sxp && zxp) { LT_DF_LAUNCH(true, true); }
                    else if (sxp) { LT_DF_LAUNCH(true, false); }
                    else { LT_DF_LAUNCH(false, false); }
#undef LT_DF_LAUNCH
                    LT_CHECK_LAUNCH();
                    continue;
                }
            }
            bool bits_done = false;
            lt_bits_job job = {g->tptr, g->trow, probes, nb, words, w.bits, w.off, w.item_pr, w.big_bits, w.big_slot,
                               w.big_slot ? w.big_slot + w.chunk : (int32_t *)nullptr, g->rowptr, observe_nodes, n_obs,
                               long_blocks > 0 ? w.hub_obs : (int32_t *)nullptr, nb + (long_blocks > 0 ? 1 : 0), g->tval, w.item_va};
            if (inline_check) {      // these blocks are the first to read the lists: they check them, the kernels behind read the copies
                job.n = n; job.err = node_err; job.probes_s = w.probes_s + p0; job.obs_s = w.obs_s;
                job.nblocks = nb + 1;
            }
            job.smem_bytes = (w.bits || w.big_bits) ? lt_item_bits_smem(words) : 0u;      // the bitmap row built in LDS when it fits
            if (p0 == 0 && items_rode) bits_done = true;      // (chunk 0's tables went along with the product rows' launch)
            else if (mode == LT_MODE_DELTA && b->Z1d && !lt_fp64_agg_active(b) && !use_marks) {
                const int rc = lt_fp64_prepare_rows(b, w.off, nb, w.item_pr, n_probe, st, &job, &bits_done);
                if (rc) return rc;
            }
            { lt_prof_scope prof_(LT_K_ITEM_BITS, st, !bits_done || use_marks);   // (nothing to time when the tables rode along)
            if (!bits_done) {
                hipLaunchKernelGGL(k_item_bits, dim3((unsigned)job.nblocks), dim3(256), job.smem_bytes, st, g->tptr, g->trow, probes, nb, words,
                                   w.bits, w.off, w.item_pr, w.big_bits, w.big_slot, job.big_count,
                                   g->rowptr, observe_nodes, n_obs, job.hub_obs, g->tval, w.item_va, job.n, job.err, job.probes_s,
                                   job.obs_s, job.smem_bytes ? words : 0);
                LT_CHECK_LAUNCH();
            }
            if (inline_check) {      // from here on: the checked lists
                probes = w.probes_s + p0;
                observe_nodes = w.obs_s;
            }
            if (use_marks) {
                LT_HIP(hipMemsetAsync(w.pm_marks, 0, (size_t)((pairs + 31) / 32) * sizeof(unsigned), st));
                hipLaunchKernelGGL(k_pm_mark, dim3(LT_ITEM_GRID), dim3(256), 0, st, w.off, nb, w.item_pr, w.pm_cnt, w.pm_start,
                                   w.pm_list, n_obs, w.pm_marks);
                LT_CHECK_LAUNCH();
                marks = w.pm_marks;
                // the marked pairs as a list, `out` zero-filled: the pair kernel then walks the list (k_item_stageB_list)
                pair_list = w.pm_pairs != nullptr && vrow == nullptr;
                if (pair_list) {
                    LT_HIP(hipMemsetAsync(w.pm_npairs, 0, sizeof(int32_t), st));
                    const long n_words = (pairs + 31) / 32;
                    hipLaunchKernelGGL(k_pm_compact, dim3((unsigned)std::min<long>((n_words + 255) / 256, 4096)), dim3(256), 0, st, marks,
                                       n_words, w.pm_pairs, w.pm_npairs);
                    LT_CHECK_LAUNCH();
                    LT_HIP(hipMemset2DAsync(orow, (size_t)ldo * sizeof(float), 0, (size_t)n_obs * sizeof(float), (size_t)nb, st));
                }
            } }
            if (mode == LT_MODE_SPARSE) {
                { lt_prof_scope prof_(LT_K_ITEM_A, st);
                LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
                    hipLaunchKernelGGL((k_item_stageA<LPR_, CP_, 0>), dim3(LT_ITEM_GRID),
                                       dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, g->tptr,
                                       g->trow, g->tval, b->S1, b->Z1, b->Z1d, b->S1d, Hp, b->b1p, b->W2p, C, probes, nb,
                                       w.off, w.Sp, delta, w.S2x, g->p_n_long, g->p_long_row, g->p_long_segptr,
                                       b->seg_part, w.item_pr, (const double *)nullptr, (const double *)nullptr,
                                       (const float *)nullptr))); }
                LT_CHECK_LAUNCH();
                lt_prof_scope prof_(LT_K_ITEM_B, st);
                if (long_blocks > 0 && hub_short) {
                    LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_hubs<CP_, false, true>), dim3((unsigned)long_blocks),
                                                           dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val,
                                                           g->tptr, g->trow, b->S2, C, b->b2, b->OUT, probes,
                                                           nb, w.off, w.S2x, observe_nodes, n_obs, delta,
                                                           orow, (long)ldo, w.bits, words, w.big_bits, w.big_slot, w.hub_obs, vrow));
                    LT_CHECK_LAUNCH();
                }
                const unsigned inl = hub_short ? 0u : (unsigned)long_blocks;   // hub blocks in front of the pair launch
                if (rows_route) {
                    if (inl > 0) {       // the observed hubs: the hub blocks of the pair kernel, launched without its pairs
                        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB<CP_, false>), dim3(inl), dim3(LT_BLOCK), 0, st,
                                                               g->rowptr, g->col, g->val, g->tptr, g->trow, b->S2, C, b->b2,
                                                               b->OUT, probes, nb, w.off, w.S2x, observe_nodes, n_obs, delta,
                                                               orow, (long)ldo, w.bits, words, (int)inl, 1, marks, w.big_bits,
                                                               w.big_slot, 0, w.hub_obs, vrow));
                        LT_CHECK_LAUNCH();
                    }
                    LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_rows<CP_, false>), dim3((unsigned)((long)n_obs * psplit)),
                                                           dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, b->S2, C, b->b2,
                                                           b->OUT, nb, w.off, w.S2x, observe_nodes, n_obs, delta, orow, (long)ldo,
                                                           w.bits, words, psplit));
                } else if (pair_list) {
                    if (inl > 0) {       // the observed hubs: the hub blocks of the pair kernel, launched without its pairs
                        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB<CP_, false>), dim3(inl), dim3(LT_BLOCK), 0, st,
                                                               g->rowptr, g->col, g->val, g->tptr, g->trow, b->S2, C, b->b2,
                                                               b->OUT, probes, nb, w.off, w.S2x, observe_nodes, n_obs, delta,
                                                               orow, (long)ldo, w.bits, words, (int)inl, 1, marks, w.big_bits,
                                                               w.big_slot, 0, w.hub_obs, vrow));
                        LT_CHECK_LAUNCH();
                    }
                    LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_list<CP_, false>), dim3(LT_ITEM_GRID), dim3(LT_BLOCK), 0, st,
                                                           g->rowptr, g->col, g->val, g->tptr, g->trow, b->S2, C, b->b2, b->OUT,
                                                           probes, nb, w.off, w.S2x, observe_nodes, n_obs, delta, orow, (long)ldo,
                                                           w.bits, words, long_blocks > 0 ? 1 : 0, marks, w.big_bits, w.big_slot,
                                                           vrow, w.pm_pairs, w.pm_npairs));
                } else
                LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB<CP_, false>), dim3(gridB + inl),
                                                       dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val,
                                                       g->tptr, g->trow, b->S2, C, b->b2, b->OUT, probes,
                                                       nb, w.off, w.S2x, observe_nodes, n_obs, delta,
                                                       orow, (long)ldo, w.bits, words, (int)inl, long_blocks > 0 ? 1 : 0, marks,
                                                       w.big_bits, w.big_slot, 0, w.hub_obs, vrow));
            } else {
                const double *spd = nullptr;
                if (b->Z1d && lt_fp64_agg_active(b)) {
                    // aggregate-first: the pre-activation rows this chunk's items read, and the probes' product rows
                    const int rc = lt_fp64_prepare_items(b, w.off, nb, w.item_pr, probes, w.Spd, st);
                    if (rc) return rc;
                    spd = w.Spd;
                } else if (b->Z1d) {
                    // S1d routes: the pre-activation rows this chunk's items read (all rows, or on demand for a small call)
                    const int rc = lt_fp64_prepare_rows(b, w.off, nb, w.item_pr, n_probe, st);
                    if (rc) return rc;
                }
                { lt_prof_scope prof_(LT_K_ITEM_A, st);
                if (b->Z1d) {
                    const float *sxp = (b->s1_f32 && !spd) ? b->S1x : (const float *)nullptr;
                    const float *zxp = b->z1x_valid ? b->Z1x : (const float *)nullptr;
#define LT_D2_LAUNCH(SX_, ZF_)                                                                                                    \
    LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,                                                                                       \
        hipLaunchKernelGGL((k_item_stageA_d2<LPR_, CP_, SX_, ZF_>), dim3(LT_ITEM_GRID), dim3(LT_BLOCK), 0, st, b->Z1d, b->S1d, sxp,  \
                           spd, b->cref_deferred ? b->fd_cref : (const double *)nullptr, Hp, b->W2p, C, nb, w.off, delta, w.S2x,  \
                           w.item_pr, w.item_va, zxp, b->S1qs)))
                    if (sxp && zxp) { LT_D2_LAUNCH(true, true); }
                    else if (sxp) { LT_D2_LAUNCH(true, false); }
                    else { LT_D2_LAUNCH(false, false); }
#undef LT_D2_LAUNCH
                } else {
                    LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
                        hipLaunchKernelGGL((k_item_stageA<LPR_, CP_, 1>), dim3(LT_ITEM_GRID), dim3(LT_BLOCK), 0,
                                           st, g->rowptr, g->col, g->val, g->tptr, g->trow, g->tval, b->S1,
                                           b->Z1, b->Z1d, b->S1d, Hp, b->b1p, b->W2p, C, probes, nb, w.off,
                                           (const float *)nullptr, delta, w.S2x, 0, (const int32_t *)nullptr,
                                           (const int32_t *)nullptr, (const float *)nullptr, w.item_pr, (const double *)nullptr,
                                           (const double *)nullptr, (const float *)nullptr)));
                } }
                LT_CHECK_LAUNCH();
                lt_prof_scope prof_(LT_K_ITEM_B, st);
                // twitch size (a bitmap row per probe, every probe walks the hub's row): the hub blocks ride in front of the per-row
                // kernel's launch
                const bool hubs_ride = long_blocks > 0 && rows_route && !hub_short && w.bits != nullptr && vrow == nullptr;
                if (long_blocks > 0 && !hubs_ride) {   // the observed hubs, a launch of their own
                    if (hub_short) {
                        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_hubs<CP_, true, true>), dim3((unsigned)long_blocks),
                                                               dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, g->tptr, g->trow,
                                                               b->S2, C, b->b2, b->OUT, probes, nb, w.off, w.S2x, observe_nodes,
                                                               n_obs, delta, orow, (long)ldo, w.bits, words, w.big_bits,
                                                               w.big_slot, w.hub_obs, vrow));
                    } else {
                        LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_hubs<CP_, true, false>), dim3((unsigned)long_blocks),
                                                               dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, g->tptr, g->trow,
                                                               b->S2, C, b->b2, b->OUT, probes, nb, w.off, w.S2x, observe_nodes,
                                                               n_obs, delta, orow, (long)ldo, w.bits, words, w.big_bits,
                                                               w.big_slot, w.hub_obs, vrow));
                    }
                    LT_CHECK_LAUNCH();
                }
                if (rows_route) {
                    const long hb = hubs_ride ? long_blocks : 0;
                    LT_REQUIRE((long)n_obs * psplit + hb < 2147483647L, "lt_influence_rows: stage-B grid limit");
                    LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_rows<CP_, true>), dim3((unsigned)((long)n_obs * psplit + hb)),
                                                           dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val, b->S2, C, b->b2,
                                                           b->OUT, nb, w.off, w.S2x, observe_nodes, n_obs, delta, orow, (long)ldo,
                                                           w.bits, words, psplit, (int)hb, g->tptr, g->trow, probes, w.hub_obs));
                } else if (pair_list) {
                LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB_list<CP_, true>), dim3(LT_ITEM_GRID), dim3(LT_BLOCK), 0, st,
                                                       g->rowptr, g->col, g->val, g->tptr, g->trow, b->S2, C, b->b2, b->OUT,
                                                       probes, nb, w.off, w.S2x, observe_nodes, n_obs, delta, orow, (long)ldo,
                                                       w.bits, words, long_blocks > 0 ? 1 : 0, marks, w.big_bits, w.big_slot, vrow,
                                                       w.pm_pairs, w.pm_npairs));
                } else
                LT_DISPATCH_CP(cp, hipLaunchKernelGGL((k_item_stageB<CP_, true>), dim3(gridB),
                                                       dim3(LT_BLOCK), 0, st, g->rowptr, g->col, g->val,
                                                       g->tptr, g->trow, b->S2, C, b->b2, b->OUT, probes,
                                                       nb, w.off, w.S2x, observe_nodes, n_obs, delta,
                                                       orow, (long)ldo, w.bits, words, 0, long_blocks > 0 ? 1 : 0,
                                                       marks, w.big_bits, w.big_slot, hub_short ? 1 : 0, w.hub_obs, vrow));
            }
            LT_CHECK_LAUNCH();
        }
    }
    if (dst64 && exported_rows < n_probe && n_obs > 0)      // (the rows no block exported itself: one launch behind the last kernel)
        return lt_export_rows_dev(out + (int64_t)exported_rows * ldo, ldo, n_probe - exported_rows, n_obs,
                                  dst64 + (int64_t)exported_rows * ldd, ldd, (hipStream_t)stream);
    return LT_OK;
}

// ---- lt_wide_combine: the slices' difference vectors -> scores (include/linkteller_hip.h) ---------------------------------
#define LT_WIDE_MAX_VEC 32
struct lt_vec_ptrs { const float *p[LT_WIDE_MAX_VEC]; };
__global__ __launch_bounds__(256) void k_wide_combine(lt_vec_ptrs v, int n_vec, long n_pairs, int C, float delta,
                                                      float *__restrict__ ss, int first, int last) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    float acc = first ? 0.f : ss[i];
    for (int c = 0; c < C; ++c) {
        float d = v.p[0][i * C + c];
        for (int s = 1; s < n_vec; ++s) d += v.p[s][i * C + c];     // hidden slices in slice order
        d = d / delta;                                              // attacker.py:105-106
        acc = fmaf(d, d, acc);
    }
    ss[i] = last ? sqrtf(acc) : acc;
}

extern "C" int lt_wide_combine(const float *const *vecs, int32_t n_vec, int64_t n_pairs, int32_t C, float delta,
                               float *ss, int32_t first, int32_t last, void *stream) {
    LT_REQUIRE(vecs != nullptr && ss != nullptr, "lt_wide_combine: NULL pointer");
    LT_REQUIRE(n_vec >= 1 && n_vec <= LT_WIDE_MAX_VEC, "lt_wide_combine: %d vectors (1 .. %d hidden slices)", n_vec, LT_WIDE_MAX_VEC);
    LT_REQUIRE(n_pairs >= 0 && C >= 1 && C <= LT_MAX_C, "lt_wide_combine: n_pairs=%lld C=%d", (long long)n_pairs, C);
    LT_REQUIRE(delta != 0.f && delta == delta, "lt_wide_combine: delta must be a non-zero number");
    if (n_pairs == 0) return LT_OK;
    lt_vec_ptrs v = {};
    for (int s_ = 0; s_ < n_vec; ++s_) {
        LT_REQUIRE(vecs[s_] != nullptr, "lt_wide_combine: vecs[%d] is NULL", s_);
        v.p[s_] = vecs[s_];
    }
    LT_REQUIRE((n_pairs + 255) / 256 < 2147483647L, "lt_wide_combine: grid limit");
    hipLaunchKernelGGL(k_wide_combine, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, n_vec,
                       (long)n_pairs, C, delta, ss, first, last);
    LT_CHECK_LAUNCH();
    return LT_OK;
}
