// Fused GCN layers, the 2-layer forward and the baseline state (the standalone SpMM lives in lt_spmm.hip).
//   reference call sites: torch.spmm + bias (gcn/layers.py:32-36), F.relu (gcn/models.py:20),
//   GCN.forward (gcn/models.py:19-24), the loop-invariant model(features, adj) of
//   attacker.py:106.
#include <new>

#include "lt_rows.hip.h"

#define LT_BLOCK 256

// --------------------------------------------------------------------------------------------
// Fused layer 1 for every row: acc = A_hat[r,:]*S1; Z1[r] = acc + b1 (optional store);
// S2[r] = relu(Z1[r]) . W2.  H1 never exists in memory.
// --------------------------------------------------------------------------------------------
template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_layer1(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S1, int Hp,
    const float *__restrict__ b1p, const float *__restrict__ W2p, int C,
    float *__restrict__ Z1, float *__restrict__ S2, int skip_long, int seg_blocks, int n_seg,
    const int32_t *__restrict__ seg_long, const int32_t *__restrict__ seg_begin,
    const int32_t *__restrict__ long_row, float *__restrict__ seg_part) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (LPR - 1);
    if ((int)blockIdx.x < seg_blocks) {
        // the first blocks of the launch take the SEGMENTS of the hub rows (one lane group per segment, the chain of a
        // row's first segment started from the bias: row_dot's canonical order); k_layer1_long adds them afterwards.
        // Same launch as the plain rows: a graph with a few hubs does not pay a launch of its own for them.
        int sg = ((blockIdx.x * LT_BLOCK + threadIdx.x) >> 6) * RPW + lane / LPR;
        if (LPR == 64) sg = __builtin_amdgcn_readfirstlane(sg);
        if (sg >= n_seg) return;
        const int rs = long_row[seg_long[sg]];
        const int s0 = seg_begin[sg], s1 = min(rowptr[rs + 1], s0 + LT_ROW_SEG);
        const int co = 4 * gl;
        const bool act = co < Hp;
        const f32x4 init = (act && s0 == rowptr[rs]) ? ld4(b1p + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 zs = seg_chain<16>(col, val, s0, s1, S1, Hp, co, act, -1, nullptr, init);
        if (act) *reinterpret_cast<f32x4 *>(seg_part + (size_t)sg * Hp + co) = zs;
        return;
    }
    const int wave = (((int)blockIdx.x - seg_blocks) * LT_BLOCK + threadIdx.x) >> 6;
    int r = wave * RPW + lane / LPR;
    if (LPR == 64) r = __builtin_amdgcn_readfirstlane(r);
    if (r >= n) return;  // LPR-lane groups exit together; shuffles below stay inside a group
    // rows of more than LT_ROW_SEG entries are summed segment by segment (above) and finished by k_layer1_long
    if (skip_long && rowptr[r + 1] - rowptr[r] > LT_ROW_SEG) return;
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    const f32x4 b1v = active ? ld4(b1p + coff) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 z = row_dot<8>(col, val, rowptr[r], rowptr[r + 1], S1, Hp, coff, active, -1, nullptr, b1v);
    float part[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = 0.f;
    if (active) {
        relu_w2_partial<CP>(z, W2p + (size_t)coff * C, C, part);
        if (Z1) *reinterpret_cast<f32x4 *>(Z1 + (size_t)r * Hp + coff) = z;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) S2[(size_t)r * C + c] = part[c];
    }
}

// Long rows of layer 1 (hubs): their segment sums come out of k_layer1's first blocks (or the tiled kernel) ...
// ... and one group per long ROW adds them in segment order and finishes like k_layer1.
template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_layer1_long(
    int n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_segptr,
    const float *__restrict__ part, int Hp, const float *__restrict__ W2p, int C,
    float *__restrict__ Z1, float *__restrict__ S2) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int gl = lane & (LPR - 1);
    int li = wave * RPW + lane / LPR;
    if (LPR == 64) li = __builtin_amdgcn_readfirstlane(li);
    if (li >= n_long) return;
    const int r = long_row[li];
    const int s0 = long_segptr[li], s1 = long_segptr[li + 1];
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        z = *reinterpret_cast<const f32x4 *>(part + (size_t)s0 * Hp + coff);
        int s = s0 + 1;
        for (; s + 8 <= s1; s += 8) {   // 8 segment sums in flight (a hub has hundreds), added in segment order
            f32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4 *>(part + (size_t)(s + k) * Hp + coff);
#pragma unroll
            for (int k = 0; k < 8; ++k) { z.x += t[k].x; z.y += t[k].y; z.z += t[k].z; z.w += t[k].w; }
        }
        for (; s < s1; ++s) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(part + (size_t)s * Hp + coff);
            z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w;
        }
    }
    float p2[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) p2[c] = 0.f;
    if (active) {
        relu_w2_partial<CP>(z, W2p + (size_t)coff * C, C, p2);
        if (Z1) *reinterpret_cast<f32x4 *>(Z1 + (size_t)r * Hp + coff) = z;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) p2[c] = group_sum<LPR>(p2[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) S2[(size_t)r * C + c] = p2[c];
    }
}

// Large graphs: Z1 comes from the tiled SpMM (lt_spmm.hip; chains started from the bias, so Z1 is the pre-activation
// itself) and this pass finishes the rows of up to LT_ROW_SEG entries: S2[r] = relu(Z1[r]) . W2 with the lane
// reduction of k_layer1 (the long rows are finished by k_layer1_long from their segment sums).
template <int LPR, int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_layer1_tail(
    int n, const int32_t *__restrict__ rowptr, const float *__restrict__ Z1, int Hp,
    const float *__restrict__ W2p, int C, float *__restrict__ S2) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LT_BLOCK + threadIdx.x) >> 6;
    const int gl = lane & (LPR - 1);
    int r = wave * RPW + lane / LPR;
    if (LPR == 64) r = __builtin_amdgcn_readfirstlane(r);
    if (r >= n) return;
    if (rowptr[r + 1] - rowptr[r] > LT_ROW_SEG) return;
    const int coff = 4 * gl;
    const bool active = coff < Hp;
    float part[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = 0.f;
    if (active) {
        const f32x4 z = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Z1 + (size_t)r * Hp + coff));
        relu_w2_partial<CP>(z, W2p + (size_t)coff * C, C, part);
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) part[c] = group_sum<LPR>(part[c]);
    if (gl == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) S2[(size_t)r * C + c] = part[c];
    }
}

// Layer 2 of a hub row: a row of 10^3 entries is 10^2 dependent load round trips for the 8 lanes of k_layer2 and the
// whole launch waits for it.  Here a block (the first blocks of k_layer2's launch) takes one long row: all 256 threads fetch entries (val, S2 row) into LDS,
// LT_L2_CHUNK at a time, and the 8 chain lanes then run row2_dot's chains out of LDS -- entry e still goes to chain
// (e - e0) & 7, chains still k-ordered, same butterfly: the bits of k_layer2.
#define LT_L2_CHUNK 1024
template <int CP>
__device__ __forceinline__ void layer2_long_row(
    int r, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S2, int C, const float *__restrict__ b2,
    float *__restrict__ OUT) {
    __shared__ float sv[LT_L2_CHUNK];
    __shared__ float sT[LT_L2_CHUNK][CP];
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    const int tid = threadIdx.x;
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    for (int cb = e0; cb < e1; cb += LT_L2_CHUNK) {
        const int nc = min(LT_L2_CHUNK, e1 - cb);
        __syncthreads();
        for (int i = tid; i < nc; i += LT_BLOCK) {
            const int cc = col[cb + i];
            sv[i] = val[cb + i];
#pragma unroll
            for (int c = 0; c < CP; ++c) sT[i][c] = c < C ? S2[(size_t)cc * C + c] : 0.f;
        }
        __syncthreads();
        if (tid < LT_L2_LANES) {
            int i = tid;
            for (; i + 7 * LT_L2_LANES < nc; i += 8 * LT_L2_LANES) {   // 8 LDS reads in flight, FMAs in entry order
                float a[8], t[8][CP];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a[k] = sv[i + k * LT_L2_LANES];
#pragma unroll
                    for (int c = 0; c < CP; ++c) t[k][c] = sT[i + k * LT_L2_LANES][c];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int c = 0; c < CP; ++c)
                        if (c < C) acc[c] = fmaf(a[k], t[k][c], acc[c]);
            }
            for (; i < nc; i += LT_L2_LANES) {
                const float a = sv[i];
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) acc[c] = fmaf(a, sT[i][c], acc[c]);
            }
        }
    }
    if (tid < 64) {   // the wave that holds the chain lanes
#pragma unroll
        for (int c = 0; c < CP; ++c) acc[c] = group_sum<LT_L2_LANES>(acc[c]);
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) OUT[(size_t)r * C + c] = acc[c] + b2[c];
        }
    }
}

// Layer 2 for every row: OUT[r] = A_hat[r,:]*S2 + b2
template <int CP>
__global__ __launch_bounds__(LT_BLOCK) void k_layer2(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ S2, int C,
    const float *__restrict__ b2, float *__restrict__ OUT, int n_long, const int32_t *__restrict__ long_row) {
    if ((int)blockIdx.x < n_long) {   // the first blocks of the launch: one hub row each
        layer2_long_row<CP>(long_row[blockIdx.x], rowptr, col, val, S2, C, b2, OUT);
        return;
    }
    const int gid = (((int)blockIdx.x - n_long) * LT_BLOCK + threadIdx.x) / LT_L2_LANES;
    const int q = threadIdx.x & (LT_L2_LANES - 1);
    if (gid >= n) return;
    if (n_long > 0 && rowptr[gid + 1] - rowptr[gid] > LT_ROW_SEG) return;   // a hub row: one of the first blocks
    float acc[CP];
    row2_dot<CP>(col, val, rowptr[gid], rowptr[gid + 1], q, C,
                 [&](int c, int) { return S2 + (size_t)c * C; }, acc);
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) OUT[(size_t)gid * C + c] = acc[c] + b2[c];
    }
}

// b1p[Hp] = b1[H] then zeros and W2p[Hp, C] = W2[H, C] then zero rows, in one launch
__global__ void k_pad_b1_w2(const float *__restrict__ b1, const float *__restrict__ W2, int H, int Hp, int C,
                            float *__restrict__ b1p, float *__restrict__ W2p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Hp) b1p[i] = i < H ? b1[i] : 0.f;
    const int j = i - Hp;
    if (j >= 0 && j < Hp * C) W2p[j] = (j / C) < H ? W2[j] : 0.f;
}

// --------------------------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------------------------
static inline unsigned blocks_for_rows(int n, int rows_per_block) {
    return (unsigned)((n + rows_per_block - 1) / rows_per_block);
}

int lt_launch_layer1(const lt_graph *g, const float *S1, int Hp, const float *b1p,
                     const float *W2p, int C, float *Z1, float *S2, hipStream_t st, float *seg_part) {
    if (g->n == 0) return LT_OK;
    if (!seg_part) seg_part = g->p_seg_scratch;
    const int lpr = lt_lpr_for(Hp), cp = lt_cp_for(C);
    const int rpb = (LT_BLOCK / 64) * (64 / lpr);
    const unsigned grid = blocks_for_rows(g->n, rpb);
    const int have_long = g->p_n_long > 0 ? 1 : 0;
    lt_prof_scope prof_(LT_K_LAYER1, st);
    if (Z1 != nullptr && lt_tiled_wanted(g, Hp)) {
        // S1 is far larger than the L2s: column-sliced work-item kernel for the chains (same chains, so the same
        // bits), then the two finishing passes
        int rc = lt_launch_rows_tiled(g, S1, Hp, Hp, b1p, nullptr, 0, Z1, Hp, seg_part, Hp, st);
        if (rc) return rc;
        if (have_long) {
            LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
                hipLaunchKernelGGL((k_layer1_long<LPR_, CP_>), dim3(blocks_for_rows(g->p_n_long, rpb)), dim3(LT_BLOCK), 0,
                                   st, g->p_n_long, g->p_long_row, g->p_long_segptr, seg_part, Hp, W2p, C, Z1, S2)));
            LT_CHECK_LAUNCH();
        }
        LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
            hipLaunchKernelGGL((k_layer1_tail<LPR_, CP_>), dim3(grid), dim3(LT_BLOCK), 0, st, g->n, g->rowptr, Z1, Hp,
                               W2p, C, S2)));
        LT_CHECK_LAUNCH();
        return LT_OK;
    }
    // plain rows and (first blocks) the segments of the hub rows in one launch, then the hub rows' ordered sums
    const unsigned seg_blocks = have_long ? blocks_for_rows(g->p_n_seg, rpb) : 0u;
    LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
        hipLaunchKernelGGL((k_layer1<LPR_, CP_>), dim3(grid + seg_blocks), dim3(LT_BLOCK), 0, st, g->n,
                           g->rowptr, g->col, g->val, S1, Hp, b1p, W2p, C, Z1, S2, have_long, (int)seg_blocks, g->p_n_seg,
                           g->p_seg_long, g->p_seg_begin, g->p_long_row, seg_part)));
    LT_CHECK_LAUNCH();
    if (have_long) {
        LT_DISPATCH_LPR(lpr, LT_DISPATCH_CP(cp,
            hipLaunchKernelGGL((k_layer1_long<LPR_, CP_>), dim3(blocks_for_rows(g->p_n_long, rpb)), dim3(LT_BLOCK), 0,
                               st, g->p_n_long, g->p_long_row, g->p_long_segptr, seg_part, Hp, W2p, C,
                               Z1, S2)));
        LT_CHECK_LAUNCH();
    }
    return LT_OK;
}

int lt_launch_layer2(const lt_graph *g, const float *S2, int C, const float *b2, float *OUT,
                     hipStream_t st) {
    if (g->n == 0) return LT_OK;
    const unsigned grid = blocks_for_rows(g->n, LT_BLOCK / LT_L2_LANES);
    lt_prof_scope prof_(LT_K_LAYER2, st);
    LT_DISPATCH_CP(lt_cp_for(C),
        hipLaunchKernelGGL((k_layer2<CP_>), dim3(grid + (unsigned)g->p_n_long), dim3(LT_BLOCK), 0, st, g->n, g->rowptr,
                           g->col, g->val, S2, C, b2, OUT, g->p_n_long, g->p_long_row));
    LT_CHECK_LAUNCH();
    return LT_OK;
}

// --------------------------------------------------------------------------------------------
// 2-layer forward with caller workspace
// --------------------------------------------------------------------------------------------
static int check_dims(const char *who, const lt_graph *g, int F, int H, int C) {
    LT_REQUIRE(g != nullptr, "%s: graph is NULL", who);
    LT_REQUIRE(F > 0 && H > 0 && C > 0, "%s: F=%d H=%d C=%d must be positive", who, F, H, C);
    if (H > LT_MAX_H || C > LT_MAX_C)
        return lt_set_error(LT_ERR_UNSUPPORTED, "%s: H=%d C=%d (supported: H <= %d, C <= %d)", who, H,
                            C, LT_MAX_H, LT_MAX_C);
    return LT_OK;
}

struct gcn2_ws {
    float *S1, *S2, *b1p, *W2p, *slabs, *Z1;
    size_t bytes;
};
static gcn2_ws carve_gcn2(void *base, int n, int H, int C, int F) {
    const int Hp = lt_round_up(H, 4);
    size_t off = 0;
    gcn2_ws w;
    char *p = (char *)base;
    w.S1 = (float *)(p + off);  off += lt_align_up((size_t)n * Hp * sizeof(float), 256);
    w.S2 = (float *)(p + off);  off += lt_align_up((size_t)n * C * sizeof(float), 256);
    w.b1p = (float *)(p + off); off += lt_align_up((size_t)Hp * sizeof(float), 256);
    w.W2p = (float *)(p + off); off += lt_align_up((size_t)Hp * C * sizeof(float), 256);
    const size_t sb = lt_gemm_splitk_slab_bytes(n, H, F, lt_gemm_pick_kslice(n, H, F));
    w.slabs = sb ? (float *)(p + off) : nullptr; off += lt_align_up(sb, 256);
    // a graph large enough for the tiled layer-1 route materialises the pre-activation
    const bool tiled = (long long)n * Hp * (long long)sizeof(float) >= lt_tune().tiled_min_bytes;
    w.Z1 = tiled ? (float *)(p + off) : nullptr; off += tiled ? lt_align_up((size_t)n * Hp * sizeof(float), 256) : 0;
    w.bytes = off;
    return w;
}

extern "C" size_t lt_gcn2_workspace_bytes(int32_t n, int32_t F, int32_t H, int32_t C) {
    if (n < 0 || F <= 0 || H <= 0 || C <= 0) return 0;
    return carve_gcn2(nullptr, n, H, C, F).bytes;
}

// S1 = X*W1 into an [n, Hp] buffer whose pad columns are zero; *b1_eff / *W2_eff = b1 / W2 zero-padded to Hp
// rows: the caller's tensors themselves when nothing needs padding (H % 4 == 0, b1 16-byte aligned for the
// float4 reads), else copies made here into b1p_buf / W2p_buf.
static int prepare_layer_inputs(int n, const float *X, int64_t ldx, int F, const float *W1,
                                const float *b1, int H, const float *W2, int C, float *S1,
                                float *b1p_buf, float *W2p_buf, const float **b1_eff, const float **W2_eff,
                                float *slabs, hipStream_t st) {
    const int Hp = lt_round_up(H, 4);
    if (Hp != H) LT_HIP(hipMemsetAsync(S1, 0, (size_t)n * Hp * sizeof(float), st));
    if (Hp == H && ((uintptr_t)b1 % 16) == 0) {
        *b1_eff = b1;
        *W2_eff = W2;
    } else {
        hipLaunchKernelGGL(k_pad_b1_w2, dim3((Hp * (C + 1) + 255) / 256), dim3(256), 0, st, b1, W2, H, Hp, C,
                           b1p_buf, W2p_buf);
        LT_CHECK_LAUNCH();
        *b1_eff = b1p_buf;
        *W2_eff = W2p_buf;
    }
    if (slabs) return lt_launch_gemm_splitk(X, ldx, W1, H, S1, Hp, n, H, F, lt_gemm_pick_kslice(n, H, F), slabs, st);
    return lt_launch_gemm(X, ldx, W1, H, S1, Hp, n, H, F, st);
}

extern "C" int lt_gcn2_forward(const lt_graph *g, const float *X, int64_t ldx, int32_t F,
                               const float *W1, const float *b1, int32_t H, const float *W2,
                               const float *b2, int32_t C, float *logits, int64_t ldl,
                               void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_dims("lt_gcn2_forward", g, F, H, C);
    if (rc) return rc;
    LT_REQUIRE(X && W1 && b1 && W2 && b2 && logits, "lt_gcn2_forward: NULL tensor pointer");
    LT_REQUIRE(ldx >= F, "lt_gcn2_forward: ldx=%lld < F=%d", (long long)ldx, F);
    LT_REQUIRE(ldl == C, "lt_gcn2_forward: logits must be dense (ldl == C)");
    if (g->n == 0) return LT_OK;
    if (!workspace || workspace_bytes < lt_gcn2_workspace_bytes(g->n, F, H, C) || ((uintptr_t)workspace % 256))
        return lt_set_error(LT_ERR_WORKSPACE, "lt_gcn2_forward: workspace needs %zu bytes, 256-byte aligned",
                            lt_gcn2_workspace_bytes(g->n, F, H, C));
    hipStream_t st = (hipStream_t)stream;
    gcn2_ws w = carve_gcn2(workspace, g->n, H, C, F);
    const int Hp = lt_round_up(H, 4);
    const float *b1e = nullptr, *W2e = nullptr;
    rc = prepare_layer_inputs(g->n, X, ldx, F, W1, b1, H, W2, C, w.S1, w.b1p, w.W2p, &b1e, &W2e, w.slabs, st);
    if (rc) return rc;
    rc = lt_launch_layer1(g, w.S1, Hp, b1e, W2e, C, w.Z1, w.S2, st);
    if (rc) return rc;
    return lt_launch_layer2(g, w.S2, C, b2, logits, st);
}

// --------------------------------------------------------------------------------------------
// baseline state
// --------------------------------------------------------------------------------------------
static void free_baseline(lt_baseline *b) {
    if (!b) return;
    if (b->S1_owned) (void)hipFree(b->S1);
    (void)hipFree(b->Z1);
    (void)hipFree(b->S2);
    (void)hipFree(b->OUT);
    (void)hipFree(b->b1p_buf);
    (void)hipFree(b->W2p_buf);
    (void)hipFree(b->slabs);
    (void)hipFree(b->seg_part);
    lt_baseline_free_fp64(b);
    if (b->side) { (void)hipStreamSynchronize(b->side); (void)hipStreamDestroy(b->side); }
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    delete b;
}

// b1 / W2 zero-padded to Hp rows: the caller's tensors themselves when nothing needs padding, else copies
static int refresh_padding(lt_baseline *b, hipStream_t st) {
    if (b->Hp == b->H && ((uintptr_t)b->b1 % 16) == 0) {
        b->b1p = b->b1;
        b->W2p = b->W2;
        return LT_OK;
    }
    hipLaunchKernelGGL(k_pad_b1_w2, dim3((b->Hp * (b->C + 1) + 255) / 256), dim3(256), 0, st, b->b1, b->W2, b->H,
                       b->Hp, b->C, b->b1p_buf, b->W2p_buf);
    LT_CHECK_LAUNCH();
    b->b1p = b->b1p_buf;
    b->W2p = b->W2p_buf;
    return LT_OK;
}

// The borrowed inputs changed: everything derived from them is stale and is recomputed by the first call that reads it,
// on THAT call's stream -- S1 = X*W1 by FULL / SPARSE rows and the logits, the fp32 layers by SPARSE rows and the logits,
// the fp64 pre-activation (and nothing in fp32) by DELTA rows.  Launches nothing.
extern "C" int lt_baseline_refresh(lt_baseline *b, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline_refresh: baseline is NULL");
    (void)stream;
    b->s1_fresh = false;
    b->pad_fresh = false;
    b->layers_fresh = false;
    b->fp64_fresh = false;
    return LT_OK;
}

int lt_baseline_ensure_padding(const lt_baseline *cb, hipStream_t st) {
    lt_baseline *b = const_cast<lt_baseline *>(cb);   // cache state only
    if (b->pad_fresh) return LT_OK;
    const int rc = refresh_padding(b, st);
    if (rc) return rc;
    b->pad_fresh = true;
    return LT_OK;
}

int lt_baseline_ensure_s1(const lt_baseline *cb, hipStream_t st) {
    lt_baseline *b = const_cast<lt_baseline *>(cb);   // cache state only
    if (b->s1_fresh || b->n == 0) return LT_OK;
    const int rc = prepare_layer_inputs(b->n, b->X, b->ldx, b->F, b->W1, b->b1, b->H, b->W2, b->C, b->S1,
                                        b->b1p_buf, b->W2p_buf, &b->b1p, &b->W2p, b->slabs, st);
    if (rc) return rc;
    b->s1_fresh = true;
    b->pad_fresh = true;
    return LT_OK;
}

// Rows [row_begin, row_end) of X*W1 into dst[(row_end - row_begin), Hp] -- the sharded baseline of a multi-GPU run:
// every rank computes its slice, one all-gather rebuilds S1 (the caller's collective, into the storage attached
// with lt_baseline_attach_s1).  The split-K slicing is the one of the FULL product, so a row has the same bits
// whichever rank, and however many ranks, computed it.
extern "C" int lt_baseline_refresh_rows(lt_baseline *b, int32_t row_begin, int32_t row_end, float *dst, void *stream) {
    LT_REQUIRE(b != nullptr, "lt_baseline_refresh_rows: baseline is NULL");
    LT_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= b->n,
               "lt_baseline_refresh_rows: rows [%d, %d) outside [0, %d]", row_begin, row_end, b->n);
    LT_REQUIRE(dst != nullptr && ((uintptr_t)dst % 16) == 0, "lt_baseline_refresh_rows: dst is NULL or not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    b->layers_fresh = false;
    b->fp64_fresh = false;
    // b1 / W2 padding follows the weights exactly as in lt_baseline_refresh; the rest of S1 is the caller's all-gather
    const int rcp = refresh_padding(b, st);
    if (rcp) return rcp;
    b->pad_fresh = true;
    b->s1_fresh = true;      // this rank's rows now, the others by the caller's all-gather
    const int m = row_end - row_begin;
    if (m == 0) return LT_OK;
    if (b->Hp != b->H) LT_HIP(hipMemsetAsync(dst, 0, (size_t)m * b->Hp * sizeof(float), st));
    const float *A = b->X + (size_t)row_begin * b->ldx;
    if (b->slabs)
        return lt_launch_gemm_splitk(A, b->ldx, b->W1, b->H, dst, b->Hp, m, b->H, b->F, lt_gemm_pick_kslice(b->n, b->H, b->F),
                                     b->slabs, st);
    return lt_launch_gemm(A, b->ldx, b->W1, b->H, dst, b->Hp, m, b->H, b->F, st);
}

// The baseline reads S1 = X*W1 from caller-owned storage from now on ([>= n, Hp] fp32, ld == Hp, 16-byte aligned;
// e.g. a torch tensor that is the output of the ranks' all-gather).  The current S1 is copied over.
extern "C" int lt_baseline_attach_s1(lt_baseline *b, float *S1, int64_t ld, void *stream) {
    LT_REQUIRE(b != nullptr && S1 != nullptr, "lt_baseline_attach_s1: NULL argument");
    LT_REQUIRE(ld == b->Hp, "lt_baseline_attach_s1: ld=%lld, must equal the padded hidden width %d", (long long)ld, b->Hp);
    LT_REQUIRE(((uintptr_t)S1 % 16) == 0, "lt_baseline_attach_s1: storage must be 16-byte aligned");
    if (S1 == b->S1) return LT_OK;
    const int rce = lt_baseline_ensure_s1(b, (hipStream_t)stream);
    if (rce) return rce;
    if (b->n > 0)
        LT_HIP(hipMemcpyAsync(S1, b->S1, (size_t)b->n * b->Hp * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (b->S1_owned) {
        LT_HIP(hipStreamSynchronize((hipStream_t)stream));   // the copy reads the buffer freed below
        (void)hipFree(b->S1);
    }
    b->S1 = S1;
    b->S1_owned = false;
    return LT_OK;
}

// Z1 / S2 / OUT (and the fp64 pre-activation when enabled and asked for) from the current S1, if stale
int lt_baseline_ensure_layers(const lt_baseline *cb, bool need_fp64, hipStream_t st, bool need_fp32) {
    lt_baseline *b = const_cast<lt_baseline *>(cb);   // cache state only: logically const for the caller
    if (b->n == 0) return LT_OK;
    int rc = lt_baseline_ensure_padding(b, st);
    if (rc) return rc;
    if (need_fp32) {
        rc = lt_baseline_ensure_s1(b, st);
        if (rc) return rc;
        if (!b->layers_fresh) {
            rc = lt_launch_layer1(b->g, b->S1, b->Hp, b->b1p, b->W2p, b->C, b->Z1, b->S2, st, b->seg_part);
            if (rc) return rc;
            rc = lt_launch_layer2(b->g, b->S2, b->C, b->b2, b->OUT, st);
            if (rc) return rc;
            b->layers_fresh = true;
        }
    }
    if (need_fp64 && b->Z1d && !b->fp64_fresh) {
        rc = lt_baseline_refresh_fp64(b, st);
        if (rc) return rc;
        b->fp64_fresh = true;
    }
    return LT_OK;
}

extern "C" int lt_baseline_create(const lt_graph *g, const float *X, int64_t ldx, int32_t F,
                                  const float *W1, const float *b1, int32_t H, const float *W2,
                                  const float *b2, int32_t C, void *stream, lt_baseline **out) {
    LT_REQUIRE(out != nullptr, "lt_baseline_create: out is NULL");
    *out = nullptr;
    int rc = check_dims("lt_baseline_create", g, F, H, C);
    if (rc) return rc;
    LT_REQUIRE(X && W1 && b1 && W2 && b2, "lt_baseline_create: NULL tensor pointer");
    LT_REQUIRE(ldx >= F, "lt_baseline_create: ldx=%lld < F=%d", (long long)ldx, F);
    (void)lt_node_err_dev();      // the mapped flag words of the node-id check exist before any call that may be captured into a hipGraph
    lt_baseline *b = new (std::nothrow) lt_baseline();
    if (!b) return lt_set_error(LT_ERR_NOMEM, "lt_baseline_create: out of host memory");
    b->g = g; b->n = g->n; b->F = F; b->H = H; b->C = C; b->Hp = lt_round_up(H, 4);
    b->X = X; b->ldx = ldx; b->W1 = W1; b->b1 = b1; b->W2 = W2; b->b2 = b2;
    const size_t nh = (size_t)(b->n > 0 ? b->n : 1) * b->Hp * sizeof(float);
    const size_t nc = (size_t)(b->n > 0 ? b->n : 1) * C * sizeof(float);
#define B_HIP(call)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            free_baseline(b);                                                               \
            return lt_set_error(LT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)
    B_HIP(hipMalloc((void **)&b->S1, nh));
    B_HIP(hipMalloc((void **)&b->Z1, nh));
    B_HIP(hipMalloc((void **)&b->S2, nc));
    B_HIP(hipMalloc((void **)&b->OUT, nc));
    B_HIP(hipMalloc((void **)&b->b1p_buf, (size_t)b->Hp * sizeof(float)));
    B_HIP(hipMalloc((void **)&b->W2p_buf, (size_t)b->Hp * C * sizeof(float)));
    if (g->p_n_seg > 0) B_HIP(hipMalloc((void **)&b->seg_part, (size_t)g->p_n_seg * b->Hp * sizeof(float)));
    // hub rows of FULL stage A run on a side stream next to the plain rows (fork / join by events inside lt_influence_rows):
    // created here so that the launch functions create nothing and stay capturable
    B_HIP(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    B_HIP(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    B_HIP(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    if (lt_gemm_splitk_slab_bytes(b->n, H, F, lt_gemm_pick_kslice(b->n, H, F)))
        B_HIP(hipMalloc((void **)&b->slabs, lt_gemm_splitk_slab_bytes(b->n, H, F, lt_gemm_pick_kslice(b->n, H, F))));
#undef B_HIP
    rc = lt_baseline_refresh(b, stream);
    if (rc) {
        free_baseline(b);
        return rc;
    }
    *out = b;
    return LT_OK;
}

extern "C" int lt_baseline_destroy(lt_baseline *b) {
    free_baseline(b);
    return LT_OK;
}

extern "C" int lt_baseline_logits(const lt_baseline *b, float *dst, void *stream) {
    LT_REQUIRE(b != nullptr && dst != nullptr, "lt_baseline_logits: NULL argument");
    if (b->n == 0) return LT_OK;
    const int rc = lt_baseline_ensure_layers(b, false, (hipStream_t)stream);
    if (rc) return rc;
    LT_HIP(hipMemcpyAsync(dst, b->OUT, (size_t)b->n * b->C * sizeof(float), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    return LT_OK;
}
