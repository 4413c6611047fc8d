// Internal declarations shared by the gfx950 translation units of liblinkteller_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "linkteller_hip.h"

#define LT_MAX_H 256   // hidden width handled by one pass of the row kernels (64 lanes x float4)
#define LT_MAX_C 8     // classes handled by the fused layer-2 epilogue

struct lt_graph {
    int32_t n = 0;
    int64_t nnz = 0;
    int32_t max_row_nnz = 0;
    int32_t max_col_nnz = 0;
    float local_frac = 0.f;   // share of the entries whose column lies within LT_LOCAL_WINDOW rows of their row
    float hot_frac = 0.f;     // share of the entries that read one of the LT_HOT_COLUMNS most-read columns (skew)
    // CSR of A_hat (device)
    int32_t *rowptr = nullptr;
    int32_t *col = nullptr;
    float *val = nullptr;
    // CSC of A_hat = CSR of A_hat^T (device): for column v, the rows r with A_hat[r,v] != 0
    int32_t *tptr = nullptr;
    int32_t *trow = nullptr;
    float *tval = nullptr;
    int2 *cv = nullptr;        // [nnz + pad] (col, val bits) interleaved, graphs that take the tiled SpMM (lt_tiled_wanted at 256 columns): one
                               // request per entry for k_rows_tiled instead of two
    int32_t *tpos = nullptr;   // [nnz] position of the CSC entry inside its ROW (k - rowptr[r]); graphs of up to 65534 nodes only
    // the fused DELTA route's per-node incidence records (lt_items.hip.h; graphs of up to 65534 nodes without hub rows whose
    // largest record fits LDS), or NULL
    int4 *dl_meta = nullptr;      // [n] (offset into dl_rec in words, items, touched nodes, incidences)
    int32_t *dl_rec = nullptr;
    int32_t dl_max_t = 0, dl_max_tu = 0;   // largest incidence / touched-node count of a node
    // Long rows (hubs).  A row of more than LT_ROW_SEG entries is summed segment by segment in EVERY kernel
    // (lt_rows.hip.h row_dot: 128-entry fmaf chains, their sums added in segment order), which lets any kernel hand
    // the segments of a hub row to separate waves and still produce the same bits: the SpMM / layer-1 segment
    // kernels, FULL stage A (k_full_stageA_lds segment mode + k_full_long_combine).
    int32_t p_n_long = 0, p_n_seg = 0;
    int32_t *p_long_row = nullptr;     // [p_n_long]     row id
    int32_t *p_long_segptr = nullptr;  // [p_n_long + 1] first segment of each long row
    int32_t *p_seg_long = nullptr;     // [p_n_seg]      index into p_long_row
    int32_t *p_seg_begin = nullptr;    // [p_n_seg]      first CSR entry of the segment
    float *p_seg_scratch = nullptr;    // [p_n_seg, LT_MAX_H] segment sums (standalone SpMM, lt_gcn2_forward; one stream at a time)
    // The fp64 row kernel's own cut (k_spmm_f64, round 5): rows of more than LT_F64_LONG entries as LT_F64_SEG-entry segments.  An
    // fp64 sum may be associated freely (the routes agree to ~1e-16), and a lane group walks its entries 16 at a time: the rows of
    // 64 .. 128 entries of a power-law graph were 4 .. 8 dependent trips each and the launch's longest path (23.4 -> 18.1 us on the
    // co-headline graph with 32-entry pieces).  Same four arrays as above; the fp32 kernels never see them.
    int32_t q_n_long = 0, q_n_seg = 0;
    int32_t *q_long_row = nullptr, *q_long_segptr = nullptr, *q_seg_long = nullptr, *q_seg_begin = nullptr;
    // Work items of the tiled SpMM (lt_spmm.hip), built for every graph: one item per row of up to LT_ROW_SEG entries
    // and one per segment of a long row.  Order: the segments first, by the column their first entry reads (waves
    // that run at the same time then gather from one sliding window of S), then the short rows, longest first.
    int32_t w_n = 0;
    int32_t *w_e0 = nullptr;    // [w_n] first CSR entry
    int32_t *w_cnt = nullptr;   // [w_n] entries (<= LT_ROW_SEG)
    int32_t *w_dst = nullptr;   // [w_n] row id, or n + segment id (index into the p_seg_* tables)
};
#define LT_HOT_COLUMNS 16384   // one XCD L2 (4 MiB) holds this many 256-byte row slices
#define LT_LOCAL_WINDOW 8192   // |col - row| up to this counts as a local entry (8192 rows x 1 KiB = two XCD L2s)
#define LT_ROW_SEG 128   // layer-1 chains: entries per segment (rows up to this length are one plain chain)
#define LT_F64_LONG 64   // k_spmm_f64: rows of more entries than this are cut ...
#define LT_F64_SEG 32    // ... into segments of this many (fp64 chains only)
static inline int lt_f64_seg_rows(const lt_graph *g) { return g->p_n_seg > g->q_n_seg ? g->p_n_seg : g->q_n_seg; }   // rows of an fp64 segment scratch
#define LT_CSR_PAD 16   // zero entries appended to col/val

struct lt_baseline {
    const lt_graph *g = nullptr;
    int32_t n = 0, F = 0, H = 0, C = 0;
    int32_t Hp = 0;  // H rounded up to a multiple of 4: leading dimension of S1 / Z1 (pad columns are 0)
    // borrowed
    const float *X = nullptr;
    int64_t ldx = 0;
    const float *W1 = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr;
    // owned (device)
    float *S1 = nullptr;   // [n, Hp]  X * W1 (caller-owned after lt_baseline_attach_s1)
    bool S1_owned = true;
    float *Z1 = nullptr;   // [n, Hp]  A_hat * S1 + b1   (pre-activation)
    float *S2 = nullptr;   // [n, C]   relu(Z1) * W2
    float *OUT = nullptr;  // [n, C]   A_hat * S2 + b2   (baseline logits)
    // b1 [Hp] / W2 [Hp, C] with zero padding up to Hp: the caller's own tensors when H is a multiple of 4 and
    // b1 is 16-byte aligned (no copy), else the zero-padded copies below (refreshed with the baseline)
    const float *b1p = nullptr;
    const float *W2p = nullptr;
    float *b1p_buf = nullptr;  // [Hp]
    float *W2p_buf = nullptr;  // [Hp, C]
    float *slabs = nullptr;  // split-K partials of X*W1 (only when the product is K-sliced)
    float *seg_part = nullptr;  // [g->p_n_seg, Hp] segment sums of the long rows of Z1 (SPARSE recomputes one segment of a hub row)
    // optional fp64-accumulated copies for the kink test of LT_MODE_DELTA (lt_baseline_enable_fp64)
    double *S1d = nullptr;      // [n, Hp]
    double *Z1d = nullptr;      // [n, Hp]
    double *slabs_d = nullptr;  // split-K partials
    int8_t *i8_wd = nullptr;    // W1's signed base-256 digits and ...
    bool i8_ew_clean = false;   // i8_ew holds zeros (the last product's slab sum cleared it)
    unsigned *i8_ew = nullptr;  // ... the exponents of its (column, K slice) scales: the int8 split of the dense fp64 product (lt_i8_split.hip.h)
    double *seg_d = nullptr;    // [g->p_n_seg, Hp] fp64 segment sums of the long rows
    bool S1d_owned = true;      // false after lt_baseline_attach_s1d: S1d is caller storage filled by the ranks' all-gather
    bool S1d_external = false;  // the fp64 product arrives from outside (lt_baseline_refresh_rows_fp64 + the caller's all-gather)
    // feature-difference route of the fp64 product (lt_fp64.hip, k_s1d_feature_rows)
    float *fd_ref = nullptr;    // [F] the reference vector m (majority value of each column over the first rows)
    bool fd_ref_valid = false;
    double *fd_rs = nullptr;    // [n] row sums of A_hat in fp64 (deferred cref: Z1d[r] += rs[r] * cref)
    float *S1x = nullptr;       // [n, Hp] the feature route's product rows, fp64-accumulated and rounded once to fp32 (s1_f32)
    mutable bool s1_f32 = false;          // the current product lives in S1x (fp32) instead of S1d
    mutable bool cref_deferred = false;   // S1d currently holds S1d - cref: its readers add fd_cref themselves
    double *fd_cref = nullptr;  // [Hp] its product m W1
    double *fd_slabs = nullptr; // [ceil(F / 64), H] its split-K partials
    int *fd_gate = nullptr;     // device words: the slice counter of k_ref_row_product; the ring kernel's row counters (FR_GATE_WORDS)
    int fd_ring_parity = 0;     // which of the two counter sets the next launch of k_s1d_feature_ring uses
    int *fd_hint_host = nullptr, *fd_hint_dev = nullptr;   // mapped host word the feature kernel sets when it meets dense rows
    int feat_sparse = -1;       // what the probe at lt_baseline_enable_fp64 found: 1 sparse differences, 0 dense, -1 not probed
    // aggregate-first route of the fp64 pre-activation (lt_fp64.hip): Z1d[r] = (A_hat X)[r] W1 + b1 on the rows a call's
    // probes reach, nothing for the others -- no n x F x H product, no S1d
    int Fp = 0;                 // F rounded up to a multiple of 4 (leading dimension of Yd)
    bool agg_default = false;   // the route lt_baseline_enable_fp64 chose for these shapes / features
    bool no_agg = false;        // set before lt_baseline_enable_fp64: never the aggregate-first route (lt_gcn3.hip needs S1d / all rows)
    double *Yd = nullptr;       // [n, Fp] (A_hat X) rows, valid where zstate != 0
    double *seg_y = nullptr;    // [g->p_n_seg, Fp] segment sums of the long rows
    int32_t *zstate = nullptr;  // [n] 0: Z1d row not computed since the last refresh, 2: wanted by the chunk in flight, 1: valid
    int32_t *zrows = nullptr;   // [n] the rows marked 2 by the chunk in flight
    int32_t *zcount = nullptr;  // [1] their number
    int32_t *zitems = nullptr;  // [lt_xf64_scratch_words] aggregate-first route: the work items of the rows marked 2, compacted in list order
    int32_t *zicount = nullptr; // [1] their number
    // lt_baseline_refresh recomputes S1 and marks what depends on it stale; Z1 / S2 / OUT (and Z1d) are
    // recomputed by the first call that reads them (logits, SPARSE / DELTA rows) -- FULL rows never do: their
    // stage A yields the baseline S2 as a by-product and stage B forms the baseline logits itself.
    mutable bool s1_fresh = false;      // S1 matches the borrowed inputs
    mutable bool pad_fresh = false;     // b1p / W2p match the borrowed weights
    mutable bool layers_fresh = false;
    mutable bool fp64_fresh = false;
    mutable bool z_all_valid = false;   // every row of Z1d matches S1d (else: zstate per row)
    double *S1qs = nullptr;             // [n] the scale of row i of S1x: that array then holds int32 q with S1d - cref = q * S1qs[i] (see k_spmm_f64)
    float *Z1x = nullptr;               // feature route with fp32 row storage: the pre-activation rounded ONCE to fp32 (see k_spmm_f64)
    mutable bool z1x_valid = false;     // ... is what the last all-rows pass wrote (instead of Z1d)
    // FULL rows on a graph with hub rows: the segment kernel + combine run on `side` next to the plain-row
    // kernel (fork after the probe-row GEMM, join before stage B).  Created on first use.
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

int lt_set_error(int code, const char *fmt, ...);

// Tuning knobs (lt_core.hip): defaults come from the environment (LT_* variables, read once), lt_set_tuning
// overrides them at run time.  Every setting gives bit-identical results; they select between kernel routes.
struct lt_tuning {
    long long tiled_min_bytes;   // S of at least this many bytes -> tiled SpMM / layer 1 (LT_SPMM_TILED_MIN_BYTES)
    long long chunk_budget;      // bytes of per-probe scratch per probe chunk (LT_CHUNK_BUDGET_BYTES)
    int full_p;                  // probes per wave of FULL stage A: 8 / 16 / 32, 0 = from the probe count (LT_FULL_P)
    int long_par;                // hub rows of FULL stage A: 1 segment-parallel, 0 one wave per row, -1 by size (LT_LONG_PAR)
    int overlap;                 // hub-row kernels on the baseline's side stream (LT_OVERLAP)
    int item_bits;               // SPARSE / DELTA stage B membership bitmap (LT_ITEM_BITS)
    int wide_min_hp;             // smallest padded hidden width served by the batched stage-A kernel (LT_WIDE_MIN_HP)
    int probe_kslice;            // K-slice of the perturbed-row GEMM, 0 = the baseline product's slicing (LT_PROBE_KSLICE)
    long long pair_marks;        // SPARSE / DELTA stage B: chunks of at least this many (probe, observed) pairs -- and any call without
                                 // a membership bitmap -- find the affected pairs through the middle-node join (k_pm_*); 0 = always,
                                 // < 0 = never (LT_PAIR_MARKS)
    int hub_short_side;          // SPARSE / DELTA stage B, observed hubs: 1 short-side search, 0 per-entry tests, -1 by bitmap (LT_HUB_SHORT_SIDE)
    long long bits_max_bytes;    // SPARSE / DELTA: a bitmap row per probe only while the chunk's rows fit this (default 128 MiB);
                                 // beyond it only the chunk's big probes get rows (LT_BITS_MAX_BYTES)
    int tiled_big;               // 1: the tiled SpMM always uses 64-bit gather offsets (test hook; default: only when S spans >= 4 GiB)
    int s1_f32;                  // feature-difference route (with defer_cref): 1 the product rows are stored in fp32 (fp64-accumulated,
                                 // rounded once), 0 in fp64 (LT_S1_F32)
    int defer_cref;              // feature-difference route: 1 the reference vector's product rides in the rows' launch and is added by the
                                 // readers of S1d, 0 it is formed first and added by the rows kernel (LT_DEFER_CREF)
    int z_on_demand;             // fp64 pre-activation rows of the S1d routes: 1 only the rows a call reads, 0 all rows, -1 by the call's
                                 // size (LT_Z_ON_DEMAND)
    int stageb_rows;             // SPARSE / DELTA stage B with a bitmap: 1 one block per (observed row, probe slice), 0 one 8-lane group
                                 // per pair (LT_STAGEB_ROWS)
    int aggregate_first;         // fp64 pre-activation as (A_hat X) W1 on the rows a call needs: 0 never, 1 whenever the shapes allow,
                                 // -1 when they do and the features are not sparse differences (LT_AGGREGATE_FIRST)
    int feature_delta;           // fp64 product X*W1 from the feature rows' differences to a reference row: 0 never, 1 always try,
                                 // -1 when the baseline's features were found to be sparse differences (LT_FEATURE_DELTA)
    int delta_fused;             // DELTA at twitch size on graphs without hub rows: 1 stage A + B of a probe in one block
                                 // (k_delta_probe_block), 0 the item kernels (LT_DELTA_FUSED)
    int records_early;           // DELTA fused route: 1 the first chunk's record blocks ride in the product rows' launch when it runs (default),
                                 // 0 in the pre-activation's launch as in round 4 (LT_RECORDS_EARLY)
    int xf64_blocks;             // aggregate-first gathers (k_rows_tiled_xf64): blocks per XCD that walk the compacted work items (default 96:
                                 // more gathers in flight than the L2s hold windows for cost more than idle CUs; tuned on BASELINE
                                 // configs[4], profiles/r05_xf64_sweep.txt) (LT_XF64_BLOCKS)
    int feature_ring;            // feature-difference route: the persistent LDS-ring form of the rows kernel (lt_feature_ring.hip.h): 0 never
                                 // (default: it measured 25.6 us against 22.0 at twitch size, profiles/r06_ring_lab.txt), 1 whenever the shapes
                                 // allow, -1 when they do and there are >= feature_ring_min_rows rows (LT_FEATURE_RING)
    int feature_flags;           // the row-per-wave kernel lists a row's differing columns from flag bits (1, default: the kernel is bound by
                                 // the issue of its 52 compare steps -- 24.8 against 25.4 us by events, profiles/r06_feat_lab_timeline.txt) or by
                                 // a ballot per value (0: round 5's list order) (LT_FEATURE_FLAGS)
    int pair_list;               // SPARSE / DELTA stage B with pair marks: 1 the marked pairs are compacted into a list and walked 8 to a wave (default),
                                 // 0 every pair's group reads its mark (LT_PAIR_LIST)
    int i8_split;                // dense-feature fp64 product: 1 the error-free split on the int8 matrix cores (default), 0 the f64 cores (LT_I8_SPLIT)
    int gcn3_product_gather;     // GCN3 `delta`: 1 the probes' fp64 product rows are read off the inner baseline's product (default), 0 formed again
                                 // as X[probes] W1 on the f64 cores (LT_GCN3_PRODUCT_GATHER)
    int export_sparse;           // lt_influence_rows_f64, fused route behind a refresh: 1 the first "export_zero_share" % rows of the float64
                                 // matrix are zero-filled by blocks of the product rows' launch and their probes' blocks write the touched
                                 // positions only (default), 0 every block widens its whole row (LT_EXPORT_SPARSE)
    int export_zero_share;       // (LT_EXPORT_ZERO_SHARE, 0 .. 100)
    int export_zero_share2;      // the next so many % of the rows, by blocks of the pre-activation's launch (LT_EXPORT_ZERO_SHARE2)
    int export_zero_blocks;      // waves that zero-fill (LT_EXPORT_ZERO_BLOCKS)
    int export_zero_inflight;    // stores each of them keeps in flight (LT_EXPORT_ZERO_INFLIGHT)
    int feature_stagger;         // the row-per-wave kernel's blocks start in (value & 255) groups, (value >> 8) ticks of 10 ns apart; 0 = together
                                 // (LT_FEATURE_STAGGER)
    int feature_ring_min_rows;   // (LT_FEATURE_RING_MIN_ROWS, default 1024: below it the CUs' waves have no row each)
    int profile_every;           // lt_profile_enable: bracket every N-th scope of an enabled class with events (1 = all; an event pair
                                 // costs ~5 us of stream time, so a timed region samples)
};
lt_tuning &lt_tune();
// the float64 export (lt_core.hip): a dst pointer's device-side alias (pinned host memory) / itself (device memory); the launch
int lt_export_resolve(double *dst, double **dev, const char *who);
int lt_export_rows_dev(const float *src, int64_t lds, int32_t rows, int32_t cols, double *dst_dev, int64_t ldd, hipStream_t stream);

#define LT_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return lt_set_error(LT_ERR_HIP, "%s failed: %s (%s:%d)", #call,                \
                                hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define LT_CHECK_LAUNCH()                                                                  \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return lt_set_error(LT_ERR_HIP, "kernel launch failed: %s (%s:%d)",            \
                                hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define LT_REQUIRE(cond, ...)                                                              \
    do {                                                                                   \
        if (!(cond)) return lt_set_error(LT_ERR_INVALID, __VA_ARGS__);                     \
    } while (0)

static inline int lt_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t lt_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel event timing (lt_core.hip) ------------------------------------------
extern unsigned g_lt_profile_mask;
bool lt_profile_sample(int kernel_id);      // "profile_every" = N: true for every N-th scope of a class (1: all of them)
void lt_profile_call_begin();                // entry / exit of a probe-primitive call: "profile_every" samples whole calls
void lt_profile_call_end();
struct lt_prof_call { lt_prof_call() { lt_profile_call_begin(); } ~lt_prof_call() { lt_profile_call_end(); } };
void lt_profile_begin(int kernel_id, hipStream_t st);
void lt_profile_end(int kernel_id, hipStream_t st);
struct lt_prof_scope {
    int id; hipStream_t st; bool on;
    lt_prof_scope(int id_, hipStream_t st_, bool wanted = true)
        : id(id_), st(st_), on(wanted && ((g_lt_profile_mask >> id_) & 1u) && lt_profile_sample(id_)) { if (on) lt_profile_begin(id, st); }
    ~lt_prof_scope() { if (on) lt_profile_end(id, st); }
};

// node ids out of range (lt_core.hip): the device-side alias of the two mapped flag words (NULL: unavailable -- ids are still
// clamped), and the entry check of the probe primitives (LT_ERR_INDEX when an earlier call raised a flag)
int32_t *lt_node_err_dev();
int lt_node_err_pending();

int lt_baseline_refresh_fp64(lt_baseline *b, hipStream_t st);
// what a call needs of the baseline, recomputed here if lt_baseline_refresh marked it stale: need_fp32 = S1 and the
// fp32 layers Z1 / S2 / OUT, need_fp64 = the fp64 pre-activation (when enabled); b1p / W2p always
int lt_baseline_ensure_layers(const lt_baseline *b, bool need_fp64, hipStream_t st, bool need_fp32 = true);
int lt_baseline_ensure_s1(const lt_baseline *b, hipStream_t st);
int lt_baseline_ensure_padding(const lt_baseline *b, hipStream_t st);
void lt_baseline_free_fp64(lt_baseline *b);

// ---- launchers implemented in the kernel translation units --------------------------------
// layer 1 for all rows: Z1 = A_hat*S1 + b1 (optional store), S2 = relu(Z1)*W2
// seg_part: [g->p_n_seg, Hp] scratch for the segment sums of the long rows (NULL: the graph's own scratch)
int lt_launch_layer1(const lt_graph *g, const float *S1, int Hp, const float *b1p,
                     const float *W2p, int C, float *Z1_or_null, float *S2, hipStream_t st,
                     float *seg_part = nullptr);
// layer 2 for all rows: OUT = A_hat*S2 + b2
// Tiled SpMM over the graph's work items (lt_spmm.hip): chains start from `init` (NULL = 0) for short rows and for
// the first segment of a long row; short rows get `+ bias_after` (NULL = none) and the optional ReLU and go to
// out[row]; segment sums go raw to seg_out[segment] for the caller's ordered combine.
bool lt_tiled_wanted(const lt_graph *g, int ncols);
int lt_launch_rows_tiled(const lt_graph *g, const float *S, int64_t lds, int ncols, const float *init,
                         const float *bias_after, int relu, float *out, int64_t ldo, float *seg_out,
                         int64_t ld_seg, hipStream_t st);
// fp64 twin (S, out, seg_out double; chains from zero, + bias_after on short rows): the pre-activation of LT_MODE_DELTA
size_t lt_xf64_scratch_words(const lt_graph *g);      // int32 words of the `zitems` scratch below
int lt_launch_rows_tiled_xf64(const lt_graph *g, const float *X, int64_t ldx, int ncols, double *out, int64_t ldo,
                              double *seg_out, int64_t ld_seg, const int32_t *state, int32_t *zitems, int32_t *zicount,
                              hipStream_t st);
// aggregate-first route active for this baseline right now?  lt_fp64_prepare_items: per probe chunk of an LT_MODE_DELTA call,
// the fp64 pre-activation rows the chunk's items read (Z1d) and the probes' own fp64 product rows Spd[nb, Hp]
bool lt_fp64_agg_active(const lt_baseline *b);
int lt_fp64_form_all(lt_baseline *b, hipStream_t st);
int lt_launch_gemm_f64_dense(const double *A, long lda, int M, const float *B, long ldb, int N, int K, const float *bias,
                             double *C, long ldc, int relu_a, hipStream_t st);
int lt_fp64_product_rows_gather(const lt_baseline *b, const int32_t *rows, int m, double *C, long ldc, hipStream_t st);   // 1: no product rows held
int lt_launch_gemm_f64_gather(const float *A, long lda, const int32_t *rows, int M, const float *B, long ldb, int N, int K,
                              double *C, long ldc, hipStream_t st);
int lt_launch_spmm_f64(const lt_graph *g, const double *S, int ld, const float *biasp, double *out, double *seg_d, hipStream_t st);
// The arguments of k_item_bits (lt_items.hip.h) as a value: lt_fp64_prepare_rows can let the item tables of a probe chunk ride
// in the launch that forms the pre-activation (extra blocks of k_spmm_f64) instead of a launch of their own in front of it.
struct lt_bits_job {
    const int32_t *tptr, *trow, *probes;
    int nb, words;
    uint2 *bits;
    int32_t *off;
    int2 *item_pr;
    uint2 *big_bits;
    int32_t *big_slot, *big_count;
    const int32_t *rowptr, *observe;
    int n_obs;
    int32_t *hub_obs;
    int nblocks;      // nb, + 1 when hub_obs is wanted; 0 = no job
    const float *tval;      // with item_va: (probe node, A_hat[r, v]) of every item next to (probe index, row)
    int2 *item_va;
    // dl_rec != NULL: the job gathers the incidence records of the chunk's probes instead (the fused DELTA route, lt_items.hip.h
    // delta_record_block: one block per probe); uses probes / nb
    int32_t *dl_rec;
    const int4 *dl_meta;
    const int32_t *dl_src;
    int dl_maxc, dl_rec_words;
    // node-id check (lt_items.hip.h): ids outside [0, n) are replaced by 0 and flagged in err[0] (probes) / err[1] (observed); the
    // item-table form also writes the checked lists to probes_s [nb] / obs_s [n_obs] for the kernels behind it (NULL: lists
    // already checked).  n == 0: no check (the 3-layer path checks up front)
    int n;
    int32_t *err, *probes_s, *obs_s;
    // zero_blocks > 0: that many more blocks of the carrying launch fill rows [zero_row0, zero_row0 + zero_rows) of zero_dst (float64,
    // leading dimension zero_ld, columns [0, zero_cols)) with +0.0 -- rows of lt_influence_rows_f64's matrix, usually pinned host memory
    double *zero_dst;
    long zero_ld;
    int zero_row0, zero_rows, zero_cols, zero_blocks, zero_inflight;
    unsigned smem_bytes;    // dynamic LDS the job's blocks need (the launch that carries them must be given it)
};
// job != NULL: *job_done says whether the job went along (it does when every row is formed by one plain launch; the on-demand
// form needs the tables BEFORE, the caller then launches k_item_bits itself and calls again without a job)
bool lt_fp64_on_demand(const lt_baseline *b, int n_probe_call);
// the record blocks of a call's first probe chunk offered to the launch that forms the fp64 product rows (k_s1d_feature_rows), should
// the baseline have to run it: lt_fp64_offer_job before lt_baseline_ensure_layers, lt_fp64_offer_taken after (true: they went along)
void lt_fp64_offer_job(const lt_bits_job *job);
bool lt_fp64_offer_taken();
int lt_fp64_prepare_rows(const lt_baseline *b, const int32_t *off, int nb, const int2 *item_pr, int n_probe_call, hipStream_t st,
                         const lt_bits_job *job = nullptr, bool *job_done = nullptr);
int lt_fp64_prepare_items(const lt_baseline *b, const int32_t *off, int nb, const int2 *item_pr, const int32_t *probes,
                          double *Spd, hipStream_t st);
int lt_launch_rows_tiled_f64(const lt_graph *g, const double *S, int64_t lds, int ncols, const float *bias_after,
                             double *out, int64_t ldo, double *seg_out, int64_t ld_seg, hipStream_t st);
int lt_launch_layer2(const lt_graph *g, const float *S2, int C, const float *b2, float *OUT,
                     hipStream_t st);
int lt_launch_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                   int64_t ldc, int M, int N, int K, hipStream_t st);
int lt_launch_gemm_mdev(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                        int m_bound, const int32_t *m_dev, int N, int K, hipStream_t st);
size_t lt_gemm_splitk_slab_bytes(int M, int N, int K, int kslice);
int lt_gemm_pick_kslice(int M, int N, int K);   // baseline X*W1: slice length that fills the CUs in whole rounds
// gather_rows != NULL: row m of the A operand is row gather_rows[m] of A, perturbed by x + x * delta
int lt_launch_gemm_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                          int64_t ldc, int M, int N, int K, int kslice, float *slabs, hipStream_t st,
                          const int32_t *gather_rows = nullptr, float delta = 0.f);
