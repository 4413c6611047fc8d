/*
 * linkteller_hip.h -- C ABI of the MI355X (gfx950) implementation of LinkTeller's
 * influence-analysis hot path.
 *
 * The reference (AI-secure/LinkTeller) is pure Python/PyTorch and has no FFI/plugin layer
 * (SURVEY.md section 8b): its "operator interface" for this path is four PyTorch call
 * sites.  Each entry point below names the reference call site it stands in for
 * (file:line relative to the reference tree); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (LT_OK) or a negative lt_status; nothing throws across the
 *     boundary; lt_last_error() returns a thread-local description of the last failure.
 *   - "device" pointers are borrowed HIP device pointers (e.g. torch tensor data_ptr());
 *     the caller keeps them alive until the stream has drained.  All matrices are
 *     row-major fp32 with an explicit leading dimension in elements.
 *   - kernels are enqueued on the caller's stream (a hipStream_t passed as void*; NULL =
 *     the default stream) and never synchronise; workspaces are caller-provided so that
 *     the launch functions are hipGraph-capturable.
 *   - handles are not thread-safe: one handle per host thread / per rank.
 */
#ifndef LINKTELLER_HIP_H
#define LINKTELLER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 5   /* 5: lt_influence_rows_f64 (the probes' blocks write the float64 matrix themselves), the on-demand pre-activation by
                            row list (lt_baseline_form / gather / scatter_rows_fp64: hub rows shared between ranks); every version-4 entry
                            point is unchanged.  4: lt_export_rows_f64 (the matrix leaves the device once, as float64), node ids checked on the device
                            (LT_ERR_INDEX, lt_node_check), lt_profile_calls; every version-3 entry point is unchanged.  2: lt_baseline_refresh launches nothing (lazy recomputation on the first reader's stream); fp64 shard entry points;
                            profile classes 9-11.  3: lt_influence_rows_vec + lt_wide_combine (layers wider than one pass of the fused
                            kernels), lt_spmm_gather_ceiling (measurement support); every version-2 entry point is unchanged */

typedef enum lt_status {
    LT_OK = 0,
    LT_ERR_INVALID = -1,     /* bad argument / malformed CSR                          */
    LT_ERR_HIP = -2,         /* a HIP runtime call failed (message has the HIP error) */
    LT_ERR_UNSUPPORTED = -3, /* shape outside what the kernels are built for          */
    LT_ERR_WORKSPACE = -4,   /* caller workspace too small / misaligned               */
    LT_ERR_NOMEM = -5,
    LT_ERR_INDEX = -6        /* a node id of an EARLIER call's device lists was out of range (see lt_node_check) */
} lt_status;

/* how lt_influence_rows evaluates a probe (all three give the reference's quantity
 * || (f(X + d*e_v x_v^T) - f(X))[u] ||_2 / d, attacker.py:100-108,227-229)            */
typedef enum lt_influence_mode {
    LT_MODE_FULL = 0,   /* every probe: perturbed row GEMV + full SpMM over all N rows + ReLU + W2
                           + layer-2 SpMM on the observed rows, then the fp32 finite difference */
    LT_MODE_SPARSE = 1, /* bit-identical to FULL, but only rows whose value can differ from the
                           baseline (the 1-/2-hop set of the probe) are recomputed             */
    LT_MODE_DELTA = 2   /* propagates the perturbation itself (piecewise-linear through ReLU):
                           no cancellation, agrees with an fp64 evaluation of the reference      */
} lt_influence_mode;

typedef struct lt_graph lt_graph;       /* device-resident normalised adjacency (CSR + CSC) */
typedef struct lt_baseline lt_baseline; /* unperturbed forward state of a 2-layer GCN       */

const char *lt_last_error(void);
int lt_abi_version(void);
/* number of visible HIP devices (0 on a CPU-only box; never initialises a context) */
int lt_device_count(int *count);

/* ---- tuning knobs (no reference counterpart) ---------------------------------------------------
 * Every setting gives bit-identical results unless its entry says otherwise (the entries that choose another fp64 summation order
 * or storage form for LT_MODE_DELTA's product: results then agree to < 1e-6 of the largest score); the knobs choose between kernel
 * routes and are what the tests use to run each route on small inputs.  Defaults come from LT_* environment variables read once.
 *   "tiled_min_bytes"     S of at least this many bytes takes the tiled SpMM / layer-1 route (default 32 MiB)
 *   "chunk_budget_bytes"  per-call scratch budget that decides the probe chunking (default 1 GiB)
 *   "full_p"              probes per wave of FULL stage A: 8 / 16 / 32, 0 = from the probe count
 *   "long_par"            hub rows in FULL stage A: 1 = segments in separate waves, 0 = one wave per row
 *   "overlap"             hub-row kernels on the baseline's side stream (1) or on the caller's (0)
 *   "item_bits"           SPARSE / DELTA stage B membership bitmap on (1) / off (0)
 *   "hub_short_side"      SPARSE / DELTA stage B on observed hub rows: 1 = the members of row(u) and R_v are found from the
 *                         shorter list, 0 = every entry is tested against every probe, negative = by whether the call has a
 *                         bitmap row per probe (default)
 *   "bits_max_bytes"      SPARSE / DELTA stage B keeps a membership bitmap row per probe while a chunk's rows fit this many
 *                         bytes (default 128 MiB); larger calls give rows to their big probes only
 *   "pair_marks"          SPARSE / DELTA stage B: calls of at least this many (probe, observed) pairs per chunk -- and every
 *                         call too large for a membership bitmap -- find the affected pairs through a join over the
 *                         middle nodes; 0 = always, negative = never (default 2^22)
 *   "wide_min_hp"         smallest padded hidden width served by the batched stage-A kernel
 *   "tiled_big"           1 = the tiled SpMM uses its 64-bit gather offsets on any graph (test hook; default: S >= 4 GiB)
 *   "feature_delta"       fp64 product X*W1 of LT_MODE_DELTA from the differences of the feature rows to one reference row
 *                         (one pass over X when every column holds two values, as standardised indicator features do;
 *                         exact for any X): 0 = never, 1 = always try, negative = when lt_baseline_enable_fp64 found the
 *                         features to be of that kind (default).  The only knob that changes fp64 summation ORDER (the
 *                         results agree to ~1e-16 relative before the final rounding to fp32).
 *   "s1_f32"              feature-difference route with "defer_cref": 1 = the fp64-accumulated product rows are stored as 32-bit
 *                         fixed point with one scale per row (31 bits against the row's largest value; half the bytes the fp64
 *                         SpMM gathers) and the pre-activation in fp32 (default), 0 = both in fp64.  Moves `delta` results by
 *                         < 1e-6 of the largest score (plain fp32 rows moved them by up to 7e-5: DESIGN.md 5d)
 *   "delta_fused"         LT_MODE_DELTA on a graph with incidence records (lt_graph_create builds them for graphs of n <= 65534 nodes
 *                         without hub rows), calls whose pre-activation is formed on all rows: 1 = stage A and stage B of a probe in
 *                         ONE block, from the probe node's record matched against the observed list inside the pre-activation's
 *                         launch (default), 0 = the item kernels.  Bit-identical
 *   "records_early"       "delta_fused" route: 1 = the record blocks of a call's first probe chunk ride in the launch that forms the fp64
 *                         product rows whenever that launch runs (a refreshed baseline on the feature-difference route: CU slots to
 *                         spare, seven times the duration) (default), 0 = in the pre-activation's launch.  Bit-identical
 *   "feature_ring"        feature-difference route: the product rows by the persistent LDS-ring kernel (two workgroups per CU, a row in
 *                         flight by LDS-DMA ahead of the row a wave works on, rows claimed from counters; rows 8-byte aligned,
 *                         H % 4 == 0, 2046 <= F <= 3326): 0 = never (one wave per row; default -- the ring form measured 25.6 us
 *                         against 22.0 at twitch size, profiles/r06_ring_lab.txt), 1 = whenever the shapes allow, negative = when
 *                         they do and the matrix has at least "feature_ring_min_rows" rows.  fp64 summation order only (as with
 *                         the "feature_delta" knob); a probe chunk's record blocks then ride in the pre-activation's launch
 *   "feature_ring_min_rows"   see "feature_ring" (default 1024, >= 2)
 *   "pair_list"           SPARSE / DELTA stage B on calls that find their affected pairs by the join over the middle nodes ("pair_marks"): 1 = the
 *                         marked pairs are compacted into a list, the result rows zero-filled, and the pair kernel walks the list (default;
 *                         calls whose list would exceed 256 MiB keep the other form), 0 = every pair's lane group reads its own mark.
 *                         Bit-identical
 *   "i8_split"            the fp64 product X*W1 of LT_MODE_DELTA on DENSE features (n >= 256, F >= 256, H a multiple of 64): 1 = as an
 *                         error-free integer split on the int8 matrix cores (X five, W1 four signed base-256 digits, fourteen digit
 *                         pairs, exact integer sums, one rounding per K slice: rows within 5e-10 of their largest value of the fp64
 *                         product; default), 0 = on the f64 matrix cores.  Results agree to < 1e-6 of the largest score
 *   "gcn3_product_gather" lt_influence3_rows, LT_MODE_DELTA: 1 = the probes' fp64 product rows X[v] W1 are read off the product the baseline
 *                         already holds for every row (default), 0 = formed again on the f64 matrix cores (as on the aggregate-first
 *                         route).  fp64 summation order / storage only: results agree to < 1e-6 of the largest score
 *   "export_sparse"       lt_influence_rows_f64 on the fused LT_MODE_DELTA route, calls that find the baseline refreshed: 1 = the first
 *                         rows of dst are zero-filled by a few waves riding in the launch that forms the fp64 product rows and in the
 *                         pre-activation's (np.zeros of attacker.py:216 crossing PCIe under those launches) and their probes' blocks
 *                         write the touched positions only; the other rows are widened whole by their blocks (default), 0 = all rows
 *                         are.  Bit-identical; so are the four keys below
 *   "export_zero_share"   per cent of dst's rows zero-filled under the product rows' launch (default 35; 0 .. 100)
 *   "export_zero_share2"  per cent of dst's rows, the next ones, zero-filled under the pre-activation's launch (default 15; 0 .. 100)
 *   "export_zero_blocks"  waves that zero-fill in each of the two launches (default 16; 1 .. 4096: more of them, or more stores in
 *                         flight, and the writes queued for the link hold up the loads of the kernels beside them)
 *   "export_zero_inflight"   1-KiB stores each such wave keeps in flight (default 4; 1 .. 64)
 *   "feature_stagger"     feature-difference route, one wave per row: the row blocks start in (value & 255) groups, (value >> 8) x 10 ns
 *                         apart, so that a group walks its lists while the next one's rows arrive; 0 = all together.  Bit-identical
 *   "xf64_blocks"         aggregate-first route: blocks per XCD that walk the compacted work items of the rows a call reaches (default 96;
 *                         1 .. 4096).  Bit-identical
 *   "feature_flags"       feature-difference route, one wave per row: 1 = a row's differing columns are found as flag bits (plain VALU)
 *                         and listed level by level (default: the kernel is bound by the issue of its compare steps), 0 = by a
 *                         ballot per value as in round 5.  Changes the order of a row's list: fp64 summation order only
 *   "defer_cref"          feature-difference route: 1 = the reference vector's product m W1 is formed by extra blocks of the rows'
 *                         launch and added by the readers of S1d (the fp64 SpMM, stage A) (default), 0 = formed first and added
 *                         by the rows kernel.  fp64 summation order only, like "feature_delta"
 *   "z_on_demand"         LT_MODE_DELTA on the S1d routes: 1 = the fp64 pre-activation is formed only on the rows a call's items
 *                         read (they stay valid until the next refresh), 0 = on all rows at the first call after a refresh,
 *                         negative = by the call's size (default: on demand when probes x average column length < n / 2 and the
 *                         whole fp64 SpMM is worth avoiding, nnz x hidden width >= 2.5e8: below that it is a 10 us launch)
 *   "stageb_rows"         SPARSE / DELTA stage B of calls with a membership bitmap and no pair marks: 1 = one block per (observed
 *                         node, slice of the probes), the observed row staged in LDS (default), 0 = one 8-lane group per pair
 *   "aggregate_first"     fp64 pre-activation of LT_MODE_DELTA as Z1d[r] = (A_hat X)[r] W1 + b1, computed only on the rows the
 *                         probes of a call reach (no n x F x H product): 0 = never, 1 = whenever the shapes allow (F <= 2 Hp,
 *                         F <= 512; set it before lt_baseline_enable_fp64 allocates), negative = then and when the features
 *                         are not sparse differences (default).  Like "feature_delta" it changes fp64 summation order only.
 *   "profile_every"       lt_profile_enable brackets every N-th launch group of an enabled class with events (default 1 = all of them)
 *   "probe_kslice"        K-slice of the perturbed-row GEMM; 0 = the slicing of the baseline X*W1 (default: S1'[v] and
 *                         S1[v] then share one summation order, like the reference's two torch.mm calls)
 * value = LT_TUNING_DEFAULT restores the default. */
#define LT_TUNING_DEFAULT (-0x7fffffffffffffffLL - 1)
int lt_set_tuning(const char *key, long long value);

/* ---- graph ------------------------------------------------------------------------------
 * Stands in for utils/load.py:552-559 (sparse_mx_to_torch_sparse_tensor) + the .cuda() at
 * worker.py:665-678: takes the normalised adjacency A_hat as HOST CSR (int32 indices,
 * fp32 values, columns strictly increasing inside each row), validates it, and uploads it
 * to the current device together with its transpose (CSC) used by the sparse/delta modes.
 * Graphs of up to 65534 nodes without hub rows (rows of more than 128 entries) also get their per-node incidence records
 * (LT_MODE_DELTA's fused route, "delta_fused"): sum over the nodes of |column| x the columns' lengths entries of 8 bytes,
 * built on the host in this call (twitch-RU: 1.5 M entries, 25 MB); skipped when a node has more than 4096 (dense clusters: the item kernels are faster there) or they pass 256 MB. */
int lt_graph_create(int32_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                    const float *val, lt_graph **out);
int lt_graph_destroy(lt_graph *g);
int lt_graph_info(const lt_graph *g, int32_t *n, int64_t *nnz, int32_t *max_row_nnz);
/* The rows of at least min_entries entries that the probe nodes reach in one hop ({r : A_hat[r, v] != 0 for a probe v}), each once,
 * in no particular order: rows[0 .. *count) (device; capacity n), flags: device scratch of n int32 (overwritten).  What the ranks
 * of a multi-GPU run agree on (every rank runs it on the WHOLE probe list and sorts the result) before they split the hub rows
 * of the on-demand pre-activation between them (lt_baseline_form_rows_fp64 below). */
int lt_graph_reached_rows(const lt_graph *g, const int32_t *probes, int32_t n_probe, int32_t min_entries, int32_t *flags,
                          int32_t *rows, int32_t *count, void *stream);
/* Host only (no device call): the incidence records lt_graph_create would build for this CSR -- what tests/test_records.py pins
 * against a plain restatement.  meta [4 n]: per node (offset into rec in 32-bit words, items, touched nodes, incidences);
 * rec (may be NULL: sizes only): per node its items (row r, A_hat[r, v] as bits), then the touched nodes (u, first entry |
 * entries << 16), u ascending, then the entries (A_hat[u, r] as bits, item << 16 | position in row u) node by node in entry
 * order.  *rec_words: the words the records take.  LT_ERR_UNSUPPORTED when the graph gets none (hub rows, more than 65534
 * nodes, a node beyond the incidence cap, more than 256 MB), LT_ERR_WORKSPACE when rec_capacity (words) is too small. */
int lt_graph_records_host(int32_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const float *val,
                          int32_t *meta, int32_t *rec, int64_t rec_capacity, int64_t *rec_words);

/* ---- dense GEMM C[M,N] = A[M,K] * B[K,N]  (torch.mm at gcn/layers.py:31) ----------------
 * exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): each output is the ordered sum of k-ordered fmaf chains of 128 terms
 * (one chain per 128 k's, each started from +0: the rounding of a K = 3170 product stays at the level of a blocked CPU
 * sgemm, which is what keeps the fp32 finite difference of FULL / SPARSE inside the reference's own noise). */
int lt_gemm_f32(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                int32_t M, int32_t N, int32_t K, void *stream);

/* ---- SpMM out[n,ncols] = A_hat * S (+ bias) (ReLU)  (torch.spmm + bias at
 * gcn/layers.py:32-36, F.relu at gcn/models.py:20) ---------------------------------------
 * Any ncols (the reference has no width limit): columns are independent chains, so wide layers are served in slices of
 * 256 columns; the 16-byte vector path takes what is aligned (S / out / bias 16-byte aligned, lds and ldo multiples of
 * 4), a tail of ncols % 4 columns and unaligned operands go through an 8-lane kernel, 8 columns per launch.
 * Row-owned, fixed-order fmaf chains: deterministic and run-to-run reproducible.
 * lt_spmm_route: 1 when the call takes the tiled (column-sliced work-item) route on this graph, 0 for the row kernels. */
int lt_spmm_csr_f32(const lt_graph *g, const float *S, int64_t lds, int32_t ncols,
                    const float *bias_or_null, int32_t relu, float *out, int64_t ldo,
                    void *stream);
int lt_spmm_route(const lt_graph *g, int32_t ncols);

/* ---- 2-layer GCN forward (GCN.forward, gcn/models.py:19-24, eval mode) ------------------
 * logits[n,C] = A_hat * (relu(A_hat * (X*W1) + b1) * W2) + b2.   H <= 256, C <= 8.
 * lt_gcn2_workspace_bytes: size of the scratch the call needs (S1, S2, split-K partials). */
size_t lt_gcn2_workspace_bytes(int32_t n, int32_t F, int32_t H, int32_t C);
int lt_gcn2_forward(const lt_graph *g, const float *X, int64_t ldx, int32_t F,
                    const float *W1, const float *b1, int32_t H,
                    const float *W2, const float *b2, int32_t C,
                    float *logits, int64_t ldl, void *workspace, size_t workspace_bytes,
                    void *stream);

/* ---- baseline state for the probe loop --------------------------------------------------
 * The reference recomputes model(features, adj) for every probe (attacker.py:106); it is
 * loop-invariant, so it is computed once here: S1 = X*W1, Z1 = A_hat*S1 + b1,
 * S2 = relu(Z1)*W2, OUT = A_hat*S2 + b2 -- lazily, by the first call that reads them (see lt_baseline_refresh).
 * Owns those four device buffers, the scratch of its hub rows and the side stream / events lt_influence_rows forks
 * hub-row work onto; borrows X, W1, b1, W2, b2 (needed again for the perturbed rows X'[v]*W1). */
int lt_baseline_create(const lt_graph *g, const float *X, int64_t ldx, int32_t F,
                       const float *W1, const float *b1, int32_t H,
                       const float *W2, const float *b2, int32_t C,
                       void *stream, lt_baseline **out);
/* Adds an fp64-accumulated copy of the pre-activation Z1 (one more X*W1 on the f64 matrix cores +
 * one fp64 SpMM; redone after every lt_baseline_refresh when next needed).  LT_MODE_DELTA then evaluates the ReLU kink
 * test on it, which is what brings it within 1e-6 of an fp64 run of the reference; without it the
 * delta mode still works but entries that cross a kink carry ~1e-4 relative error. */
int lt_baseline_enable_fp64(lt_baseline *b, void *stream);
/* The borrowed inputs (X, weights; same pointers) changed, e.g. once per benchmark step.  Launches nothing: everything
 * derived from them is marked stale and recomputed by the first call that reads it, on THAT call's stream (a caller that
 * uses several streams orders them itself) -- S1 = X*W1 by LT_MODE_FULL / LT_MODE_SPARSE rows and lt_baseline_logits;
 * Z1 / S2 / OUT by SPARSE rows and lt_baseline_logits (FULL rows never need them: their stage A yields the unperturbed
 * layer as a by-product and stage B forms the baseline logits of the observed nodes itself); the fp64 pre-activation by
 * LT_MODE_DELTA rows, which then read nothing fp32 of the baseline (the probe's S1 row comes off the fp64 product). */
int lt_baseline_refresh(lt_baseline *b, void *stream);
/* Multi-GPU: the loop-invariant X*W1 sharded over ranks instead of replicated.
 * lt_baseline_attach_s1: the baseline reads S1 = X*W1 from caller-owned storage from now on ([>= n, Hp] fp32 with
 *   Hp = H rounded up to 4, ld == Hp; e.g. the torch tensor the ranks' all-gather writes); the current S1 is copied in.
 * lt_baseline_refresh_rows: like lt_baseline_refresh, but computes only rows [row_begin, row_end) of X*W1, into
 *   dst[row_end - row_begin, Hp] (typically the rank's send buffer of the all-gather).  Rows carry the same bits
 *   whichever rank computed them (the split-K slicing is that of the full product).  The rest of S1 is the caller's
 *   job; Z1 / S2 / OUT are marked stale exactly as by lt_baseline_refresh. */
int lt_baseline_attach_s1(lt_baseline *b, float *S1, int64_t ld, void *stream);
int lt_baseline_refresh_rows(lt_baseline *b, int32_t row_begin, int32_t row_end, float *dst, void *stream);
/* The fp64 twins (LT_MODE_DELTA with lt_baseline_enable_fp64 on): lt_baseline_attach_s1d hands the library caller-owned
 * storage for S1d = X*W1 in fp64 ([>= n, Hp] doubles, ld == Hp, 16-byte aligned: the output of the ranks' all-gather) and
 * switches the baseline to "S1d arrives from outside"; lt_baseline_refresh_rows_fp64 computes rows [row_begin, row_end)
 * of the fp64 product into dst[row_end - row_begin, Hp] (the rank's send buffer).  Same K slicing whatever the range, so a
 * row carries the same bits whichever rank computed it.  lt_baseline_fp64_route: 1 when the baseline's features were
 * found to be sparse differences to a reference row (two-valued columns, as standardised indicator features are) and
 * the fp64 product therefore costs one pass over X -- then sharding it buys nothing -- 0 when it runs on the f64
 * matrix cores, 2 when the pre-activation is formed aggregate-first on the rows a call needs (no S1d at all), -1 when fp64 is
 * not enabled. */
int lt_baseline_attach_s1d(lt_baseline *b, double *S1d, int64_t ld, void *stream);
int lt_baseline_refresh_rows_fp64(lt_baseline *b, int32_t row_begin, int32_t row_end, double *dst, void *stream);
int lt_baseline_fp64_route(const lt_baseline *b, int32_t *route);
/* The on-demand route by ROW LIST (lt_baseline_fp64_route == 2; LT_ERR_UNSUPPORTED otherwise): a call of lt_influence_rows forms
 * the fp64 pre-activation Z1d on the rows ITS probes reach.  On a heavy-tailed graph most of that work lies in hub rows every
 * rank's probes reach (BASELINE configs[4], 8 ranks x 512 probes: ~3 800 rows of >= 1 024 entries hold 84 % of a rank's gathers,
 * and they are the same rows on every rank).  Every rank knows the whole probe list (attacker.py:220: one list), so the ranks
 * can split those rows: each forms its share (lt_baseline_form_rows_fp64: the listed rows now; rows valid since the last refresh
 * are skipped), packs it (lt_baseline_gather_rows_fp64: dst[i, 0 .. Hp) = Z1d[rows[i]], the send buffer of ONE all-gather) and
 * adopts everybody's (lt_baseline_scatter_rows_fp64: Z1d[rows[i]] = src[i, 0 .. Hp), marked valid).  The calls of
 * lt_influence_rows that follow skip valid rows.  A row's bits do not depend on who formed it.  rows: device int32 lists (ids out
 * of range are skipped); dst / src: device, [n_rows, Hp] doubles, 16-byte aligned; Hp = H rounded up to 4. */
int lt_baseline_form_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, void *stream);
int lt_baseline_gather_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, double *dst, void *stream);
int lt_baseline_scatter_rows_fp64(lt_baseline *b, const int32_t *rows, int32_t n_rows, const double *src, void *stream);
int lt_baseline_destroy(lt_baseline *b);
/* copies the baseline logits OUT [n, C] (dense, ld = C) to a device buffer */
int lt_baseline_logits(const lt_baseline *b, float *dst, void *stream);

/* ---- the probe primitive: rows of the influence matrix ----------------------------------
 * Stands in for Attacker.get_gradient_eps_mat + the inner j-loop of
 * link_prediction_attack_efficient (attacker.py:100-108, 220-229):
 *   out[i*ldo + j] = || (f(X + delta * e_v x_v^T) - f(X))[u_j] ||_2 / delta,
 *   v = probe_nodes[i], u_j = observe_nodes[j]
 * probe_nodes / observe_nodes / out are device pointers.  One call handles any n_probe
 * (internally chunked so the workspace stays bounded); no host synchronisation. */
size_t lt_influence_workspace_bytes(const lt_baseline *b, int32_t n_probe, int32_t n_obs,
                                    int32_t mode);
int lt_influence_rows(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                      const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode,
                      float *out, int64_t ldo, void *workspace, size_t workspace_bytes,
                      void *stream);

/* Node ids arrive as DEVICE lists, which the host side of this ABI cannot read: the first kernel of a call that touches a list
 * checks every id against [0, n) -- where the reference raises IndexError at grad_mat[test_nodes[j]] / features[v]
 * (attacker.py:103, 226-229) -- replaces an id out of range by node 0 for the rest of the call (no out-of-bounds access;
 * the rows / columns of such ids are meaningless) and raises a flag in mapped host memory.  The flag is reported
 *   - by lt_node_check: *bad_probe / *bad_observe (may be NULL) = 1 when a list of a call whose kernels have COMPLETED held such
 *     an id (call it after synchronising the stream); clears the flag; returns LT_ERR_INDEX when either is set;
 *   - by the next lt_influence_rows / _vec / lt_influence3_rows* call that finds it set: LT_ERR_INDEX, nothing enqueued. */
int lt_node_check(int32_t *bad_probe, int32_t *bad_observe);

/* ---- the finished rows, once, as float64 ------------------------------------------------------------------------------
 * Stands in for influence_val = np.zeros((n_test, n_test)) (float64, attacker.py:216) and the n_test^2 `.norm().item()` host
 * round trips that fill it (attacker.py:227-229): dst[i * ldd + j] = (double)src[i * lds + j] by one launch.  src: device
 * [rows, lds] fp32 (what lt_influence_rows wrote).  dst: device memory, or PINNED host memory (hipHostMalloc /
 * hipHostRegister, e.g. a torch tensor with pin_memory=True): the kernel then writes over PCIe through the buffer's
 * device-side alias -- no staging copy, no second operation on the stream; the bytes are valid on the host once the stream has
 * drained.  Pageable host pointers are refused (LT_ERR_INVALID). */
int lt_export_rows_f64(const float *src, int64_t lds, int32_t rows, int32_t cols, double *dst, int64_t ldd, void *stream);

/* lt_influence_rows and lt_export_rows_f64 in ONE call: out ([n_probe, ldo] fp32, device) is written as by lt_influence_rows, and
 * dst[i * ldd + j] = (double)out[i * ldo + j] -- the reference's influence_val (attacker.py:216, 227-229) -- in device memory or
 * PINNED host memory (as lt_export_rows_f64).  On the fused LT_MODE_DELTA route (graphs with incidence records) every probe's
 * block writes its own finished row into dst and no export launch follows; when the call also recomputes the baseline (it
 * follows an lt_baseline_refresh) the first half of dst's rows is zero-filled by a few waves riding in the launches that form
 * the fp64 product rows and the pre-activation -- np.zeros crossing PCIe under kernels that do not touch the link -- and the
 * blocks of those rows send their touched positions only (tuning keys "export_sparse", "export_zero_*").  Every other route
 * ends with the launch lt_export_rows_f64 makes.  Same values as the two calls, bit for bit. */
int lt_influence_rows_f64(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                          const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode,
                          float *out, int64_t ldo, double *dst, int64_t ldd, void *workspace, size_t workspace_bytes,
                          void *stream);

/* ---- measurement support: the gather ceiling of the tiled SpMM ------------------------------------------------------
 * The tiled (column-sliced work-item) kernel of lt_spmm_csr_f32 with everything but its gathers removed: the same work
 * items, order, column stream, slice placement and piece size, `in_flight` (8 = the kernel's own, or 16) gathers per lane;
 * no values, no arithmetic, no result rows.  Its duration bounds from below ANY row-gather SpMM that issues this index
 * stream -- what bench.py reports as roofline_spmm.gather_ceiling next to the kernel it bounds.  `sink`: device scratch
 * of lt_spmm_gather_ceiling_bytes(g) bytes (one word per item and slice keeps the loads alive).  `out` != NULL ([n, ldo]
 * fp32): the result rows are stored as the real kernel stores them (their content is meaningless): gathers + result stores. */
size_t lt_spmm_gather_ceiling_bytes(const lt_graph *g);
int lt_spmm_gather_ceiling(const lt_graph *g, const float *S, int64_t lds, int32_t ncols, int32_t in_flight,
                           void *sink, size_t sink_bytes, float *out, int64_t ldo, void *stream);

/* ---- layers wider than one pass of the fused kernels (hidden > 256 or > 8 classes) ------------------------------------
 * The reference has no width limit (gcn/layers.py:14-36, main.py:30 --hidden).  The perturbation's effect on the logits is
 * a SUM over the hidden units, so a wide model is served slice by slice: for every slice s of <= 256 hidden units and every
 * slice t of <= 8 classes a baseline on (W1[:, s], b1[s], W2[s, t], b2[t]) and one
 *   lt_influence_rows_vec   = lt_influence_rows that also writes the pair's difference VECTOR, unscaled:
 *                             vec[(i * ldo + j) * C_t + c] = (f_s(X + d e_v x_v^T) - f_s(X))[u_j, c]   (LT_MODE_SPARSE: the fp32
 *                             finite difference of the slice's own forward; LT_MODE_DELTA: the propagated difference; exact zeros
 *                             outside the probe's 2-hop set in both).  FULL has no vector form (it names SPARSE's bits).
 * and per class slice one
 *   lt_wide_combine         d_c = (vec_0 + vec_1 + ... )[c] / delta in slice order, ss = fma(d_c, d_c, ss) in class order on top of
 *                             the running ss[pair] of the previous class slices (first != 0: start from 0), and on the last class
 *                             slice (last != 0) ss[pair] <- sqrt(ss[pair]): the influence score of attacker.py:227-229.
 * vecs: HOST array of n_vec device pointers, each [n_pairs, C] dense.  Launch count per matrix: ~5 per (s, t) + one per t,
 * independent of n_probe (the round-3 route looped ~5 launches per PROBE). */
int lt_influence_rows_vec(const lt_baseline *b, const int32_t *probe_nodes, int32_t n_probe,
                          const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode,
                          float *out, int64_t ldo, float *vec, void *workspace, size_t workspace_bytes,
                          void *stream);
int lt_wide_combine(const float *const *vecs, int32_t n_vec, int64_t n_pairs, int32_t C, float delta,
                    float *ss, int32_t first, int32_t last, void *stream);

/* ---- the same for the 3-layer model (GCN3, gcn/models.py:28-46; --n-layer 3, gcn_trainer.py:81-86) -------------
 *   logits = A (relu(A (relu(A (X W1) + b1) W2) + b2) W3) + b3,   H1, H2 <= 256, C <= 8.
 * lt_baseline3_create owns the unperturbed forward (S1, H1, S2, Z2, S3, OUT) and borrows X and the six parameter tensors;
 * lt_baseline3_refresh marks everything stale and launches nothing: the next reader recomputes what IT reads on its own stream
 * (a LT_MODE_DELTA build the fp64 pre-activations only, the fp32 modes and lt_baseline3_logits the fp32 forward).  lt_influence3_rows evaluates the
 * reference's fp32 finite difference (f(X + d e_v x_v^T) - f(X))[u] / d only where it can be non-zero: the rows a
 * probe reaches in 1, 2 and 3 hops, each recomputed with the arithmetic of the baseline forward, so unreachable
 * pairs are exactly 0.  The conventions of lt_influence_rows (enqueue only, no host synchronisation: the item
 * count that sizes the level-2 GEMM stays on the device). */
typedef struct lt_baseline3 lt_baseline3;
int lt_baseline3_create(const lt_graph *g, const float *X, int64_t ldx, int32_t F,
                        const float *W1, const float *b1, int32_t H1,
                        const float *W2, const float *b2, int32_t H2,
                        const float *W3, const float *b3, int32_t C,
                        void *stream, lt_baseline3 **out);
int lt_baseline3_refresh(lt_baseline3 *b, void *stream);
int lt_baseline3_destroy(lt_baseline3 *b);
int lt_baseline3_logits(const lt_baseline3 *b, float *dst, void *stream);
size_t lt_influence3_workspace_bytes(const lt_baseline3 *b, int32_t n_probe, int32_t n_obs);
int lt_influence3_rows(const lt_baseline3 *b, const int32_t *probe_nodes, int32_t n_probe,
                       const int32_t *observe_nodes, int32_t n_obs, float delta,
                       float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);
/* The same with a mode: LT_MODE_SPARSE (= LT_MODE_FULL here: lt_influence3_rows, the fp32 finite difference on the 3-hop set)
 * or LT_MODE_DELTA -- the perturbation propagated exactly through the three layers (no subtraction of nearly equal numbers;
 * the two ReLU kink tests read fp64-accumulated pre-activations), which needs lt_baseline3_enable_fp64 first: an fp64 copy of
 * the first two layers' pre-activations (the layer-1 product takes the routes of lt_baseline_enable_fp64), recomputed after
 * every lt_baseline3_refresh when next needed. */
int lt_baseline3_enable_fp64(lt_baseline3 *b, void *stream);
int lt_influence3_rows_mode(const lt_baseline3 *b, const int32_t *probe_nodes, int32_t n_probe,
                            const int32_t *observe_nodes, int32_t n_obs, float delta, int32_t mode,
                            float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);

/* ---- LapGraph cell selection (worker.py:302-335: A += noise, the n_keep largest cells of the strict lower
 * triangle by a 50-way np.argpartition) --------------------------------------------------------------------------------
 * The noise stays numpy's (stream compatibility with --noise-seed): the host draws the N x N float64 matrix and uploads
 * it as `cells` (device, [n, n], only j < i is read; overwritten with adjacency + noise).  lower_rowptr / lower_col:
 * device CSR of the 0/1 adjacency (entries with j >= i are ignored).  The call adds 1.0 on the edges (the reference's fp64
 * add), radix-selects the n_keep-th largest key and writes the flat indices i * n + j of the selected cells to out_idx
 * ([n_keep] int64, device, unordered).  work: >= 4096 bytes of device scratch.  threshold_out (host, may be NULL): the
 * n_keep-th largest value.  Synchronises the stream.  The selected set equals np.argpartition's unless the threshold
 * value is tied; a threshold <= 0 (the selection would reach the zero cells of the upper triangle, where the reference
 * asserts) returns LT_ERR_UNSUPPORTED. */
int lt_lapgraph_select(int32_t n, const int32_t *lower_rowptr, const int32_t *lower_col, double *cells, int64_t n_keep,
                       int64_t *out_idx, void *work, size_t work_bytes, double *threshold_out, void *stream);

/* ---- per-kernel timing (used by bench.py for the roofline object) --------------------------
 * lt_profile_enable(mask): bit k of mask set = launches of kernel class k are bracketed by a pair of
 * hipEvents on the caller's stream (mask 0 = off, -1 = every class; an event pair costs a few
 * microseconds of stream time, so a timed region should enable only what it reports -- and may sample:
 * lt_set_tuning("profile_every", N) brackets only every N-th launch group of an enabled class).
 * lt_profile_summary synchronises on the recorded events and returns the summed duration and
 * launch count of one kernel class.  No reference counterpart (the reference only
 * has wall-clock prints, attacker.py:213,231). */
typedef enum lt_kernel_id {
    LT_K_GEMM = 0,        /* k_gemm_f32_mfma                           */
    LT_K_LAYER1 = 1,      /* k_layer1 (baseline fused SpMM1+ReLU+W2)   */
    LT_K_LAYER2 = 2,      /* k_layer2 (baseline SpMM2 + b2)            */
    LT_K_PERTURB = 3,     /* (unused since the perturbation moved into the probe-row GEMM's loads; id kept) */
    LT_K_FULL_A = 4,      /* k_full_stageA / k_full_stageA_lds: batched perturbed SpMM1+ReLU+W2 */
    LT_K_FULL_B = 5,      /* k_full_stageB: SpMM2 on observed rows + diff + norm    */
    LT_K_ITEM_A = 6,      /* k_item_stageA (sparse / delta)            */
    LT_K_ITEM_B = 7,      /* k_item_stageB (sparse / delta)            */
    LT_K_SPMM = 8,        /* k_spmm_rows / k_spmm_narrow               */
    LT_K_FP64_PRODUCT = 9,/* S1d = X*W1 in fp64: k_s1d_feature_rows (+ the reference row's product), or k_gemm_f64acc_128 + k_sum_slabs_f64 */
    LT_K_FP64_SPMM = 10,  /* Z1d = A_hat*S1d + b1: k_spmm_f64 / k_rows_tiled_f64 (+ long-row combine) */
    LT_K_ITEM_BITS = 11,  /* k_item_bits (+ the pair-mark kernels): item offsets, (probe, row) table, membership bitmap */
    LT_K_COUNT = 12
} lt_kernel_id;
int lt_profile_enable(int mask);
int lt_profile_reset(void);
int lt_profile_summary(int kernel_id, double *total_ms, int64_t *launches);
/* "profile_every" = N samples whole CALLS of lt_influence_rows* (every scope of a sampled call is bracketed, none of the others):
 * the number of calls sampled since lt_profile_enable -- a class's time per call = its total / this, however many scopes it opens */
int lt_profile_calls(int64_t *calls_sampled);

#ifdef __cplusplus
}
#endif
#endif /* LINKTELLER_HIP_H */
