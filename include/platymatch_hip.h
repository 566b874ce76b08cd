/*
 * platymatch_hip.h — C ABI of libplatymatch_hip.so, the MI355X (gfx950) implementation of
 * PlatyMatch's estimate_transform hot path.
 *
 * The reference (juglab/PlatyMatch 0.0.4) is pure Python and has no FFI; its boundary is a set
 * of module-level functions the napari widget imports by name (_dock_widget.py:15-21).  Each
 * entry point below names the reference function (file:line) whose inner loops it replaces; the
 * Python mirror of those functions (the modules under platymatch_amd/estimate_transform/ and platymatch_amd/utils/)
 * binds these symbols with ctypes.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd or a torch ROCm tensor's data_ptr());
 *     scalars the kernels consume (centroid, PCA axis, mean distance, 4x4 matrices) are also
 *     read from device memory, so a pipeline can be enqueued without host round trips;
 *   - point clouds are float64, 3 x N row-major — the reference's own layout (rows z, y, x;
 *     `transposed=False`), i.e. structure-of-arrays, so every coordinate load is coalesced;
 *   - descriptors are float64 [N][360] row-major per frame, as get_unary returns them;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue
 *     work and return; no entry point allocates, frees or synchronises;
 *   - scratch comes from the caller: ask pm_*_workspace() for the size (bytes), pass a device
 *     buffer at least that large; workspace contents need not be initialised;
 *   - re-entrant: no global mutable state; safe from any host thread, any device, any stream;
 *   - return value: PM_OK or a negative PM_ERR_* code; never throws, aborts or prints.
 *     Argument errors are detected before anything is enqueued.
 */
#ifndef PLATYMATCH_HIP_H
#define PLATYMATCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 2
#define PM_NBINS 360 /* 5 r x 6 theta x 12 phi: the only binning the reference uses (shape_context.py:10) */

#define PM_OK 0
#define PM_ERR_INVALID_ARG (-1)   /* NULL pointer, non-positive size, bad enum */
#define PM_ERR_WORKSPACE (-2)     /* workspace missing or too small */
#define PM_ERR_LAUNCH (-3)        /* HIP reported an error enqueuing work; see pm_last_hip_error() */
#define PM_ERR_UNSUPPORTED (-4)   /* valid in the reference, not implemented on the device path */
#define PM_ERR_NO_MEMORY (-5)     /* pm_device_alloc: the device has no block of that size left */

#define PM_AFFINE 0  /* transform='Affine'  (find_transform.py:4-17)  */
#define PM_SIMILAR 1 /* transform='Similar' (find_transform.py:21-99) — host-side only, see DESIGN.md */

int pm_version(void);
const char *pm_error_string(int code);
int pm_last_hip_error(void); /* hipError_t of the calling thread's most recent PM_ERR_LAUNCH */

/* Device memory for the matrices of this path, straight from the driver (hipMalloc / hipFree on `device`; the calling thread's
 * current device is left as it was).  The reference keeps its eight N x M matrices in NumPy arrays (_dock_widget.py:547-602:
 * eight np.zeros((N, M))); here they are 64 N M bytes of HBM — 160 GB at 50 000 nuclei — and a general-purpose caching
 * allocator that splits such a block for a small request can neither reuse nor return it afterwards.  The Python mirror
 * (platymatch_amd/device_memory.py) takes every block of 4 GiB and more from these entries and keeps the idle ones itself, whole;
 * a C caller needs them anyway: every other entry of this header takes device pointers.  pm_device_free waits for the device's
 * outstanding work.  PM_ERR_NO_MEMORY: no block of that size left (nothing was allocated, *out = NULL). */
int pm_device_alloc(int device, size_t bytes, void **out);
int pm_device_free(int device, void *ptr);
int pm_device_memory(int device, size_t *free_bytes, size_t *total_bytes);

/* Measurement aid (bench.py; nothing of the path calls it): ONE wave that samples the shader clock while other work runs on the
 * device.  Every period_ticks ticks of the constant 100 MHz counter (s_memrealtime) it stores the pair (s_memtime, s_memrealtime)
 * — samples[2k], samples[2k + 1], k < n_samples — and sleeps in between; the clock between two samples is
 * (d s_memtime / d s_memrealtime) x 100 MHz (MI355X_MICROARCH.md: the in-kernel clock test).  Launch it on a stream of its own
 * BEFORE the kernel of interest: it ends by itself after n_samples x period_ticks ticks (bounded here: at most 65 536 samples
 * and 5 s in all).  samples: device memory, 16 n_samples bytes. */
int pm_clock_probe(unsigned long long *samples, int n_samples, unsigned long long period_ticks, void *stream);

/* ---- cloud statistics --------------------------------------------------------------------- */

/* get_centroid (utils/utils.py:48-56): out[3] = mean of each coordinate row, in NumPy's own summation order (np.mean over
 * the rows of the 3 x N array the widget passes: pieces of 8 192 elements, pairwise inside a piece — csrc/pm_pairwise.h), so
 * the centroid has the reference's bits.  pm_centroid_sequential: the order np.mean takes over axis 0 of an N x 3 array
 * (transposed=True): one point after the other. */
size_t pm_centroid_workspace(int n);
int pm_centroid(const double *xyz, int n, double *out3, void *ws, size_t ws_bytes, void *stream);
int pm_centroid_sequential(const double *xyz, int n, double *out3, void *stream);

/* get_mean_distance (utils/utils.py:58-75): out[1] = np.average of [np.linalg.norm(p_i - p_j) for i < j], restated exactly:
 * the norm as BLAS ddot's x86-64 kernels round it (fused multiply-adds), the mean in NumPy's summation order over the
 * lexicographic pair list (pieces of 8 192, pairwise inside a piece).  Bit-identical to the reference (all fixtures).
 * workspace: one float64 per piece (pm_mean_distance_workspace).  n <= 2 000 000. */
size_t pm_mean_distance_workspace(int n);
int pm_mean_distance(const double *xyz, int n, double *out1, void *ws, size_t ws_bytes, void *stream);

/* The same in two steps, for a run sharded over G devices: pm_mean_distance_rows zeroes `partials`
 * (pm_mean_distance_workspace(n) bytes: one float64 per 8 192-element piece of the pair list) and fills the pieces
 * row_offset, row_offset + row_stride, ... (rank g passes g, G); the ranks' buffers are then summed element-wise (an
 * all-reduce: every entry is non-zero on one rank only, so the sum is exact) and pm_mean_distance_finish adds the pieces
 * first to last as pm_mean_distance does — the result is bit-identical to the one-device call, whatever G. */
int pm_mean_distance_rows(const double *xyz, int n, int row_offset, int row_stride, double *partials, size_t partial_bytes,
                          void *stream);
int pm_mean_distance_finish(const double *partials, int n, double *out1, void *stream);

/* First principal axis as sklearn PCA(3).fit(X).components_[0] (shape_context.py:162-165):
 * unit eigenvector of the sample covariance with the largest eigenvalue, sign chosen so that its
 * largest-magnitude entry is positive.  out[3]. */
size_t pm_pca_axis_workspace(int n);
int pm_pca_axis(const double *xyz, int n, double *out3, void *ws, size_t ws_bytes, void *stream);

/* All three principal axes as sklearn PCA(3).fit(X).components_ (the widget's PCA-only alignment,
 * _dock_widget.py:722-731): out[9] row-major, rows ordered by decreasing variance, each row's
 * largest-magnitude entry positive. */
int pm_pca_components(const double *xyz, int n, double *out9, void *stream);

/* ---- rows "next" of SURVEY.md §8f: evaluation and label images --------------------------------- */

/* scipy.spatial.distance.cdist(a.T, b.T) (EvaluateMetrics._calculate_metrics, _dock_widget.py:1030-1051):
 * out[i*ld + j] = sqrt(((a0-b0)^2 + (a1-b1)^2) + (a2-b2)^2), a is 3 x n, b is 3 x m.  An HBM-write-bound
 * kernel (8 bytes per pair): lanes own adjacent column pairs, 16-byte stores. */
int pm_cdist(const double *a, int n, const double *b, int m, double *out, size_t ld, void *stream);

/* Label image -> per-label voxel count and coordinate sums (the widget's centroid loop over np.where,
 * _dock_widget.py:497-521, as one pass): labels is a Z x Y x X int32 volume (0 = background, labels in
 * [1, n_labels)); counts[n_labels] and sums[3][n_labels] (z, y, x index sums) are uint64 accumulators that
 * the call zeroes itself.  Integer sums are exact, so centroid = sum / count reproduces np.mean bit for bit. */
int pm_label_moments(const int32_t *labels, int nz, int ny, int nx, int n_labels, unsigned long long *counts,
                     unsigned long long *sums3, void *stream);

/* ---- shape context ------------------------------------------------------------------------- */

/* get_unary (shape_context.py:144-188) for rows [row0, row0+nrows) of one cloud of n points:
 * local frame per point (:169-175, 180-181), neighbours expressed in it (transform, :61-84),
 * log-spherical histogram (get_shape_context :10-42, get_bin_index :46-58).
 *   n_frames  2 (type='moving': sc, sc2) or 4 (type='fixed': sc, sc2, sc3, sc4)
 *   counts    [n_frames][nrows][360] int32 — index.count(i) (:39-40); may be NULL
 *   totals    [n_frames][nrows]      int32 — sc.sum() before normalising (:41); may be NULL
 *   hist      [n_frames][nrows][360] float64 = counts / total (:41); may be NULL
 * A row with nothing counted (point == centroid, axis parallel to z) is NaN in `hist`, as the
 * reference's 0/0 is.  Row-block form: ranks of a multi-GPU run pass disjoint [row0, nrows). */
int pm_shape_context(const double *xyz, int n, int row0, int nrows, const double *centroid3,
                     const double *x0_3, const double *mean_dist1, int n_frames, int32_t *counts,
                     int32_t *totals, double *hist, void *stream);

/* The same histograms by the tile kernel (round 3; what the Python mirror calls): a workgroup owns 16 consecutive rows, a
 * lane keeps one neighbour in registers and walks the tile's queries (frames by scalar loads), neighbours are
 * pre-classified in float32 and accepted only when clear of every bin boundary by more than float32 can be off (about
 * 3 999 of 4 000; pm_binning.h: pm_bin_fast32), the others are decided by the float64 expressions of pm_shape_context; one
 * histogram per row is kept (frame 1) and frames 2..4 are written as its phi permutations.  Tiles holding a neighbour for
 * which that permutation does not hold exactly (on a sector edge or pole of the frame) are recomputed by pm_shape_context's
 * kernel inside the same call.  Outputs are identical to pm_shape_context's in every case.
 *   edge_guard2  (device, may be NULL) two uint32 counters of (point, neighbour) pairs whose bin the reference itself would not
 *                decide reproducibly: [1] azimuth within 1e-12 / sin(angle(axis, z)) of a sector edge — the PCA axis agrees
 *                with sklearn's to 1e-12 —, [0] distance within 4e-14 (relative) of a ring radius (informative since
 *                pm_mean_distance returns the reference's bits: a caller's own mean distance may not); and, both counters,
 *                any neighbour within 1.6e-13 x (|x| + |y| + |z| of the queried point) of a ring radius [0], a sector plane
 *                or a polar cone [1] — four times the error np.linalg.inv leaves on the reference's local coordinates
 *                (shape_context.py:81; lattice-like input puts neighbours exactly there).  Zero for generic data: "the
 *                reference's histograms" then holds by construction for this call, not only by the fixtures.
 * workspace: pm_shape_context_workspace(nrows) bytes, 256-byte aligned (frames, thresholds, per-tile flags, counts). */
size_t pm_shape_context_workspace(int nrows);
int pm_shape_context_tiled(const double *xyz, int n, int row0, int nrows, const double *centroid3,
                           const double *x0_3, const double *mean_dist1, int n_frames, int32_t *counts,
                           int32_t *totals, double *hist, uint32_t *edge_guard2, void *workspace, size_t workspace_bytes,
                           void *stream);

/* get_shape_context (shape_context.py:10-42) on an explicit neighbour list already expressed in the
 * local frame: nb is n x 3 row-major (x_, y_, z_ per row, as the reference passes it); counts[360]
 * int32 and/or hist[360] float64 (= counts / counts.sum()); total[1] int32 may be NULL. */
int pm_shape_context_neighbors(const double *nb, int n, double mean_dist, int32_t *counts, int32_t *total,
                               double *hist, void *stream);

/* get_shape_context called with its OWN binning arguments (shape_context.py:10: r_inner, r_outer, n_rbins, n_thetabins, n_phibins —
 * get_unary never passes any, the stand-alone function accepts them): nb as above; r_edges[n_rbins] = np.logspace(log10(r_inner),
 * log10(r_outer), n_rbins) (:24; compared as the reference's loop does, :53-56); cos_steps[n_cos_steps]: cos_steps[k] = the largest
 * float64 c with arccos(c) // (pi/n_thetabins) >= k + 1 under the HOST's libm (decreasing; theta_index = #{k : z_/r_ <= cos_steps[k]});
 * phi_steps[n_phi_steps]: phi_steps[m] = the smallest float64 phi in [0, 2 pi] with phi // (2 pi/n_phibins) >= m + 1 (increasing).  The host
 * builds both tables with the reference's own NumPy calls (platymatch_amd/estimate_transform/binning.py).  counts[n_rbins * n_thetabins *
 * n_phibins] int32 and total[1] are zeroed and filled here; a neighbour whose wrapped atan2 lies within 2^-46 of a phi step is NOT
 * counted — the device library's atan2 may differ from the host's in the last bits — but listed: unsure[<= n] receives its row
 * index, n_unsure[1] their number; the caller bins those rows with the reference's own expressions and adds them.  All pointers
 * are device memory.  PM_ERR_UNSUPPORTED beyond 2^24 bins or 4 096 steps / edges. */
int pm_shape_context_neighbors_binned(const double *nb, int n, double mean_dist, const double *r_edges, int n_rbins,
                                      const double *cos_steps, int n_cos_steps, const double *phi_steps, int n_phi_steps,
                                      int n_thetabins, int n_phibins, int32_t *counts, int32_t *total, int32_t *unsure,
                                      int32_t *n_unsure, void *stream);

/* ---- chi-square cost ----------------------------------------------------------------------- */
/* Sizes: every build of this section covers its matrix with tiles of 16 x 64 entries, one 256-thread workgroup each.  A launch
 * holds fewer than 2^32 work-items, i.e. ~16.7 M tiles (~130 000 x 130 000 entries); larger matrices are written by several
 * launches over bands of tile rows, enqueued back to back on `stream` — same bits, nothing for the caller to do (140 000 and
 * 200 000 nuclei run this way).  PM_ERR_INVALID_ARG only if ONE tile row is wider than a launch (nF > 1.07e9). */

/* get_unary_distance (shape_context.py:88-99) for every pair: out[i*ld + j] =
 * 0.5 * sum_k (a_ik - b_jk)^2 / (a_ik + b_jk), bins with a == b skipped, summed k = 0..359
 * in order in float64 — bit-identical to the reference's scalar loop. */
int pm_chi2_cost(const double *scA, int nA, const double *scB, int nB, double *out, size_t ld,
                 void *stream);

/* The widget's eight N x M loops (_dock_widget.py:547-602) in one launch, for a row block of
 * the moving cloud: out + h*matrix_stride is U_h[nM][ld] for h = 0..7 in the widget's order
 * 11,12,13,14,21,22,23,24 (U_ab = moving frame a vs fixed frame b).
 *   sc_m1, sc_m2          [nM][360]  this rank's rows of get_unary(moving) frames 1, 2
 *   sc_f1 .. sc_f4        [nF][360]  all of get_unary(fixed) frames 1..4
 * Every matrix is bit-identical to pm_chi2_cost on the same two descriptor sets. */
int pm_chi2_cost8(const double *sc_m1, const double *sc_m2, int nM, const double *sc_f1,
                  const double *sc_f2, const double *sc_f3, const double *sc_f4, int nF,
                  double *out, size_t ld, size_t matrix_stride, void *stream);

/* Frames 2..4 of get_unary are, away from exact sector edges, phi-sector permutations of frame 1
 * (sc2[q] = sc1[(q+6)%12], sc3[q] = sc1[11-q], sc4[q] = sc1[(5-q) mod 12] within each (r, theta) shell),
 * which lets the eight matrices be built from the frame-1 descriptors with half the divisions.
 * pm_chi2_symmetry_check verifies the relation bit for bit on the given arrays: flag1[0] = 0 if it
 * holds for every row of both clouds, 1 otherwise (the flag is reset by the call itself).
 * pm_chi2_cost8_sym then produces exactly what pm_chi2_cost8 would (same bits, same layout) and must
 * only be called when the flag is 0; otherwise use pm_chi2_cost8. */
int pm_chi2_symmetry_check(const double *sc_m1, const double *sc_m2, int nM, const double *sc_f1,
                           const double *sc_f2, const double *sc_f3, const double *sc_f4, int nF,
                           int32_t *flag1, void *stream);
int pm_chi2_cost8_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                      size_t matrix_stride, void *stream);

/* One hypothesis and its twin only (pairing t = 0..3: U11/U22, U12/U21, U13/U24, U14/U23 — the same terms summed in two
 * orders): out2 + 0*matrix_stride = the natural-order matrix (U11, U12, U13, U14), out2 + 1*matrix_stride = the rolled-order
 * one (U22, U21, U24, U23).  Same bits as the corresponding two matrices of pm_chi2_cost8_sym at a quarter of its work; for
 * clouds whose eight matrices do not fit in HBM together (64 N M bytes) but two do.  Same precondition (symmetry flag 0). */
int pm_chi2_cost_pair_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out2, size_t ld,
                          size_t matrix_stride, void *stream);

/* What estimate_transform's default cost_mode='auto' starts from between 1 024 and 8 191 nuclei (round 4: opt-in; round 5: the
 * default route, always behind the exact matrix's certificate): the same eight matrices in RELAXED float64 arithmetic — no bit identity with
 * the reference's scalar loop (shape_context.py:88-99), every entry within pm_chi2_relaxed_delta() (absolute) of it.
 * U = 0.5 (sum a + sum b) - 2 sum_k a_k b_k / (a_k + b_k): four running sums per row instead of eight (the twins U11/U22,
 * U12/U21, U13/U24, U14/U23 coincide once the order of summation is free and are written twice), v_rcp_f64 + one Newton step
 * instead of a correctly rounded division.  An assignment solved on these matrices counts only once it is proven to be the
 * exact matrices': by pm_chi2_entries_sym below (the Python mirror: lsap.certify_listed), or by a uniqueness margin above
 * 2 min(N, M) delta on the relaxed matrix itself (lsap.certify(min_eps=...)).  variant 0: every term computed;
 * 1 / 2: sparsely filled shells from a 94 x 94 / 64 x 64 table of relaxed terms. */
size_t pm_chi2_relaxed_workspace_bytes(int nM, int nF);
double pm_chi2_relaxed_delta(void);
int pm_chi2_cost8_relaxed(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                          size_t matrix_stride, void *ws, size_t ws_bytes, int variant, void *stream);
/* The FILTER build (what cost_mode='auto' starts from at 8 192 nuclei and above; sharded runs: a rank's row block of it) — the
 * four pairings' matrices (out4 + t * matrix_stride, t = 0..3: U11/U22, U12/U21, U13/U24,
 * U14/U23, each pair one matrix) in packed float32 arithmetic, written as float64; every entry within pm_chi2_filter_delta()
 * (absolute, 1.1e-6) of the exact cost.  3.0x faster than the exact eight-matrix launch, a quarter of its output as float32.  (A
 * 94 x 94 float32 term table for the sparsely filled shells, as the exact kernel has, was built and measured in round 5: no gain —
 * one LDS gather per term costs what the packed arithmetic does; profiles/r05_filter_table.txt.)  Not the reference's
 * values and never handed out as such: the matrices only tell the assignment solver WHICH entries can matter, every cost it uses
 * is evaluated exactly by pm_chi2_entries_sym (the Python mirror: lsap.FilteredMatrix, estimate_transform(cost_mode='filter')).
 * ws: pm_chi2_filter_workspace_bytes of 16-byte aligned device memory.  Same precondition as pm_chi2_cost8_sym (symmetry flag 0). */
size_t pm_chi2_filter_workspace_bytes(int nM, int nF);
double pm_chi2_filter_delta(void);
int pm_chi2_filter4(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out4, size_t ld, size_t matrix_stride,
                    void *ws, size_t ws_bytes, void *stream);
/* One pairing's filter matrix alone (out1 [nM][ld]; a quarter of the launch, the same values): for clouds whose four filter
 * matrices do not fit in HBM together. */
int pm_chi2_filter_pair(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out1, size_t ld,
                        void *ws, size_t ws_bytes, void *stream);
/* The same filter matrices STORED as float32 (within pm_chi2_filter_delta() of the exact cost all the same): half the memory — four
 * matrices of 100 000 x 100 000 resident, one of 200 000 x 200 000 — and half the traffic of the solver's dense passes, which
 * have float32 forms for them: pm_lsap_row_select_f32, pm_lsap_col_min_f32, pm_lsap_certificate_f32 (same contracts as their
 * float64 namesakes; reduced costs are formed in float64 from the converted entry). */
int pm_chi2_filter4_f32(const double *sc_m1, int nM, const double *sc_f1, int nF, float *out4, size_t ld, size_t matrix_stride,
                        void *ws, size_t ws_bytes, void *stream);
int pm_chi2_filter_pair_f32(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, float *out1, size_t ld,
                            void *ws, size_t ws_bytes, void *stream);
int pm_lsap_row_select_f32(const float *U, int nr, int nc, size_t ld, const double *v, int k, int32_t *out_col, double *out_cost,
                           int32_t *nonfinite1, void *stream);
int pm_lsap_col_min_f32(const float *U, int nr, int nc, size_t ld, double *v, void *ws, size_t ws_bytes, void *stream);
int pm_lsap_certificate_f32(const float *U, int nr, int nc, size_t ld, const double *u, const double *v, const int32_t *col4row,
                            double delta, double eps, int32_t *summary4, double *stats2, int32_t *tight, double *tight_red, int cap,
                            double *row_slack, double *row_neg, void *stream);

/* Listed entries of one pairing's two EXACT matrices (the bits of pm_chi2_cost_pair_sym's): out_natural[e], out_rolled[e] =
 * entry (rows[e], cols[e]) of the natural-order matrix (U11, U12, U13, U14 for pairing 0..3) and of its rolled-order twin (U22,
 * U21, U24, U23).  rows / cols / outputs: device, n_entries each; an index out of range yields NaN in both outputs.  What turns
 * an assignment solved on the relaxed matrices into a certificate for the exact ones without building them: with duals (u, v)
 * and assignment s optimal for the relaxed matrix R, |R - C| <= delta entrywise, the row duals retuned to u'_i = C[i][s(i)] -
 * v[s(i)] keep every entry whose relaxed reduced cost exceeds 2 delta feasible for the exact matrix C; only the matched
 * entries and the few below that threshold (+ the uniqueness margin) need their exact values (the Python mirror:
 * lsap.certify_listed).  Same precondition as the pair build (symmetry flag 0). */
int pm_chi2_entries_sym(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, const int32_t *rows,
                        const int32_t *cols, int n_entries, double *out_natural, double *out_rolled, void *stream);

/* The same two calls with a caller-provided workspace (pm_chi2_sym_workspace_bytes, 16-byte aligned device memory), which lets
 * the kernel take most terms of the sparsely filled inner shells from a table: a descriptor value is count / total
 * (get_shape_context's sc / sc.sum(), shape_context.py:40-43), so (a-b)^2/(a+b) is a function of two small integers.  The
 * calls recover the counts from the doubles and verify bit for bit that every value is fl(count / total); the table is filled
 * by the same operations on the same operands, so the result has the bits of pm_chi2_cost8_sym / pm_chi2_cost_pair_sym for ANY
 * input (descriptors that are not count / total, or shells with counts of 88 and more, are computed term by term). */
size_t pm_chi2_sym_workspace_bytes(int nM, int nF);
int pm_chi2_cost8_sym_ws(const double *sc_m1, int nM, const double *sc_f1, int nF, double *out, size_t ld,
                         size_t matrix_stride, void *ws, size_t ws_bytes, void *stream);
int pm_chi2_cost_pair_sym_ws(const double *sc_m1, int nM, const double *sc_f1, int nF, int pairing, double *out2, size_t ld,
                             size_t matrix_stride, void *ws, size_t ws_bytes, void *stream);
/* Report (benchmarks, tests): tabled30[g] = 1 if the last *_ws launch on this workspace took shell g = 6*ring + theta sector from
 * the table, table_size = its side.  Copies 300 bytes to the host and synchronises the stream. */
int pm_chi2_sym_table_info(const void *ws, int32_t *tabled30, int32_t *table_size, void *stream);

/* np.argmin(U_h, axis=1) for n_mat stacked cost matrices (U_h = U + h*matrix_stride, rows x cols, leading
 * dimension ld): idx[h*rows + i] = first index of the minimum of row i, or of the first NaN if the row holds one
 * (NumPy's rule); val (may be NULL) receives the minimum itself.  This is the per-row check BASELINE.json's
 * 200k-point configuration is compared on, and the greedy correspondence for clouds too large for the
 * assignment solver: rows are sharded across ranks like the matrices, so no exchange is needed. */
int pm_row_argmin(const double *U, int n_mat, int rows, int cols, size_t ld, size_t matrix_stride, int32_t *idx,
                  double *val, void *stream);

/* ---- assignment (HOST function: host pointers, no stream) ---------------------------------------- */

/* scipy.optimize.linear_sum_assignment(cost) as the widget calls it (_dock_widget.py:604-611): cost is an
 * nr x nc row-major float64 HOST array; rows/cols receive min(nr, nc) int64 indices (rows ascending), exactly the
 * indices SciPy 1.15.3 returns (its shortest-augmenting-path solver restated, ties included).  Unlike SciPy it can
 * be called from several host threads at once, so the eight hypotheses are solved concurrently.
 * PM_ERR_INVALID_ARG: NaN or -inf entry; PM_ERR_UNSUPPORTED: infeasible matrix. */
int pm_lsap_solve(const double *cost, long nr, long nc, int64_t *rows, int64_t *cols);

/* ---- assignment with the matrix resident on the device ---------------------------------------------------------------
 * The same call sites (_dock_widget.py:604-611) without moving the N x M matrix to the host.  An optimal assignment lives
 * on each row's cheapest entries: the GPU hands the host a sparse CORE of the matrix (pm_lsap_row_select), the host solves
 * the core exactly (pm_lsap_core_*: shortest augmenting paths over k edges per row instead of M), the GPU prices the duals
 * against the whole matrix and returns the offenders until none is left, and pm_lsap_certificate proves the result on
 * every entry: dual feasibility + complementary slackness = optimal for the dense matrix (LP duality); no alternating
 * cycle among the entries within eps of tight = unique with margin eps, hence the assignment SciPy returns.  Matrices that
 * fail the certificate (ties: duplicate nuclei, symmetric clouds) are solved by pm_lsap_solve, SciPy's algorithm itself.
 * platymatch_amd/lsap.py (solve_on_device) is the driver. */

/* DEVICE: per row of U (nr x nc, leading dimension ld) up to k <= 256 entries with small cost - v[col] (v may be NULL = 0):
 * the minima of 256 interleaved column classes, ranked by (reduced cost, column); the row's overall minimum is always
 * rank 0.  out_col [nr][k] (-1 = no entry), out_cost [nr][k] = the entries' raw costs.  nonfinite1[0] = 1 if U holds a NaN or
 * an infinity (such matrices go to pm_lsap_solve, which reports them as SciPy does). */
int pm_lsap_row_select(const double *U, int nr, int nc, size_t ld, const double *v, int k, int32_t *out_col,
                       double *out_cost, int32_t *nonfinite1, void *stream);

/* DEVICE: bids of the listed rows (rows[n_rows], each in [0, nr) — the CALLER checks the range) against prices v: per row the
 * column j1 with the smallest U[i][j] - v[j] (lowest column on ties), that value u1, and the row's second smallest value u2.
 * The Jacobi form of the augmenting row reduction that warms up a solve on the dense rows: lsap._row_reduction. */
int pm_lsap_bid(const double *U, int nr, int nc, size_t ld, const double *v, const int32_t *rows, int n_rows,
                int32_t *out_j1, double *out_u1, double *out_u2, void *stream);

/* DEVICE: column minima v[j] = min_i U[i][j] — the column reduction a square solve starts its duals from. */
size_t pm_lsap_col_min_workspace(int nr, int nc);
int pm_lsap_col_min(const double *U, int nr, int nc, size_t ld, double *v, void *ws, size_t ws_bytes, void *stream);

/* DEVICE: certificate of (u[nr], v[nc], col4row[nr]) against every entry of U.  summary4 = { entries with reduced cost
 * (U[i][j] - v[j]) - u[i] < -delta; non-matching entries with reduced cost <= eps, appended to tight[cap][2] as (row, col)
 * with their reduced costs in tight_red[cap] — if the count exceeds cap the list is incomplete; matched entries with
 * |reduced cost| > delta; 0 }.  stats2 = { largest |reduced cost| on a matched entry, largest negative reduced cost };
 * row_slack[nr], row_neg[nr] (may be NULL) = the same two quantities per row: any other assignment costs at least
 * (reduced costs of its new entries) - sum(row_slack) - sum(row_neg) more than this one, so the caller knows how small an
 * eps still separates the optimum from every alternative, and filters the list by it. */
int pm_lsap_certificate(const double *U, int nr, int nc, size_t ld, const double *u, const double *v,
                        const int32_t *col4row, double delta, double eps, int32_t *summary4, double *stats2,
                        int32_t *tight, double *tight_red, int cap, double *row_slack, double *row_neg, void *stream);
/* dst[j*ld_dst + i] = src[i*ld_src + j] for a rows x cols matrix (dst: cols x rows): the solver takes the short side of a cost
 * matrix as its rows (as SciPy does); with more moving than fixed nuclei that is the transpose of what the cost kernels write. */
int pm_transpose_f64(const double *src, int rows, int cols, size_t ld_src, double *dst, size_t ld_dst, void *stream);

/* HOST: the sparse core solver, one instance per matrix (nr <= nc; nc - nr implicit zero-cost dummy rows square the
 * problem).  add: k candidate edges per real row (cols [nr][k], -1 skipped, duplicates skipped).  solve: augment every
 * free row.  reprice: cand_* are pm_lsap_row_select's output for the solver's current v; rows whose dense minimum reduced
 * cost is below -delta get the offending entries as new edges, a repaired dual, and are freed for the next solve;
 * *n_violated = number of such rows (0 = the core optimum is dual feasible on the dense matrix).  get: duals and
 * assignment of the real rows; stats4 = { edges, Dijkstra steps, augmentations, dummy-row scans }. */
void *pm_lsap_core_create(int nr, int nc);
/* Start from given duals instead of zeros (call before any solve): v[nc] any values with, for every core edge added so far
 * or later, cost - v[col] >= u[row]; u[nr] = each row's minimum of cost - v[col] over the DENSE row (pm_lsap_row_select's
 * rank-0 entry).  Rows whose minimising column is still free are matched to it at once (a tight edge).  nr == nc only. */
int pm_lsap_core_init_duals(void *core, const double *u, const double *v, const int32_t *argmin_col);
/* Start from a given dual-feasible state with a partial matching (call before any solve, after the edges were added):
 * u[nr], v[nc], col4row[nr] (-1 = free); every matched pair must be a core edge with cost - v[col] = u[row] (tight). */
int pm_lsap_core_init_state(void *core, const double *u, const double *v, const int32_t *col4row);
void pm_lsap_core_destroy(void *core);
int pm_lsap_core_add(void *core, int k, const int32_t *cols, const double *costs);
/* Optional warm start of the first solve (after the edges and, for square problems, pm_lsap_core_init_duals): a forward auction
 * with eps-scaling from eps0 down to eps_min (divided by factor per round) over the core's edges.  Leaves feasible duals
 * close to the optimum's and the tight part of the auction's assignment; pm_lsap_core_solve completes it exactly as it
 * would any other feasible start.  May be called repeatedly (e.g. after pm_lsap_core_reprice added edges) as long as
 * pm_lsap_core_solve has not run.  max_bids > 0 bounds the work of this call; bids (may be NULL): bids placed so far. */
int pm_lsap_core_auction(void *core, double eps0, double eps_min, double factor, long max_bids, long *bids);
/* The same auction continued from a state produced elsewhere (any synchronous — Jacobi — rounds over the same core; the
 * library ships none: a device form was simulated with this entry, tests/probes/auction_sim.py, and not built, DESIGN.md
 * §4.5c): price[nc] (= -v) and assigned[nr] (-1 = unassigned; otherwise a column of that row's core edges,
 * no column twice — PM_ERR_INVALID_ARG if not).  The first phase (eps0) does not reset the assignment: only the unassigned rows
 * bid — the narrow, sequential tail of the imported phase; further phases down to eps_min (if eps0 > eps_min) run as in
 * pm_lsap_core_auction.  Ends with the same tightening; pm_lsap_core_solve completes the rest. */
int pm_lsap_core_auction_resume(void *core, const double *price, const int32_t *assigned, double eps0, double eps_min,
                                double factor, long max_bids, long *bids);
int pm_lsap_core_solve(void *core);
int pm_lsap_core_reprice(void *core, int k, const int32_t *cand_col, const double *cand_cost, double delta, int *n_violated);
int pm_lsap_core_get(void *core, double *u, double *v, int32_t *col4row, long *stats4);
/* How many pricing rounds pm_lsap_core_reprice settled from the COLUMN side (round 5): when the matching is complete and many
 * violated rows meet on few columns — the signature of one huge alternating tree lifted by a solve's last augmentations — the
 * offending columns' duals drop by their worst violation and only their holders are freed, instead of every violated row
 * (a 50 000 x 50 000 filtered solve: 48 179 rows freed before, 1-2 thousand now).  PM_LSAP_COLUMN_REPAIR=0 in the environment
 * switches it off, =force takes it whenever the matching is complete (tests).  -1 for a NULL core. */
long pm_lsap_core_column_repairs(void *core);

/* DEVICE: out[r] = U[r][col0 + r] for r < min(nr, nc - col0): the diagonal entries of a row block that starts at matrix row col0
 * (the safety edges of the core: with them it always holds a perfect matching). */
int pm_lsap_diagonal(const double *U, int nr, int nc, size_t ld, int col0, double *out, void *stream);

/* HOST + DEVICE (round 4): the whole scheme above for ONE resident matrix in one call — column reduction, core selection,
 * eps-scaling auctions with pricing in between, shortest augmenting paths, pricing rounds until no entry of U violates dual
 * feasibility (pm_lsap_solve_resident: platymatch_amd/lsap.py's solve_core), and the certificate with its uniqueness check
 * (pm_lsap_certify_resident: lsap.certify) — as native host code around the kernels and the core solver declared here: one
 * foreign call per hypothesis instead of ~60, no interpreter lock shared by the hypotheses' threads.  U: device, nr <= nc (the
 * caller transposes otherwise); u [nr], v [nc], col4row [nr], tight_out: HOST; dev_ws: pm_lsap_resident_workspace bytes of device
 * memory, 256-byte aligned; blocks until the result is on the host (it synchronises `stream` several times); staging goes
 * through a pinned host buffer owned by the calling thread (grown on demand, freed with the thread).
 * report.status: 0 solved (duals feasible on every entry of U), 2 pricing did not converge, 3 non-finite entries, 4 infeasible.
 * certify: report.optimal / report.unique as lsap.certify's info; tight_out (may be NULL) receives the (row, col) pairs of the
 * non-matching entries within eps of tight (report.n_tight of them; -1 if more than tight_capacity). */
typedef struct pm_lsap_options {
    int core_edges, price_edges, max_pricing_rounds, column_reduction;
    double rel_delta, rel_eps_collect, rel_eps_floor, eps_safety;
    int auction, a_rounds, a_bids_per_row, a_later_bids_per_row;
    double a_eps0, a_eps_min, a_factor, a_later_eps0, a_stop_below, a_max_free_columns;
    double min_eps;
} pm_lsap_options;
typedef struct pm_lsap_report {
    int status, rounds, violations, loose, tight_within_eps, n_tight, optimal, unique, n_auction_violated, pad_;
    long bids, steps, augmentations, edges, dummy_scans;
    double slack_bound, delta, eps, seconds_total, seconds_auction, seconds_core, seconds_device, seconds_certify;
    int auction_violated[8];
    int violated_per_round[32];
} pm_lsap_report;
void pm_lsap_default_options(pm_lsap_options *options);
size_t pm_lsap_resident_workspace(int nr, int nc);
int pm_lsap_solve_resident(const double *U, int nr, int nc, size_t ld, const pm_lsap_options *options, double *u, double *v,
                           int32_t *col4row, pm_lsap_report *report, void *dev_ws, size_t dev_ws_bytes, void *stream);
int pm_lsap_certify_resident(const double *U, int nr, int nc, size_t ld, const pm_lsap_options *options, const double *u,
                             const double *v, const int32_t *col4row, int32_t *tight_out, int tight_capacity,
                             pm_lsap_report *report, void *dev_ws, size_t dev_ws_bytes, void *stream);

/* HOST: 1 if the entries listed by pm_lsap_certificate (tight [n_tight][2]) admit no alternating cycle — and, for
 * nr < nc, no alternating path between a free column and a column whose dual is within eps of v_free_level (the dual the
 * free columns carry) — i.e. the certified optimum is unique with margin eps; 0 if an alternative exists. */
int pm_lsap_unique(int nr, int nc, const int32_t *col4row, const double *v, double v_free_level, double eps,
                   const int32_t *tight, int n_tight);

/* The index sets do_ransac draws (HOST function): `trials` successive np.random.choice(n, k, replace=False) calls on
 * NumPy's legacy global generator (shape_context.py:122), reproduced from its MT19937 state — key[624] and *pos of
 * np.random.get_state(), advanced in place exactly as NumPy would advance them.  out: trials x k int32. */
int pm_legacy_choice(uint32_t *key, int *pos, long n, int k, long trials, int32_t *out);

/* ---- RANSAC -------------------------------------------------------------------------------- */

/* do_ransac's trial loop (shape_context.py:121-138) with the index sets drawn by the caller
 * (the reference draws them with np.random.choice; the host mirror does the same, in the same
 * order, so a seeded run sees identical sets).  Matched clouds are given indirectly:
 * point k of the matched pair list is (mov[:, rows[k]], fix[:, cols[k]]) — the widget's
 * moving[:, row_indices], fixed[:, col_indices] (_dock_widget.py:622-675); rows/cols may be NULL
 * for the identity.  Per trial t: fit the affine through its `min_samples` (>= 4) pairs
 * (get_affine_transform, find_transform.py:4-17: the interpolating affine for four pairs, least squares for more),
 * apply it to all n pairs and count ||fixed - predicted|| <= error.
 *   samples     [trials][min_samples] int32 indices into the matched list
 *   A_out       [trials][16] float64 row-major 4x4 (last row 0 0 0 1)
 *   inliers     [trials] int32
 *   degenerate  [trials] int32 (may be NULL): 1 where the sample is (nearly) rank deficient — coplanar or repeated
 *               points.  The reference's pinv returns a minimum-norm answer there; such trials get A = NaN and 0 inliers
 *               here and the caller refits them with pinv on the host and scores them with pm_ransac_score.
 * min_samples < 4 (rank deficient by construction) returns PM_ERR_UNSUPPORTED: host pinv + pm_ransac_score. */
int pm_ransac_affine(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows,
                     const int32_t *cols, int n, const int32_t *samples, int min_samples, int trials, double error,
                     double *A_out, int32_t *inliers, int32_t *degenerate, void *stream);

/* Index sets drawn ON THE DEVICE, for runs nobody seeded.  do_ransac draws a trial's pairs with
 * np.random.choice(n, min_samples, replace=False) from NumPy's global generator (shape_context.py:122), which the reference
 * never seeds: the contract is "min_samples distinct pairs, every subset equally likely".  Trial t draws from its own
 * counter-based stream — Philox-4x32-10 with key = seed and counter = (t, block, run, 0); `run` separates the eight
 * do_ransac calls of one registration — by Floyd's subset algorithm with exactly uniform bounded integers (Lemire).  The
 * same (seed, run, n, min_samples) always yields the same sets, on any device.
 *   pm_ransac_draw         samples[trials][min_samples] int32 only (transform='Similar' fits on the host; tests)
 *   pm_ransac_affine_draw  pm_ransac_affine with the draw fused in front of each trial's fit; the sets are also written to
 *                          samples_out (the caller refits flagged trials with pinv and may want the winner's pairs)
 * Seeded calls keep the host replica of NumPy's stream (pm_legacy_choice) and pm_ransac_affine. */
int pm_ransac_draw(int n, int min_samples, int trials, uint64_t seed, uint32_t run, int32_t *samples, void *stream);
int pm_ransac_affine_draw(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows,
                          const int32_t *cols, int n, int min_samples, int trials, uint64_t seed, uint32_t run,
                          double error, int32_t *samples_out, double *A_out, int32_t *inliers, int32_t *degenerate,
                          void *stream);

/* Score caller-supplied transforms instead of fitting (transform='Similar', whose 4x4 eigen-decomposition stays on
 * the host, and the pinv refits of degenerate affine samples): A_in [trials][16]; only rows 0-2 are applied, as
 * apply_affine_transform does (apply_transform.py:14-17). */
int pm_ransac_score(const double *mov, int n_mov, const double *fix, int n_fix, const int32_t *rows,
                    const int32_t *cols, int n, const double *A_in, int trials, double error,
                    int32_t *inliers, void *stream);

/* ---- transforms ---------------------------------------------------------------------------- */

/* apply_affine_transform (apply_transform.py:3-17): out(3 x n) = (A . [in; 1])[:3].  A is 16
 * float64 row-major on the device.  in == out is allowed. */
int pm_apply_affine(const double *A16, const double *in, int n, double *out, void *stream);

/* get_affine_transform (find_transform.py:4-17) for full-rank input: least-squares 4x4 with
 * [fixed;1] ~ A [moving;1], solved as centred normal equations (3x3 SPD solve + translation).
 * If nn != NULL the pairing is (mov[:, i], fix[:, nn[i]]), i < n (ICP, perform_icp.py:18);
 * else (mov[:, i], fix[:, i]).  A_out[16].  status1[0] (may be NULL) = 1 if the moving points are (nearly) coplanar /
 * fewer than four — the normal equations are singular where pinv is not, the caller must then use pinv — else 0. */
size_t pm_fit_affine_workspace(int n);
int pm_fit_affine(const double *mov, int n, const double *fix, int n_fix, const int32_t *nn,
                  double *A_out16, int32_t *status1, void *ws, size_t ws_bytes, void *stream);

/* ---- ICP ----------------------------------------------------------------------------------- */

/* perform_icp's correspondence step (perform_icp.py:15-16): nn[i] = argmin_j
 * sqrt(sum((fix[:,j]-mov[:,i])**2)), first index on ties (scipy distance_matrix + np.argmin);
 * dist[i] (may be NULL) = that distance.  The N x M matrix is never materialised.
 * pm_icp_nn bins the fixed cloud into a uniform grid and searches only the cells around each moving point
 * (exactly the brute-force answer, O(N) per call instead of O(N*M)); pm_icp_grid_build / pm_icp_grid_nn expose
 * the two halves so that a loop over a constant fixed cloud (ICP) bins it once; pm_icp_nn_brute evaluates every
 * pair (one wave-uniform fixed point against 128 moving points per wave) and is kept as the O(N*M) yardstick. */
size_t pm_icp_nn_workspace(int n, int m);
int pm_icp_nn(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist,
              void *ws, size_t ws_bytes, void *stream);
size_t pm_icp_grid_workspace(int m);
int pm_icp_grid_build(const double *fix, int m, void *grid, size_t grid_bytes, void *stream);
int pm_icp_grid_nn(const double *mov, int n, int m, const void *grid, size_t grid_bytes, int32_t *nn,
                   double *dist, void *stream);
size_t pm_icp_nn_brute_workspace(int n, int m);
int pm_icp_nn_brute(const double *mov, int n, const double *fix, int m, int32_t *nn, double *dist,
                    void *ws, size_t ws_bytes, void *stream);

/* Partial sums of one rank's block for the affine refit: sums[PM_ICP_NSUMS] =
 * { n, sum m(3), sum f(3), sum m m^T (6: 00 01 02 11 12 22), sum f m^T (9 row-major), sum |f|^2, 0 } with
 * m = mov[:, i] - origin_m, f = fix[:, nn[i]] - origin_f (origin6 = 3+3 doubles on the device:
 * any fixed shift, e.g. the first points; improves conditioning, cancels in the solve).
 * Multi-GPU ICP: every rank accumulates its rows, the 24 doubles are all-gathered and added in
 * rank order, then pm_icp_update runs identically on every rank. */
#define PM_ICP_NSUMS 24
size_t pm_icp_accumulate_workspace(int n);
int pm_icp_accumulate(const double *mov, int n, const double *fix, int m, const int32_t *nn,
                      const double *origin6, double *sums, void *ws, size_t ws_bytes, void *stream);

/* From (globally summed) sums: A_est = least-squares affine (perform_icp.py:18), mov <- A_est.mov
 * in place (:23), A_icp <- A_est . A_icp (:25), residual1[0] += nothing; residual_parts receives
 * this block's sum of ||mov_new - fix[nn]|| and count so the caller can form get_error (:24,
 * utils.py:77-88) across ranks: residual_parts[2] = { sum, n }.  status1 (may be NULL): set to 1 — never cleared, the
 * caller zeroes it before a loop — if the moment matrix is (nearly) singular, i.e. the moving cloud is planar. */
size_t pm_icp_update_workspace(int n);
int pm_icp_update(const double *sums, const double *origin6, double *mov, int n, const double *fix,
                  int m, const int32_t *nn, double *A_icp16, double *A_est16, double *residual_parts2,
                  int32_t *status1, void *ws, size_t ws_bytes, void *stream);

/* Same as pm_icp_update with the 4x4 given by the caller instead of solved from sums (a step fitted elsewhere:
 * apply, residual and composition A_icp = A_est . A_icp of perform_icp.py:23-25 on the device). */
int pm_icp_apply(const double *A_est16, double *mov, int n, const double *fix, int m, const int32_t *nn,
                 double *A_icp16, double *residual_parts2, void *ws, size_t ws_bytes, void *stream);

/* get_error (utils/utils.py:77-88): out[1] = mean over columns of ||a[:, i] - b[:, i]||, a and b 3 x n. */
size_t pm_get_error_workspace(int n);
int pm_get_error(const double *a, const double *b, int n, double *out1, void *ws, size_t ws_bytes, void *stream);

/* perform_icp (perform_icp.py:7-26), transform='Affine', whole loop on one device:
 *   mov        3 x n, updated in place to the final moved cloud
 *   A_icp      16 doubles out (starts from identity)
 *   residuals  [iters] mean ||moving - fixed[:, nn]|| after each update (the value the reference prints); may be NULL
 *   nn_all     [iters][n] int32 NN indices of every iteration; may be NULL
 *   status1    [1] int32, may be NULL: 0, or 1 if some iteration's moving cloud was (nearly) planar — the result is then
 *              meaningless (the reference's pinv handles that case) and the caller must rerun with pinv fits
 *              (pm_icp_grid_nn + host fit + pm_icp_apply per iteration)
 * iters == 0 returns the identity. */
size_t pm_icp_workspace(int n, int m);
int pm_icp(double *mov, int n, const double *fix, int m, int iters, double *A_icp16,
           double *residuals, int32_t *nn_all, int32_t *status1, void *ws, size_t ws_bytes, void *stream);

/* The same loop — same arguments, workspace and results, bit for bit — with iterations 1 .. iters-1 in ONE launch of
 * persistent workgroups (round 3): a workgroup keeps its points, their matches and the current 4 x 4 in registers / LDS,
 * the search tables stay warm in L2, and an iteration ends with the reduction tree of pm_icp plus the publication of the
 * fitted transform through a generation word (no kernel boundary, no reload of the cloud, no gather of the previous matches).
 * CONTRACT: at most ONE pm_icp_one_launch may be in flight per device — two half-resident persistent grids can starve each
 * other.  The library enforces it (round 4): a per-device flag is taken by the call and handed back by a host function its
 * stream runs when the call's work has finished; a second call on the same device meanwhile returns PM_ERR_UNSUPPORTED and
 * enqueues nothing (call pm_icp instead, or wait).  (The only process-global state in the library besides the occupancy cache;
 * the Python mirror additionally holds a per-device lock until the stream has drained, so it never sees the refusal.)  Ordinary kernels on
 * other streams may run beside it, but while they hold CUs that workgroups of this grid are waiting for, the resident ones
 * spin: meant for a device that is otherwise idle (one registration at a time) — a batch of registrations on many streams
 * should call pm_icp.  If the grid does not fit the device at once with one workgroup per CU to spare (about 49 000 points at four
 * workgroups per CU on 256 CUs), or iters < 3, the call runs
 * pm_icp's launch-per-iteration path.  A workgroup that has waited 2 s for a round gives up and reports status1 = 2: results
 * are then undefined and the caller reruns with pm_icp. */
int pm_icp_one_launch(double *mov, int n, const double *fix, int m, int iters, double *A_icp16,
                      double *residuals, int32_t *nn_all, int32_t *status1, void *ws, size_t ws_bytes,
                      void *stream);

/* ---- transform='Similar' ------------------------------------------------------------------------ */

/* The O(N) part of get_similar_transform (find_transform.py:21-99) in NumPy's own arithmetic, so that the 4 x 4 quaternion
 * matrix the host hands to np.linalg.eig is the reference's, bit for bit (its result hangs on the last bit: the fit takes
 * ROW 0 of the eigenvector matrix, :60-66).  Pairs are (mov[:, i], fix[:, nn[i]]), or (mov[:, i], fix[:, i]) if nn is NULL.
 *   out17 = { com_source[3] (:28), com_target[3] (:27), Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz (:43-53), D, Sp (:86-91) }
 *   mov_sequential / fix_sequential: how np.mean adds that cloud up — 0: a C-ordered 3 x N array (rows contiguous: NumPy's
 *   chunked pairwise sum, csrc/pm_pairwise.h), 1: a Fortran-ordered one (what fancy indexing makes of fixed[:, nn]: the
 *   columns are added one after the other).  The nine sums are always pairwise (np.sum of fresh product vectors); D and Sp
 *   are serial, their 3-vector dot products fused as BLAS ddot's x86 kernels fuse them.
 * workspace: pm_similar_workspace(n) bytes, 256-byte aligned.  Everything is enqueued on `stream`; nothing is awaited. */
size_t pm_similar_workspace(int n);
int pm_similar_moments(const double *mov, int n, const double *fix, int m, const int32_t *nn, int mov_sequential,
                       int fix_sequential, double *out17, void *ws, size_t ws_bytes, void *stream);

/* mov <- (A . [mov; 1])[:3] in place as np.matmul computes it (apply_transform.py:14-17: BLAS dgemm, fused multiply-adds over
 * k = 0 .. 3 starting from zero); if residual1 != NULL also np.mean(np.linalg.norm(mov - fix[:, nn], axis=0)) (get_error,
 * utils.py:77-88) of the moved cloud.  A16: sixteen float64 on the device, row-major.  ws is needed only for the residual. */
int pm_similar_apply(const double *A16, double *mov, int n, const double *fix, int m, const int32_t *nn,
                     double *residual1, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PLATYMATCH_HIP_H */
