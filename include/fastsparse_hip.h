/*
 * fastsparse_hip.h -- C-ABI of libfastsparse_hip.so (MI355X / gfx950).
 *
 * Two layers, both plain C (no C++/torch types cross this boundary):
 *
 *  (1) the reference's own entry points -- A_mul_B, At_mul_B, csr_A_mul_B, bcsr_*,
 *      bsbm_*, sdm_*, bsdm_*, cbcsr_* and the format constructors -- declared in
 *      include/sparse.h, dsparse.h, csr.h, cbcsr.h with the reference's struct layouts
 *      and signatures.  They accept host OR device pointers for x / y and keep a
 *      device copy of each matrix in a side table keyed by the host struct
 *      (fs_invalidate / fs_release below manage that table).
 *
 *  (2) the device-resident layer declared here: opaque matrix handles living in HBM,
 *      products on device pointers and an explicit HIP stream.  This is what a
 *      device-resident caller (CG loop, multi-GPU driver, bench.py) binds, and what
 *      layer (1) is implemented on.
 *
 * Every function that can fail returns 0 on success and a negative fs_status
 * otherwise; fs_last_error() gives the message for the calling thread.  There is no
 * CPU fallback anywhere: without a usable HIP device the calls fail.
 *
 * Reference interfaces replaced (all in /root/reference):
 *   csr_A_mul_B csr.h:425, csr_A_mul_Bn csr.h:441, bcsr_A_mul_B csr.h:149,
 *   bcsr_A_mul_B2/_B4/_B8/_B8_auto/_Bn/_B32n csr.h:164-302, bcsr_AA_mul_B csr.h:305,
 *   parallel_bcsr_AA_mul_B csr.h:323, A_mul_B sparse.h:58, At_mul_B sparse.h:68,
 *   bsbm_A_mul_B/_B2/_B4/_Bn sparse.h:259-336, sdm_A_mul_B dsparse.h:43,
 *   sdm_At_mul_B dsparse.h:54, bsdm_A_mul_B dsparse.h:176, cbcsr_A_mul_B cbcsr.h:76.
 */
#ifndef FASTSPARSE_HIP_H
#define FASTSPARSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_matrix_s *fs_matrix_t;   /* device-resident sparse matrix (CSR, valued or pattern-only) */
typedef struct fs_cbcsr_s  *fs_cbcsr_t;    /* device-resident column-blocked binary CSR (cbcsr.h:5-14)   */
typedef void *fs_stream_t;                 /* a hipStream_t; NULL = the legacy default stream            */
typedef struct fs_dist_s *fs_dist_t;              /* the GPUs of one node as one context (one process, N devices, RCCL) */
typedef struct fs_dist_matrix_s *fs_dist_matrix_t; /* a CSR row-sharded over the devices of an fs_dist_t              */

enum fs_status {
  FS_OK = 0,
  FS_ERR_HIP = -1,        /* a HIP runtime call failed (message has the HIP error string) */
  FS_ERR_ARG = -2,        /* bad argument (NULL handle, negative size, k < 1, ...)          */
  FS_ERR_NO_DEVICE = -3,  /* no gfx950 device visible                                        */
  FS_ERR_NO_TRANSPOSE = -4, /* transposed product asked for before fs_matrix_build_transpose */
  FS_ERR_RELEASED = -5     /* the operation reads the plain CSR arrays, which fs_matrix_release_csr gave back */
};

enum fs_memspace { FS_HOST = 0, FS_DEVICE = 1 };

/* ---- runtime ---------------------------------------------------------------------- */
const char *fs_version(void);
const char *fs_last_error(void);
int  fs_device_count(void);
int  fs_set_device(int device);
/* option names: "strict_order" (0/1: storage-order sums, bit-identical to the strict-IEEE CPU order for
 * arbitrary x), "spmv_kernel" (0 = auto, 1 streaming, 2 lanes-per-row, 6 tiled, 7 two-pass, 8 LDS-staged
 * tiled), "tiling" (0 never build the L2-tiled copy, 1 build it when the estimates do not rule it out and let the
 * builder's timing of the candidates decide, 2 always; read when a matrix is created), "tile_rows" / "tile_cols"
 * (0 = auto), "binning" and "ldsx" (the same three values for the two-pass copy and the LDS-staged tiled copy), "reproducible" (0/1, default 0: with 1 only kernels whose
 * sums are bit-identical from run to run are used -- the two-pass kernels add a row's terms with LDS atomics in
 * arrival order, so their last bits can differ between runs for non-integer data; read when a matrix is created
 * and at every product).
 * Options are process-wide.
 *
 * Threads and streams: every entry point may be called from several host threads; launches on one handle are serialised by
 * a lock.  A handle keeps scratch vectors for some kernels (rows that cross chunks in the streaming kernel, sums of cut rows in
 * the tiled kernel, the products of the two-pass kernels for k = 1 and for the k-column sweeps k = 2..4, the cell sums of a
 * column-blocked matrix, the column-major copies of X and Y of multi-column products on the LDS-staged copy, k = 2..16), so
 * products on ONE handle must not overlap in time on different streams: order them, or use one handle per stream.  Only the
 * row kernel (option "spmm_kernel" = 1; k >= 5 on matrices that keep the two-pass copy, k > 16 otherwise) and distinct handles
 * are unrestricted.  (fs_spmv_host / fs_spmv_t_host order themselves behind the handle's last device-vector product.)
 * Options are read at the call, on the calling thread, without a lock: set them before other threads create or multiply.
 * "spmm_kernel" (multi-column products: 0 auto, 1 row kernel, 2 k-column two-pass sweep for k = 2..4, 3 one single-vector sweep
 * per column, 4 the v_mfma_f64_16x16x4_f64 experiment), "ata_kernel" (fs_ata_mul: 0 two products, 2 the fused single kernel),
 * "device_build" (format constructors: 0 host loops, 1 on the device from 4 M entries, 2 on the device always).
 * "tile_split": rows longer than this are cut into virtual rows in the tiled copy (0 = 256).
 * "cg_fixed_order" (default 1; FS_CG_FIXED_ORDER): fs_cg / fs_cg2 / fs_dist_cg run their products with fixed-order sums, as under
 * "reproducible", so that a solve is bit-identical from run to run like the reference's loops (cg.h:25-187); 0 = the default kernels.
 * "dist_cg_scheme" (FS_DIST_CG_SCHEME): fs_dist_cg, 0 every device keeps whole vectors, 1 every device keeps its slice (see there). */
int  fs_set_option(const char *name, int value);
int  fs_get_option(const char *name);

/* ---- dense vectors in HBM (thin wrappers over hipMalloc/hipMemcpy for C callers without HIP headers) */
void *fs_device_alloc(int64_t bytes);
void  fs_device_free(void *p);
int   fs_copy_to_device(void *dst_dev, const void *src_host, int64_t bytes);
int   fs_copy_to_host(void *dst_host, const void *src_dev, int64_t bytes);
int   fs_device_synchronize(void);

/* ---- matrices in HBM ---------------------------------------------------------------- */
/* CSR arrays -> handle.  vals == NULL makes a pattern-only (BinaryCSR) matrix.
 * space says where row_ptr/cols/vals live.  With space == FS_DEVICE and borrow != 0 the
 * handle uses the caller's device arrays in place (they must outlive the handle and be
 * 16-byte aligned); otherwise the arrays are copied.  replaces: struct CSR csr.h:358-366,
 * struct BinaryCSR csr.h:15-22 as device-side containers. */
fs_matrix_t fs_csr_create(int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols,
                          const double *vals, int space, int borrow);
/* COO arrays -> handle; entries are stably ordered by row on the device, so every row keeps
 * the caller's entry order (what new_csr csr.h:375-422 / new_bcsr csr.h:30-67 produce and
 * what the serial COO loops sparse.h:58-65 / dsparse.h:43-51 sum in). */
fs_matrix_t fs_coo_create(int nrow, int ncol, int64_t nnz, const int *rows, const int *cols,
                          const double *vals, int space);
void fs_matrix_destroy(fs_matrix_t A);
/* builds and caches the stably column-ordered CSR of A' (needed by *_t products) */
int  fs_matrix_build_transpose(fs_matrix_t A, fs_stream_t stream);
int  fs_matrix_has_transpose(fs_matrix_t A);
/* which kernel fs_spmv (or fs_spmv_t) runs on this matrix under the current options: 1 chunk-streaming,
 * 2 lanes-per-row, 6 L2-tiled, 7 two-pass, 8 LDS-staged tiled; the choice is made by the format builder when the matrix is created */
int  fs_matrix_spmv_kernel(fs_matrix_t A, int transposed);
/* what the format builder measured when it made that choice: ms per product (median of 5 runs on a zero vector) of
 * [0] the chunk-streaming kernel, [1] the L2-tiled kernel, [2] the LDS-staged tiled kernel, [3] the two-pass pair;
 * 0 for a candidate that was not built (ruled out by the estimates, or the matrix is small) */
int  fs_matrix_candidate_ms(fs_matrix_t A, int transposed, float *ms4);
/* where the one-time work of the matrix (transposed != 0: of its A') went, ms of host wall time: [0] arrays into HBM + validation,
 * [1] ordering (COO -> CSR; the transpose), [2] chunk schedule, [3] two-pass copy built, [4] L2-tiled copy built, [5] LDS-staged copy
 * built, [6] candidates timed, [7] losers freed + scratch trimmed.  The reference's one-time step is new_csr / new_bcsr (csr.h:375-422,
 * 30-67); fs_matrix_device_bytes says what the result holds. */
int  fs_matrix_build_ms(fs_matrix_t A, int transposed, float *ms8);
int  fs_matrix_nrow(fs_matrix_t A);
int  fs_matrix_ncol(fs_matrix_t A);
int64_t fs_matrix_nnz(fs_matrix_t A);
/* HBM bytes one product has to move at least (SURVEY.md 8d formula):
 * (12 or 4)*nnz + 4*(nrow+1) + 8*k*nrow + 8*k*ncol */
int64_t fs_matrix_algorithmic_bytes(fs_matrix_t A, int k);
/* copy the device CSR back (any pointer may be NULL); used by tests */
int  fs_matrix_download(fs_matrix_t A, int transposed, int *row_ptr, int *cols, double *vals);

/* ---- products on device pointers ---------------------------------------------------- */
/* y[nrow] = A x[ncol]           (csr_A_mul_B / bcsr_A_mul_B / A_mul_B / sdm_A_mul_B / bsbm / bsdm) */
int fs_spmv(fs_matrix_t A, double *y, const double *x, fs_stream_t stream);
/* y[ncol] = A' x[nrow]          (At_mul_B / sdm_At_mul_B; CSR At_mul_B of BASELINE config 2) */
int fs_spmv_t(fs_matrix_t A, double *y, const double *x, fs_stream_t stream);
/* The same product in nparts parts (1..64), for callers that ship finished rows while the rest still computes -- the
 * all-gather of the y shards of a row-sharded product overlapped with the product (csr.h:429: the row loop that shards).
 * Call with part = 0 .. nparts-1 in order on ONE stream; once the stream has passed part p, rows [rows[p], rows[p+1]) of y
 * are final, where rows[0 .. nparts] comes from fs_spmv_part_rows (fixed for a handle, nparts and the current options:
 * the product is cut where its kernel finishes rows anyway -- panels of pass 2 of the two-pass pair, generations of
 * workgroups of the tiled kernels; a kernel that cannot be cut does everything with part 0 and reports rows = {0, nrow,
 * nrow, ...}).  All parts together are exactly fs_spmv / fs_spmv_t.
 * The cuts of a (handle, nparts, kernel) are computed on first use -- a small synchronous download of the panel tables -- and kept
 * (the eight most recent): call fs_spmv_part_rows once before a timed or captured loop; fs_spmv_part itself then only launches. */
int fs_spmv_part_rows(fs_matrix_t A, int transposed, int nparts, int *rows /* nparts + 1 */);
int fs_spmv_part(fs_matrix_t A, int transposed, double *y, const double *x, int part, int nparts, fs_stream_t stream);
/* the same for k row-major columns (fs_spmm / fs_spmm_t in parts: the block-CG iteration): the one-sweep kernel of k = 2, 4 is cut
 * like the single-vector pair, every other plan does everything with part 0; k = 1 is fs_spmv_part */
int fs_spmm_part_rows(fs_matrix_t A, int transposed, int k, int nparts, int *rows /* nparts + 1 */);
int fs_spmm_part(fs_matrix_t A, int transposed, double *Y, const double *X, int k, int part, int nparts, fs_stream_t stream);
/* dst[dst_off[i] + j] = src[src_off[i] + j] for j < count[i], i < nseg, in one launch; table_dev = int64[3 * nseg] in HBM:
 * dst_off[nseg], src_off[nseg], count[nseg]; max_count = the largest count (sizes the grid).  Unpacks the padded receive
 * buffer of an all-gather of unequal y shards. */
int fs_copy_segments(int nseg, const int64_t *table_dev, int64_t max_count, const double *src, double *dst, fs_stream_t stream);
/* Y[nrow,k] = A X[ncol,k], X and Y row-major  (csr_A_mul_Bn, bcsr_A_mul_B2..._B32n, bsbm_A_mul_B2/_B4/_Bn).
 * Stream-ordered like fs_spmv: a product never builds a copy and never waits for the device.  What a given k can use
 * beyond the copies made at creation -- the k-column two-pass copy (k = 2..4), the measured choice between one sweep per
 * column and the row kernel (LDS-staged copy, k = 3..16) -- is made by fs_matrix_prepare; without it the product runs on
 * what the handle already holds (fs_matrix_spmm_plan tells which kernel that is). */
int fs_spmm(fs_matrix_t A, double *Y, const double *X, int k, fs_stream_t stream);
int fs_spmm_t(fs_matrix_t A, double *Y, const double *X, int k, fs_stream_t stream);
/* One-time work for products with k columns on A (transposed != 0: on A', after fs_matrix_build_transpose): builds the
 * k-column two-pass copy (k = 2, 3, 4 on matrices that keep the two-pass copy: +(4.25 + 8 kw [+ 8]) bytes per padded entry
 * of HBM with kw = 2 for k = 2, 3 and 4 for k = 4, fs_matrix_device_bytes reports it;
 * 30-40 ms at 160 M entries), allocates the column-major scratch and times column sweeps against the row kernel (k = 3..16
 * on matrices that keep the LDS-staged copy).  Synchronous; idempotent; k = 1 and k > 16 need nothing.  The drop-in
 * layer calls it on the first product with a new k, and for every k listed in the environment variable FS_PREPARE_K
 * (e.g. "2,4") when it makes the device copy of a matrix, so that no later product call carries the work. */
int fs_matrix_prepare(fs_matrix_t A, int k, int transposed, fs_stream_t stream);
/* which kernel fs_spmm / fs_spmm_t runs for this k right now: 1 row kernel, 2 k-column two-pass sweep, 3 one
 * single-vector two-pass sweep per column, 4 MFMA experiment, 5 column-major sweeps of the LDS-staged kernel, 6 / 7 strided
 * sweeps of the LDS-staged / L2-tiled kernel */
int fs_matrix_spmm_plan(fs_matrix_t A, int k, int transposed);
/* HBM the handle holds (A and, once built, A'), in bytes: [0] the CSR arrays it owns + chunk schedule, [1] the kept
 * single-vector copy (two-pass incl. its product stream / L2-tiled / LDS-staged), [2] k-column copies and multi-column scratch */
int fs_matrix_device_bytes(fs_matrix_t A, int64_t *bytes3);
/* Give back what the handle no longer needs.  fs_matrix_release_csr: once the format builder has KEPT a re-ordered copy (two-pass,
 * LDS-staged or L2-tiled) the products run on it alone; the plain row_ptr / cols / vals -- owned arrays are freed, borrowed arrays
 * are forgotten so that the caller may free them -- and the chunk schedule are dead weight (config 2: 1.96 of 5.3 GB; a config-5
 * shard: 4.8 of 12.9 GB).  Applies to A and, when built, A'; returns the number of sides released (0: no kept copy, nothing done).
 * What still needs the plain arrays afterwards fails with FS_ERR_RELEASED: strict_order and spmv_kernel 1 / 2 / 3 products, the row
 * kernel of multi-column products (k >= 5 on two-pass matrices, k > 16), fs_matrix_prepare for a new k, fs_matrix_build_transpose,
 * fs_matrix_download, the fused A'A kernel, and option "reproducible" on an LDS-staged copy that is not orderable.
 * fs_matrix_restore_csr hands the same arrays back (same meaning of space / borrow as fs_csr_create; they are NOT validated or
 * compared again) -- "rebuild on demand" is the caller's: the re-ordered copies do not keep the storage order of a row.
 * Option "release_csr" (FS_RELEASE_CSR, default 0) = 1 releases at the end of fs_csr_create / fs_coo_create /
 * fs_matrix_build_transpose; the drop-in layer never releases (its callers keep the host arrays, and call any entry point next).
 * fs_matrix_release_prepared: the k-column copy and scratch fs_matrix_prepare made for k (0: for every k), on A and A'. */
int fs_matrix_release_csr(fs_matrix_t A);
int fs_matrix_restore_csr(fs_matrix_t A, int transposed, const int *row_ptr, const int *cols, const double *vals, int space, int borrow);
int fs_matrix_release_prepared(fs_matrix_t A, int k);
/* y[ncol] = A'A x[ncol]; tmp is caller scratch of nrow doubles in HBM   (bcsr_AA_mul_B, parallel_bcsr_AA_mul_B) */
int fs_ata_mul(fs_matrix_t A, double *y, const double *x, double *tmp, fs_stream_t stream);

/* ---- products on HOST vectors (what a caller of the reference's csr_A_mul_B(y, A, x) holds: csr.h:149) -----
 * Synchronous: y_host is complete on return.  The handle keeps its own staging vectors, stream and events.  With a
 * two-pass copy (the default for large x) x goes up band range by band range and y comes down panel range by panel
 * range while the kernels of the other ranges run (FS_HOST_CHUNKS ranges, default 8; 1 = copy, product, copy); the
 * panel kernels are launched in ranges of workgroups with y coming down behind each (tall matrices) or, when chunks
 * share panels (few long rows), with x going up in the ranges the launch order needs. */
int fs_spmv_host(fs_matrix_t A, double *y_host, const double *x_host);
int fs_spmv_t_host(fs_matrix_t A, double *y_host, const double *x_host);

/* ---- consumers of the path, device resident (cg.h of the reference) -------------------- */
/* (A'A + lambda I) x = b by conjugate gradients; A and the handle of its transpose as the reference passes
 * them (bsbm_cg cg.h:25); x, b: F = ncol(A) doubles in HBM; stops at ||r|| <= tol ||b|| or after F iterations */
int fs_cg(fs_matrix_t A, fs_matrix_t At, double *x, const double *b, double lambda, double tol, int *out_iter,
          fs_stream_t stream);
/* the same for two right-hand sides, X and B row-major F x 2 (bsbm_cg2 cg.h:85) */
int fs_cg2(fs_matrix_t A, fs_matrix_t At, double *X, const double *B, double lambda, double tol, int *out_iter,
           fs_stream_t stream);
/* y += a x on device vectors (the "+ lambda x" of bsbm_AtA, cg.h:17-21) */
int fs_axpy(int n, double a, const double *x, double *y, fs_stream_t stream);

/* ---- column-blocked binary CSR (cbcsr.h) -------------------------------------------- */
fs_cbcsr_t fs_cbcsr_create(int nrow, int ncol, int nblocks, int colblocksize, const int *row_ptr,
                           const int *cols, int space);
void fs_cbcsr_destroy(fs_cbcsr_t A);
/* y[nrow] = A x: per column block the x tile is staged in LDS, cell sums are added block by block */
int fs_cbcsr_spmv(fs_cbcsr_t A, double *y, const double *x, fs_stream_t stream);

/* ---- several GPUs, one process (fs_dist.hip) ------------------------------------------------
 * Rows are cut into one contiguous shard per device with (almost) equal numbers of non-zeros, every device holds the
 * whole x, multiplies its shard with the single-GPU kernels and the y shards are all-gathered over RCCL (xGMI), so that
 * every device ends up with the whole y: the reference's row-parallel `omp parallel for` (csr.h:429) across devices
 * (SURVEY.md 8e).  The exchange runs INSIDE the product: the local product is cut into FS_DIST_PARTS parts (default 4,
 * fs_spmv_part) and the ncclAllGather of the rows a part finished -- one call per part on a padded buffer, the ranks' calls
 * in one group -- runs on a second stream under the later parts; one fs_copy_segments launch unpacks.  z = A' u is the same
 * scheme on row shards of A' (fs_dist_matrix_build_transpose), with u = the y of the last product in place.
 * librccl.so is loaded with dlopen when a context with more than one distinct device is created.
 * A C caller of ANY product entry point of sparse.h / dsparse.h / csr.h / cbcsr.h / cg.h gets this path by setting
 * FASTSPARSE_NGPU=N (optionally FASTSPARSE_DEVICES=0,1,...) in the environment (fs_dropin.hip, "several GPUs").
 * devices == NULL means devices 0 .. ndev-1; ndev < 1 means every visible device.  A device may be listed more than once
 * ("virtual ranks" on one GPU, for testing the sharding on a one-GPU machine): such a context exchanges the parts
 * with device-to-device copies, because RCCL refuses duplicate devices.  Products on one fs_dist_matrix_t are serialised
 * by a lock (its vectors and part buffers are per matrix). */
fs_dist_t fs_dist_create(int ndev, const int *devices);
void fs_dist_destroy(fs_dist_t D);
int  fs_dist_ndev(fs_dist_t D);
int  fs_dist_uses_rccl(fs_dist_t D);
/* 1 once the context exchanges conservatively: ONE whole-shard all-gather behind the finished local product instead of one per
 * part under the later parts.  Chosen with FS_DIST_PARTS=1, and taken for good when a group call of the overlapped mode returns an
 * error on VIRTUAL ranks (that product is finished conservatively).  With RCCL a failed group call aborts the communicators
 * (ncclCommAbort): that product and every later one on the context return the error -- no second collective is attempted on a
 * half-issued group.  The overlapped mode is UNVERIFIED on more than one GPU. */
int  fs_dist_is_conservative(fs_dist_t D);
/* host CSR arrays -> nnz-balanced row shards, one fs_matrix_t per device; vals == NULL: pattern-only */
fs_dist_matrix_t fs_dist_csr_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols,
                                    const double *vals);
/* host COO arrays (vals == NULL: pattern-only) -> the same shards; the entries are bucketed stably by row first (new_csr / new_bcsr,
 * csr.h:375-422, 30-67), so every row adds in the caller's entry order like the serial COO loops (sparse.h:58-65, dsparse.h:43-51):
 * A_mul_B / sdm_A_mul_B / bsbm_* / bsdm_* across the GPUs.  At_mul_B passes (cols, rows). */
fs_dist_matrix_t fs_dist_coo_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *rows, const int *cols, const double *vals);
/* A and its transpose as the caller holds them -- TWO matrices, like bsbm_cg(x, B, Bt, ...) cg.h:25 -- as one handle for fs_dist_cg /
 * fs_dist_cg2 / fs_dist_ata / the *_t products: direct side = A's, transposed side = At's direct side (a row of A' adds in the order
 * the caller's own At stores it).  Shares the shards of both (no copy; they live until the last of the three handles is destroyed);
 * own vectors and solver work space. */
fs_dist_matrix_t fs_dist_matrix_pair(fs_dist_matrix_t A, fs_dist_matrix_t At);
/* The matrix as per-rank shards -- no whole-matrix array anywhere, so the TOTAL may exceed 2^31 - 1 entries (every shard stays
 * below it: int row_ptr, csr.h:358-366); BASELINE config 5 (3.2 G entries, 8 shards) enters this way.  Shard r = the next
 * shard_rows[r] rows of A: a LOCAL row_ptr (shard_rows[r] + 1 ints from 0), GLOBAL column ids, optional values (vals == NULL or
 * vals[r] == NULL for all r: pattern-only).  space = FS_HOST: host arrays; FS_DEVICE: shard r's arrays are on rank r's device
 * (copied; the caller may free them).  The caller chooses the cuts (by non-zeros for power-law matrices). */
fs_dist_matrix_t fs_dist_csr_create_from_shards(fs_dist_t D, int nrow, int ncol, const int *shard_rows /* ndev */,
                                                const int64_t *shard_nnz /* ndev */, const int *const *row_ptr, const int *const *cols,
                                                const double *const *vals, int space);
void fs_dist_matrix_destroy(fs_dist_matrix_t M);
/* row shards of A' (columns of A cut by non-zeros; every row of A' in ascending A-row order) from the SAME host arrays the
 * matrix was created from (the handle keeps no host copy); idempotent */
int  fs_dist_matrix_build_transpose(fs_dist_matrix_t M, const int *row_ptr, const int *cols, const double *vals);
/* the same shards of A' from the device-resident shards of A: per-rank column counts added up and cut on one device, a stable
 * partition of every shard's entries by owner, device-to-device copies, a stable local sort -- no host array of the matrix
 * (what a matrix created from shards needs; works for any).  Same bounds, same entry order as the host build.  Idempotent. */
int  fs_dist_matrix_build_transpose_device(fs_dist_matrix_t M);
int  fs_dist_matrix_has_transpose(fs_dist_matrix_t M);
int  fs_dist_matrix_bounds(fs_dist_matrix_t M, int *bounds /* ndev + 1: row cuts of A */);
int  fs_dist_matrix_bounds_t(fs_dist_matrix_t M, int *bounds /* ndev + 1: row cuts of A' = column cuts of A */);
int64_t fs_dist_matrix_shard_nnz(fs_dist_matrix_t M, int rank);
int64_t fs_dist_matrix_nnz(fs_dist_matrix_t M);
/* the handle of rank `rank`'s shard of A (transposed != 0: of A'), owned by M, on that rank's device: for inspection
 * (fs_matrix_download, fs_matrix_spmv_kernel, fs_matrix_device_bytes) */
fs_matrix_t fs_dist_matrix_shard(fs_dist_matrix_t M, int rank, int transposed);
/* y[nrow] = A x[ncol] / z[ncol] = A' u[nrow]; synchronous.  The vectors may be in HOST memory -- the input goes to every device over
 * its own PCIe link through a pinned staging buffer (chunks, the host copy of the next under the uploads of the last), the output
 * comes back from device 0 -- or in HBM of any device (hipPointerGetAttributes decides, per vector): then nothing touches the host.
 * The ranks on the input's device read it in place, the others receive it device to device; an output in HBM IS the gathered vector
 * of the first rank on its device (the unpack launch writes it).  The caller's earlier work on the legacy default stream of those
 * devices is waited for first.  The same holds for every fs_dist_* function below that takes vectors. */
int  fs_dist_spmv(fs_dist_matrix_t M, double *y, const double *x);
int  fs_dist_spmv_t(fs_dist_matrix_t M, double *z, const double *u);
/* z[ncol] = A'(A x[ncol]) + lambda x: bcsr_AA_mul_B / parallel_bcsr_AA_mul_B (csr.h:305-355; lambda = 0) and bsbm_AtA (cg.h:9-22)
 * across the GPUs; A x stays on the devices */
int  fs_dist_ata(fs_dist_matrix_t M, double *z, const double *x, double lambda);
/* device-resident forms for iterating callers: fill fs_dist_x(M, r) (ncol doubles on rank r's device) on every rank ONCE;
 * fs_dist_spmv_resident leaves y = A x in fs_dist_y(M, r) (nrow doubles, complete on every rank when the call returns),
 * fs_dist_spmv_t_resident leaves z = A' y in fs_dist_z(M, r) (y of the last product is u, in place), fs_dist_swap_xy makes y
 * the next x (square matrices: power iteration) -- nothing crosses PCIe between products */
int  fs_dist_spmv_resident(fs_dist_matrix_t M);
int  fs_dist_spmv_t_resident(fs_dist_matrix_t M);
int  fs_dist_swap_xy(fs_dist_matrix_t M);
/* (A'A + lambda I) x = b by conjugate gradients on the sharded matrix (bsbm_cg, cg.h:25-82, across the GPUs): everything
 * resident, the scalars of the iteration on the devices.  Option "dist_cg_scheme" 0: the vector steps replicated on every
 * device (identical vectors everywhere, the dots need no exchange, both products carry their all-gather); 1: every device keeps
 * its slice of the unknowns (vector work divided by the devices; the partial dots are all-gathered, 8 bytes per rank, and added in
 * rank order; the new search direction is all-gathered).  Every device decides convergence for itself and the host compares
 * the flags of all of them: a disagreement is an error return (FS_ERR_HIP), not a hang.  Products add in a fixed order unless
 * option "cg_fixed_order" is 0.  b_host, x_host: ncol doubles; stops at ||r|| <= tol ||b||.
 * CLOBBERS fs_dist_x / fs_dist_y / fs_dist_z (they are the solver's work vectors): upload x again before the next resident product. */
int  fs_dist_cg(fs_dist_matrix_t M, double *x_host, const double *b_host, double lambda, double tol, int *out_iter);
/* k row-major columns across the GPUs (csr_A_mul_Bn csr.h:441, bcsr_A_mul_Bn csr.h:257, bsbm_A_mul_Bn sparse.h:318): every device
 * multiplies its shard with the k-column kernels of fs_spmm (prepared on the first call with a new k: synchronous) and the Y shards
 * are all-gathered -- inside the product, part by part, where the k-column product finishes its rows that way (the one-sweep plan
 * of k = 2, 4: fs_spmm_part), otherwise in one whole-shard exchange behind it; host matrices Y[nrow, k], X[ncol, k] / Z[ncol, k], U[nrow, k]; k = 1 is fs_dist_spmv */
int  fs_dist_spmm(fs_dist_matrix_t M, double *Y_host, const double *X_host, int k);
int  fs_dist_spmm_t(fs_dist_matrix_t M, double *Z_host, const double *U_host, int k);
/* (A'A + lambda I) X = B with two right-hand sides, row-major ncol x 2: bsbm_cg2 (cg.h:85-187) across the GPUs; the vector steps
 * and the 2x2 algebra replicated on every device, convergence compared across all devices like fs_dist_cg */
int  fs_dist_cg2(fs_dist_matrix_t M, double *X_host, const double *B_host, double lambda, double tol, int *out_iter);
double *fs_dist_x(fs_dist_matrix_t M, int rank);
double *fs_dist_y(fs_dist_matrix_t M, int rank);
double *fs_dist_z(fs_dist_matrix_t M, int rank);

/* ---- format construction on the device ------------------------------------------------ */
/* Stable bucketing of host COO entries, the operation behind new_csr / new_bcsr (csr.h:375-422, 30-67: kind 0, key =
 * row), new_cbcsr (cbcsr.h:16-65: kind 1, key = (col / param) * nrow + row) and new_bsbm / new_bsdm (sparse.h:175-213,
 * dsparse.h:132-173: kind 2, key = row / param).  Host arrays in, host arrays out: offsets[nbuckets + 1], and the
 * payload arrays in bucket order with every bucket keeping the input order (rows_out / vals_out may be NULL).  The
 * constructors of include/csr.h, cbcsr.h, sparse.h, dsparse.h call it when fs_device_build_wanted(nnz) says so. */
int fs_bucket_coo(int kind, int param, int nrow, int ncol, int64_t nbuckets, int64_t nnz, const int *rows, const int *cols,
                  const double *vals, int *offsets, int *rows_out, int *cols_out, double *vals_out);
int fs_device_build_wanted(int64_t nnz);

/* ---- side table of layer (1) ---------------------------------------------------------- */
/* forget the device copy made for a host struct (call after mutating its arrays in place) */
void fs_invalidate(const void *host_struct);
/* drop every cached device copy */
void fs_release_all(void);
/* number of device copies currently in the side table.  The table is bounded (FS_DROPIN_MAX_ENTRIES, default 64:
 * the least recently used idle copy goes first) and a copy whose build runs out of HBM is retried after every idle
 * copy has been dropped.  Environment, read once: FS_STRICT_CACHE=1 hashes every host array in full on every call
 * (default: in full up to 8 MB per matrix, 2048 samples per array beyond), FS_DROPIN_CACHE=0 reuses nothing. */
int fs_cache_entries(void);

/* ---- synthetic inputs for bench/tests (counter-based, identical to oracle/fs_synth.c) -- */
/* exactly per_row entries per row, columns uniform on [0,ncol), vals uniform(-1,1) (vals may be NULL) */
int fs_synth_uniform(int nrow, int ncol, int per_row, uint64_t seed, int64_t row_offset,
                     int *row_ptr_dev, int *cols_dev, double *vals_dev, fs_stream_t stream);
/* power-law row lengths (P(len >= L) ~ scale/L, clipped to [1,max_len]) for rows [row_offset, row_offset+nrow) */
int fs_synth_powerlaw_lengths(int nrow, double scale, int max_len, uint64_t seed, int64_t row_offset,
                              int *len_dev, fs_stream_t stream);
/* fill cols/vals for a given row_ptr (uniform columns) */
int fs_synth_fill(int nrow, int ncol, uint64_t seed, int64_t row_offset, const int *row_ptr_dev,
                  int *cols_dev, double *vals_dev, fs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTSPARSE_HIP_H */
