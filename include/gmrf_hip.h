/*
 * gmrf_hip.h -- C ABI of libgmrf_hip.so: MI355X (gfx950) block-tridiagonal Cholesky
 * factor / solve / sample path for GMRF posteriors.
 *
 * Drop-in boundary for ONE path of timweiland/DiffEqGMRFs.jl (all citations relative to
 * /root/reference):
 *     src/tridiagonal_cholesky.jl   tridiagonal_cholesky :65-82, forward_solve :43-52,
 *                                   backward_solve :24-33, ldiv!/ldiv :54-63,
 *                                   TridiagonalCholeskyFactor :5-9
 *     scripts/solve_burger.jl       extract_blocks :182-254 (block input form)
 *     SpMV call sites               scripts/solve_burger.jl:157-158,166,177 (Q * x)
 * The reference has no FFI of its own (pure Julia); these are the entry points a
 * `ccall` shim binds (julia/DiffEqGMRFsHIP.jl).
 *
 * Conventions
 *   - every function returns a gmrf_status (0 = OK, negative = error class);
 *   - no C++ types, no exceptions, no torch types cross this boundary;
 *   - data pointers (`double*`, `float*`) may be HOST or DEVICE pointers; the library asks
 *     the HIP runtime which (hipPointerGetAttributes) and stages host data itself;
 *   - matrices of right-hand sides are column-major n x k with leading dimension ld >= n
 *     (Julia Matrix{Float64} layout);
 *   - one handle = one GPU + one HIP stream; a handle is used by one host thread at a time;
 *   - calls are synchronous at return unless the name ends in _async.
 */
#ifndef GMRF_HIP_H
#define GMRF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t gmrf_status;

enum {
    GMRF_OK = 0,
    GMRF_ERR_NOT_SPD = -1,     /* Julia PosDefException: *info = failing block (1-based)   */
    GMRF_ERR_BAD_SHAPE = -2,   /* n % N_blocks != 0, k <= 0, ld < n, null pointer ...       */
    GMRF_ERR_BAND = -3,        /* an entry lies outside the block tri-band of the partition */
    GMRF_ERR_HIP = -4,         /* a HIP runtime call failed (gmrf_last_error has the text)  */
    GMRF_ERR_NO_FACTOR = -5,   /* solve/sample before a successful factor                    */
    GMRF_ERR_NO_DEVICE = -6,   /* no gfx950 device visible                                   */
    GMRF_ERR_ALLOC = -7,
    GMRF_ERR_RCCL = -8         /* librccl missing or a collective failed (gmrf_last_error)  */
};

enum { GMRF_SOLVE_FULL = 0, GMRF_SOLVE_FORWARD = 1, GMRF_SOLVE_BACKWARD = 2 };
enum { GMRF_VAR_EXACT = 0, GMRF_VAR_RBMC = 1, GMRF_VAR_MC = 2 };
enum { GMRF_BLOCK_L = 0, GMRF_BLOCK_C = 1, GMRF_BLOCK_LINV = 2 };

typedef struct gmrf_handle gmrf_handle;   /* block-tridiagonal factor context  */
typedef struct gmrf_csr gmrf_csr;         /* device-resident CSR matrix (K6)   */
typedef struct gmrf_comm gmrf_comm;       /* RCCL communicator of this process */

/* One sparse block in compressed-sparse-row or -column form, as extract_blocks
 * (scripts/solve_burger.jl:240-247) returns them: `ptr` has dim+1 entries. */
typedef struct {
    int64_t nnz;
    const int64_t* ptr;
    const int64_t* idx;
    const double* val;
} gmrf_sparse_block;

/* Per-phase statistics of the last calls on a handle (gmrf_bt_stats). */
typedef struct {
    double factor_ms;          /* last gmrf_bt_factor_*                                     */
    double solve_ms;           /* last gmrf_bt_solve (device part)                          */
    double sample_ms;          /* last gmrf_bt_sample                                       */
    double factor_flops;       /* N bs^3/3 + (N-1) 2 bs^3 (LAPACK counts, logical bs)       */
    double sweep_bytes;        /* 8 [N bs(bs+1)/2 + (N-1) bs^2] + 16 n k of the last sweep (the reference's dense blocks) */
    double sweep_ms;           /* duration of the last single sweep                         */
    int64_t n, n_blocks, block_size, block_size_padded;
    int64_t factor_bytes;      /* device bytes held by L (if kept), C (stored window), Linv */
    /* per-kernel accounting, filled when profiling is on (gmrf_bt_set_profiling); one class per
     * kernel symbol so that a class compares with one row of a rocprofv3 kernel trace:
     *  0 / 11 / 12 gemm_f64_mfma<false,false> / <false,true> / <true,*> (64 x 64 tile GEMM: G2,
     *    rank-256 updates, GEMM sweeps / doubling assembly, GEMM sweeps / selected inversion)
     *  1 potrf_step<false> (tile Cholesky + inverse; fused with panel + update for one problem)
     *  2 sweep_mm (k >= 2 right-hand sides)     3 sweep_gemv_n / _t (k = 1)
     *  4 csr_spmm                                5 other (scatter, pack, Philox, ...)
     *  6 gemm_f64_big<false>                     7 gemm_f64_big<true>   (128 x 128 tile GEMM)
     *  8 potrf_panel                             9 potrf_update
     * 10 spmm_bxt (sparse C = B X^T)          13 gemm_f64_ll (32 x 32 tile GEMM of small launches)
     * 14 / 15 gemm_f64_dma<.., B [n][k]> / <.., B [k][n]> (LDS-DMA staged GEMM: the batches' products since round 3)
     * 16 potrf_diag128 (128 x 128 diagonal block of a batch: two tile Choleskys + the block's inverse)
     * 17 potrf_persist (one problem: the in-block Cholesky of a block / a 256-column panel in ONE persistent launch; small
     *    batches: the 256 x 256 diagonal block of a panel.  Round 4 booked these under class 1; class 17 was potrf_panel256, removed)
     * 18 gemm_f64_dma<.., A [k][m]> (the A^T B products of selected inversion: round 4)
     * 19 / 20 sweep_persist<k = 1> / <k >= 16> (one problem: a whole sweep in ONE persistent launch; round 5)
     * work = algorithmic flops (0-2, 6-9, 11-18, 20) or algorithmic bytes (3-5, 10, 19). */
#define GMRF_KERNEL_CLASSES 24
    double kernel_ms[GMRF_KERNEL_CLASSES];
    double kernel_work[GMRF_KERNEL_CLASSES];
    int64_t kernel_launches[GMRF_KERNEL_CLASSES];
    double sweep_bytes_streamed;  /* bytes the last sweep really read: Linv triangles + C inside the staircase + 16 n k */
    /* persistent launches (potrf_persist: one launch per block / per 256-column panel / per panel diagonal block; round 4),
     * made observable in round 5 (failure detection): */
    int32_t persist_route;        /* form the LAST factorisation launched: 0 none (launch per step / potrf_diag128 + GEMM), 1 one launch per
                                   * block (one problem, blocks of up to 16 tiles), 2 one per 256-column panel (one problem, larger
                                   * blocks), 3 one per 256 x 256 diagonal block of a panel (small batches) */
    int32_t persist_aborts;       /* times a bounded wait inside a persistent launch of this handle gave up: the range was repeated
                                   * with the launch-per-step form, which the handle keeps until it is destroyed */
    int32_t persist_cus;          /* CUs this handle holds of its device's budget for persistent launches (every workgroup of such a
                                   * launch must be resident; the claims of all handles of a device never exceed its CU count) */
    int32_t persist_refused;      /* 1: the budget refused this handle's claim (other handles hold the CUs): no persistent launches */
    int32_t sweep_persist;        /* 1: the last gmrf_bt_solve / gmrf_bt_sample ran its sweeps as ONE persistent launch each (one problem,
                                   * blocks of 512 .. 1024, the handle holds the whole chip); a launch that gives up counts in persist_aborts */
    int32_t sweep_persist_launches;   /* such launches since the handle was created */
} gmrf_stats;

/* ------------------------------------------------------------------ life cycle */

/* device >= 0: HIP device ordinal.  stream: a hipStream_t to run on (e.g. the caller's
 * PyTorch stream) or NULL to let the handle create its own. */
gmrf_status gmrf_bt_create(int32_t device, void* stream, gmrf_handle** out);
gmrf_status gmrf_bt_destroy(gmrf_handle* h);
const char* gmrf_last_error(void);
int32_t gmrf_version(void);

/* ------------------------------------------------------------------ factor
 * tridiagonal_cholesky(A::SparseMatrixCSC, N_blocks)  (src/tridiagonal_cholesky.jl:65-82).
 * colptr/rowval/nzval are the SparseMatrixCSC fields (index_base = 1 from Julia, 0 from
 * SciPy).  Only entries in the lower blocks (i,i) and (i,i-1) are used, as in the
 * reference (:73,:76); the (i,i+1) blocks are ignored; anything further out is
 * GMRF_ERR_BAND.  The factor stays resident on the device inside `h`. */
gmrf_status gmrf_bt_factor_csc(gmrf_handle* h, int64_t n, int64_t n_blocks,
                               const int64_t* colptr, const int64_t* rowval,
                               const double* nzval, int32_t index_base, int32_t* info);

/* Same from the output of extract_blocks (scripts/solve_burger.jl:182-254): n_blocks
 * diagonal blocks and n_blocks-1 lower off-diagonal blocks, each block_size square, CSC
 * (compressed by column, like SparseMatrixCSC) when `compressed_by_column` != 0. */
gmrf_status gmrf_bt_factor_blocks(gmrf_handle* h, int64_t n, int64_t n_blocks,
                                  const gmrf_sparse_block* diag,
                                  const gmrf_sparse_block* lower,
                                  int32_t index_base, int32_t compressed_by_column,
                                  int32_t* info);

/* Re-run the factorisation with new values on the SAME sparsity pattern as the last
 * gmrf_bt_factor_csc (Gauss-Newton loop, scripts/solve_burger.jl:143-149). */
gmrf_status gmrf_bt_refactor_values(gmrf_handle* h, const double* nzval, int32_t* info);

/* ------------------------------------------------------------------ solves
 * mode FULL:     y = A^-1 b           ldiv!/ldiv      (:54-63)
 * mode FORWARD:  y = L^-1 b           forward_solve   (:43-52)
 * mode BACKWARD: y = L^-T b           backward_solve  (:24-33)
 * b, y: n x k column-major with leading dimensions ldb, ldy (a strided Julia view and a dense
 * result may differ); b == y (in place, as ldiv! allows) needs ldb == ldy. */
gmrf_status gmrf_bt_solve(gmrf_handle* h, const double* b, double* y, int64_t k,
                          int64_t ldb, int64_t ldy, int32_t mode);

/* k samples  x_s = mean + L^-T z_s  (rand(rng, x_cond), solve_darcy_gmrf-fem.jl:191).
 * z == NULL: z_s[dof] is Philox4x32-10(key = seed, counter = (dof, first_id + s)) through
 * Box-Muller, independent of GPU count and launch geometry.  z != NULL: n x k column-major
 * standard normals supplied by the caller (parity mode).  mean may be NULL (zero). */
gmrf_status gmrf_bt_sample(gmrf_handle* h, uint64_t seed, int64_t first_id, int64_t k,
                           const double* mean, const double* z, double* out, int64_t ld);

/* mean = A^-1 b and k samples mean + L^-T z(id = first_id + s) in ONE call: what
 * scripts/darcy/solve_darcy_gmrf-fem.jl:190-191 asks of one factor (`mean`, then `rand`).  Results bitwise those of
 * gmrf_bt_solve(mode 0) followed by gmrf_bt_sample(mean = that mean, z = NULL); where the sweeps of a handle are persistent
 * launches (one problem, blocks of 512 .. 1024, device pointers, 2 <= k <= 128) the samples' sweep runs BESIDE the mean's two on a
 * second stream.  b, mean: n doubles; samples: n x k column-major, leading dimension ld.  stats.solve_ms = the whole call. */
gmrf_status gmrf_bt_posterior(gmrf_handle* h, const double* b, uint64_t seed, int64_t first_id, int64_t k,
                              double* mean, double* samples, int64_t ld);

/* The N(0,1) draws gmrf_bt_sample would use (for tests and for callers that need z). */
gmrf_status gmrf_bt_normals(gmrf_handle* h, uint64_t seed, int64_t first_id, int64_t k,
                            double* z, int64_t ld);

/* Marginal variances diag(A^-1)  (std(x_cond), solve_darcy_gmrf-fem.jl:192).
 * EXACT: block-tridiagonal selected inversion (deterministic); serves a batch too, var_out is then
 *        [batch][n].
 * RBMC:  Rao-Blackwellised Monte Carlo over k Philox samples, needs Q (the factored matrix).
 * MC:    plain Monte Carlo over k samples.
 * var_out (here and in the batch call below): host or DEVICE memory -- a device pointer keeps the variances on the device (no
 * copy out; the call still returns after the handle's stream has finished). */
gmrf_status gmrf_bt_marginal_var(gmrf_handle* h, int32_t method, int64_t k, uint64_t seed,
                                 const gmrf_csr* Q, double* var_out);

/* RBMC / MC variances of every problem of a batch in one call (var_out is [batch][n]).  Q gives
 * the sparsity pattern only; q_vals[batch][nnz] are the problems' values in Q's CSR order (for
 * a symmetric matrix: the nzval arrays the factor was given).  Problem p draws the sample ids
 * p*k .. p*k+k-1. */
gmrf_status gmrf_bt_marginal_var_batch(gmrf_handle* h, int32_t method, int64_t k, uint64_t seed,
                                       const gmrf_csr* Q, const double* q_vals, double* var_out);

/* Accumulators for sharded variance estimation: adds this rank's contribution of samples
 * [first_id, first_id + k) to acc (length n; RBMC: sum of squared off-diagonal terms,
 * MC: sum of squares).  The caller all-reduces acc and finishes with
 * var = 1/Q_ii + acc / K_total (RBMC) or acc / K_total (MC). */
gmrf_status gmrf_bt_var_accumulate(gmrf_handle* h, int32_t method, int64_t first_id,
                                   int64_t k, uint64_t seed, const gmrf_csr* Q,
                                   double* acc);

/* logdet(A) = 2 sum_i sum_j log (L_i)_jj. */
gmrf_status gmrf_bt_logdet(gmrf_handle* h, double* out);

/* ------------------------------------------------------------------ factor access
 * Copy one dense block of the factor to `out` (block_size x block_size, column-major,
 * leading dimension ld):  kind L -> chos[i].L, kind C -> Cs[i] (= L_{i+2,i+1} in 1-based
 * block numbering; i in [0, n_blocks-1)), kind LINV -> inv(chos[i].L).  i is 0-based. */
gmrf_status gmrf_bt_get_block(gmrf_handle* h, int32_t kind, int64_t i, double* out,
                              int64_t ld);

/* Flat host image of the selected problem's factor (SURVEY 8b `gmrf_bt_export_factor` /
 * `_import_factor`; what a Julia caller materialises `F.chos` / `F.Cs` from in one call, and the
 * unit a checkpoint or a host-side broadcast moves).  Layout: int64 header[8] = {0x46524d47,
 * version 1, n, N, bs, 1, 0, 0}, then column-major bs x bs blocks L_1..L_N, C_1..C_{N-1},
 * Linv_1..Linv_N.  Import re-creates the device factor (no numeric work). */
gmrf_status gmrf_bt_export_size(gmrf_handle* h, int64_t* bytes);
gmrf_status gmrf_bt_export_factor(gmrf_handle* h, void* host_buf, int64_t bytes);
gmrf_status gmrf_bt_import_factor(gmrf_handle* h, const void* host_buf, int64_t bytes);

/* Storage of the factor on the device (all fp64, row-major blocks padded to bsp = 64 * 2^p):
 *   LINV  [batch][N][bsp][bsp]      Linv_i = inv(chos[i].L), lower triangular -- what the sweeps stream.
 *         SPLIT representation (batches whose coupling window starts at cmin >= 256; p = the largest power of two
 *         <= cmin): with a = [0, p), b = [p, bsp) the block holds X_aa, X_bb and, where X_ba = -X_bb L_ba X_aa would
 *         be, the factor's own L_ba -- nobody multiplies with X_ba (C_i = B_i Linv_{i-1}^T reads X_bb only), and the
 *         sweeps apply X_aa, L_ba, X_bb in turn (same bytes, same flops).  gmrf_bt_get_block(LINV), export and the
 *         exact variances convert the factor to the full inverse in place; the last entry of the layout record says
 *         which form a factor is in (0: full).
 *   C     [batch][N-1][rmax][bsp - cmin]   the non-zero WINDOW of Cs[i]: rows 0 .. rmax, columns cmin ..
 *         (a FEM coupling block is zero outside it; inside, row tile t is zero left of cmin + kst[t]).
 *         The layout record {cmin, rmax, n_row_tiles, kst[...], split p} comes from the symbolic phase
 *         (gmrf_bt_get_layout) and travels with a broadcast factor (gmrf_bt_adopt_layout).
 *   L     [batch][N][bsp][bsp]      chos[i].L -- only F.chos / export / logdet read it.  With
 *         gmrf_bt_set_keep_l(h, 0) L_i lives in a one-block work buffer (log-determinant parts are taken
 *         during the factorisation): darcy256 1.60 -> 0.84 GB per posterior; F.chos is then unavailable.
 * gmrf_bt_factor_buffer gives base pointer and byte count of one array; gmrf_bt_block_range the element
 * range (per problem, plus the problem stride) that holds blocks [i0, i1) -- the unit of a block-range
 * broadcast (for kind C: the coupling blocks i0-1 .. i1-2). */
gmrf_status gmrf_bt_factor_buffer(gmrf_handle* h, int32_t kind, void** dev_ptr,
                                  int64_t* bytes);
gmrf_status gmrf_bt_block_range(gmrf_handle* h, int32_t kind, int64_t i0, int64_t i1,
                                int64_t* first_elem, int64_t* n_elems, int64_t* problem_stride);
gmrf_status gmrf_bt_set_keep_l(gmrf_handle* h, int32_t keep);
gmrf_status gmrf_bt_get_layout(gmrf_handle* h, int64_t* out, int64_t cap, int64_t* count);

/* Packed transport image of the blocks [i0, i1) -- the unit that moves when a factor is shared (over xGMI by
 * gmrf_bt_bcast_blocks_async, or by a transport of the caller's own).  Per problem one segment of
 * gmrf_bt_packed_size doubles: the lower-triangular 64 x 64 tiles of Linv_i0 .. Linv_{i1-1} (tile (r, c), c <= r, of
 * block i at ((i - i0) * nt (nt + 1) / 2 + r (r + 1) / 2 + c) * 4096, row-major; nt = bsp / 64 -- the tiles above the
 * block diagonal are zero and never read, so they do not travel: 136 of 256 tiles at bsp = 1024), the stored windows
 * of the coupling blocks C_{i0-1} .. C_{i1-2}, a two-double representation tag {split column of the SENDER's block inverses
 * when it packed (0: full inverses), 1196249670.0}, and the blocks' log-determinant parts ((i1 - i0) doubles, rounded up to
 * an even count).  The tag is authoritative: gmrf_bt_adopt_commit follows it, not the layout record (which can be older
 * than the image: the sender converts to the full inverses for gmrf_bt_get_block(LINV) / export / exact variances and goes
 * back to the split form at its next factorisation); raw-buffer transfers carry no tag and follow the record.  darcy256: 0.56 GB per posterior instead of the 0.83 GB of the raw Linv / C buffers.  dev_buf:
 * device memory, [batch][segment].  pack reads the handle's factor storage, unpack fills it (a receiving rank:
 * gmrf_bt_adopt_layout first, gmrf_bt_adopt_commit after the last range; gmrf_bt_logdet then works there too).
 * Both are enqueued on the handle's stream and return.  (No reference counterpart: the reference is one process.) */
gmrf_status gmrf_bt_packed_size(gmrf_handle* h, int64_t i0, int64_t i1, int64_t* elems_per_problem);
gmrf_status gmrf_bt_pack_blocks_async(gmrf_handle* h, int64_t i0, int64_t i1, double* dev_buf);
gmrf_status gmrf_bt_unpack_blocks_async(gmrf_handle* h, int64_t i0, int64_t i1, const double* dev_buf);

/* Caller-owned factor storage (e.g. torch tensors that RCCL broadcasts): upper bounds of the sizes
 * for a shape and batch (C at its dense size), then the three device buffers (dev_L may be NULL after
 * gmrf_bt_set_keep_l(h, 0)).  `batch` must equal the handle's batch.  Must be set before the first
 * factor / adopt of that shape; the buffers must outlive the handle's use of them; whatever the
 * handle had factored before is dropped. */
gmrf_status gmrf_bt_storage_bytes(int64_t n, int64_t n_blocks, int64_t batch, int64_t* bytes_L,
                                  int64_t* bytes_C, int64_t* bytes_Linv);
gmrf_status gmrf_bt_set_storage(gmrf_handle* h, int64_t n, int64_t n_blocks, int64_t batch,
                                void* dev_L, void* dev_C, void* dev_Linv);

/* A rank that RECEIVES the factor (broadcast): shape plus the root's layout record (adopt_shape =
 * dense coupling blocks), storage allocated WITHOUT factoring; once the buffers have been filled
 * adopt_commit marks the factor valid (l_blocks_valid != 0: the L buffer was filled too); it waits for the handle's stream
 * (the ranges must have been unpacked on it, or ordered behind it by gmrf_comm_wait) and reads the representation tag. */
gmrf_status gmrf_bt_adopt_layout(gmrf_handle* h, int64_t n, int64_t n_blocks,
                                 const int64_t* layout, int64_t count);
gmrf_status gmrf_bt_adopt_shape(gmrf_handle* h, int64_t n, int64_t n_blocks);
gmrf_status gmrf_bt_adopt_commit(gmrf_handle* h, int32_t l_blocks_valid);

/* ------------------------------------------------------------------ multi-GPU (RCCL over xGMI)
 * One process per GPU.  The factor is shared, right-hand sides / samples are sharded:
 * rank 0 factors block ranges (gmrf_bt_factor_begin_csc / _step_async) and each finished range is
 * broadcast (R1) while the next one is being factored; every rank then solves / samples its own
 * sample ids; variance accumulators are summed with one all-reduce (R2).
 *   rank 0:  gmrf_comm_unique_id(id)  -> ship the 128 bytes to the other processes by any means
 *   all:     gmrf_comm_create(device, rank, world, id, &c)
 *   ranks>0: gmrf_bt_adopt_layout(h, n, N, layout)   (layout: gmrf_bt_get_layout on rank 0, moved with
 *            gmrf_comm_bcast_host)
 *   per job, for each block range:  rank 0: gmrf_bt_factor_step_async(h, i0, i1);
 *                                   all:    gmrf_bt_bcast_blocks_async(h, c, 0, i0, i1, 0)   (one packed image per
 *                                           range: gmrf_bt_pack_blocks_async -> ncclBroadcast -> _unpack_)
 *            then  all: gmrf_comm_wait(h, c);  rank 0: gmrf_bt_factor_end;  ranks>0: gmrf_bt_adopt_commit(h, 0)
 * librccl is opened with dlopen on first use (GMRF_RCCL_PATH overrides the search). */
gmrf_status gmrf_comm_unique_id(void* id128);
gmrf_status gmrf_comm_create(int32_t device, int32_t rank, int32_t world, const void* id128,
                             gmrf_comm** out);
gmrf_status gmrf_comm_destroy(gmrf_comm* c);
gmrf_status gmrf_comm_bcast_host(gmrf_comm* c, void* host_buf, int64_t bytes, int32_t root);
gmrf_status gmrf_comm_allreduce_sum(gmrf_comm* c, gmrf_handle* stream_of, double* dev_buf,
                                    int64_t count);
gmrf_status gmrf_bt_bcast_blocks_async(gmrf_handle* h, gmrf_comm* c, int32_t root, int64_t i0,
                                       int64_t i1, int32_t with_l);
gmrf_status gmrf_comm_wait(gmrf_handle* h, gmrf_comm* c);
/* The all-gather form of sharing a BATCH of factors (round 4): every rank factors its own share of the batch on `src`
 * (src batch b) and the packed images of the blocks [i0, i1) are all-gathered (ncclAllGather) into `dst`, a handle of the same
 * shape with batch world * b that adopted src's layout record: problem r * b + p of dst = problem p of rank r.  Every rank
 * then takes means / samples of ALL world * b posteriors on dst (its own sample ids).  Against the root broadcast the
 * factorisation is spread over the ranks and every xGMI link carries 1 / world of the images instead of the root's links
 * carrying all of them.
 *   per job, for each block range:  all: gmrf_bt_factor_step_async(src, i0, i1);
 *                                        gmrf_bt_allgather_blocks_async(src, dst, c, i0, i1)
 *            then  all: gmrf_bt_factor_end(src);  gmrf_comm_wait(dst, c);  gmrf_bt_adopt_commit(dst, 0)
 * (No reference counterpart: the reference is one process.) */
gmrf_status gmrf_bt_allgather_blocks_async(gmrf_handle* src, gmrf_handle* dst, gmrf_comm* c, int64_t i0, int64_t i1);
/* Factor bytes this communicator has broadcast so far (reset != 0 clears the counter afterwards). */
gmrf_status gmrf_comm_bytes(gmrf_comm* c, int32_t reset, double* bytes);

/* Streams for several handles driven side by side (one host thread and one stream per handle: the latency-bound
 * launches of one factorisation leave CUs to the others -- bench.py runs 4 handles x 32 problems).  The HIP runtime
 * multiplexes streams onto a few hardware queues (4 by default) and two streams that land on the same queue
 * SERIALISE: measured on MI355X, 4 handles on 4 streams of which two share a queue deliver 27.3 k solves/s, on four
 * distinct queues 32.2 k (tools/stream_pairs.py).  gmrf_streams_create makes candidate streams
 * (hipStreamNonBlocking), runs a 1 ms spin kernel on pairs of them to find out which overlap, keeps n mutually
 * overlapping ones (as far as there are: *n_distinct tells how many of the returned streams are on queues of their
 * own) and destroys the rest.  streams: n hipStream_t, to be passed to gmrf_bt_create / released with
 * gmrf_streams_destroy.  (No reference counterpart: the reference is single-threaded CPU code.) */
gmrf_status gmrf_streams_create(int32_t device, int32_t n, void** streams, int32_t* n_distinct);
gmrf_status gmrf_streams_destroy(int32_t device, int32_t n, void** streams);

/* Pipelined factorisation (factor block ranges so a broadcast of finished blocks can
 * overlap): begin uploads the matrix, step_async enqueues blocks [i0, i1) and returns,
 * end synchronises and reports SPD failures. */
gmrf_status gmrf_bt_factor_begin_csc(gmrf_handle* h, int64_t n, int64_t n_blocks,
                                     const int64_t* colptr, const int64_t* rowval,
                                     const double* nzval, int32_t index_base);
gmrf_status gmrf_bt_factor_step_async(gmrf_handle* h, int64_t i0, int64_t i1);
gmrf_status gmrf_bt_factor_end(gmrf_handle* h, int32_t* info);

/* Batch of B independent problems that share ONE sparsity pattern (the reference's loop over
 * data-set problems, scripts/darcy/solve_darcy_gmrf-fem.jl:176-198): they are factored and
 * solved in lock step, problem = one more grid dimension of every kernel.  With B > 1:
 * nzval holds B value arrays one after the other; b / y / z / out hold B consecutive groups of
 * k columns (problem-major); mean holds B vectors; sample ids are first_id + p*k + s.
 * Accessors (get_block, logdet) address the problem chosen by gmrf_bt_select_problem. */
gmrf_status gmrf_bt_set_batch(gmrf_handle* h, int64_t batch);
gmrf_status gmrf_bt_select_problem(gmrf_handle* h, int64_t p);

gmrf_status gmrf_bt_stats(gmrf_handle* h, gmrf_stats* out);
gmrf_status gmrf_bt_set_profiling(gmrf_handle* h, int32_t level);
/* bit 0: plain stream launches instead of replaying captured HIP graphs; bit 1: three-launch
 * panel step also for batch 1; bit 2: keep 64-multiples of right-hand sides on sweep_mm;
 * bit 3: C = B X^T by the dense GEMM even when the lower blocks are sparse; bit 4: second
 * branch in the captured factor graph (inverse assembly beside the panel chain; experiment); bit 5:
 * ignore the staircase of the coupling blocks (dense window; takes effect at the next factor_csc); bit 7:
 * one problem assembles Linv by recursive doubling after the panel steps instead of row by row inside them; bit 8:
 * one problem re-factors the diagonal tile in every workgroup of a step instead of the look-ahead chain;
 * bit 12: batches always assemble the full block inverses (no split representation; comparison); bit 13: no persistent
 * launches (potrf_persist) -- the launch-per-step forms; bit 15: small batches keep potrf_diag128 + the 128^3 products
 * instead of one persistent launch per panel diagonal block; bit 16: one problem's sweeps keep one launch per product
 * instead of ONE persistent launch per sweep (sweep_persist; comparison -- bit 13 switches both persistent forms off); bit 17: one problem
 * keeps scatter_block (S += D_i) as a launch of its own behind S = -C C^T instead of zero + scatter inside the spmm_bxt_tiles launch and an
 * accumulating product (comparison; same bits).
 * (Bits 6, 9, 10, 11, 14 selected comparison routes that lost twice -- left-looking panels, in-panel updates on the GEMM kernel,
 * rank-64 panel steps of batches, 128-column panels, potrf_panel256 -- and were removed with them in round 5; they are ignored.) */
gmrf_status gmrf_bt_set_eager(gmrf_handle* h, int32_t eager);
gmrf_status gmrf_bt_synchronize(gmrf_handle* h);

/* ------------------------------------------------------------------ K6: CSR SpMV / SpMM
 * Device-resident sparse matrix built from CSR arrays (rowptr has n_rows+1 entries) --
 * or, for a symmetric matrix, from the CSC arrays of the same matrix.  values_f32 != 0
 * stores the values in fp32 and accumulates in fp64 (BASELINE config 5). */
gmrf_status gmrf_csr_create(int32_t device, void* stream, int64_t n_rows, int64_t n_cols,
                            const int64_t* rowptr, const int64_t* colidx,
                            const double* vals, int32_t index_base, int32_t values_f32,
                            gmrf_csr** out);
gmrf_status gmrf_csr_destroy(gmrf_csr* m);
/* Y = S X ;  X: n_cols x k, Y: n_rows x k, column-major with leading dimensions ldx, ldy. */
gmrf_status gmrf_spmm(const gmrf_csr* S, const double* X, double* Y, int64_t k,
                      int64_t ldx, int64_t ldy);
/* The same product with NODE-MAJOR operands: X is n_cols x k with the k values of one column index
 * contiguous (row stride ldx >= k), Y likewise -- in Julia the k x n matrices permutedims(X),
 * permutedims(Y).  This is the layout of the LDS-tiled kernel (csr_spmm_tiles): one gathered column
 * index fetches k contiguous doubles, and the distinct rows of X a 64-row tile needs are staged in
 * LDS once (the RBMC estimator of `std(x)` runs on it: 50 samples x Q * x,
 * scripts/darcy/solve_darcy_gmrf-fem.jl:100,192). */
gmrf_status gmrf_spmm_rows(const gmrf_csr* S, const double* X, double* Y, int64_t k,
                           int64_t ldx, int64_t ldy);
/* Stream-ordered variants for DEVICE operands: the product is enqueued on the matrix's stream (gmrf_csr_create) and
 * the call returns without synchronising -- `Q * x` inside a device-resident loop. */
gmrf_status gmrf_spmm_async(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy);
gmrf_status gmrf_spmm_rows_async(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy);

/* ------------------------------------------------------------------ posterior assembly
 * The step before the factorisation in the reference's Gauss-Newton loop
 * (scripts/solve_burger.jl:143-149) and in `condition_on_observations`:
 *     A   = Q + noise * J' * J                       (gmrf_assemble_precision)
 *     rhs = base + noise * J' * (J * x + obs_diff)   (gmrf_assemble_rhs; base = Q * x_prior)
 * Patterns are fixed, values change per iteration.  create() does the symbolic work on the host
 * (Q: CSC n x n with ascending rows per column, both triangles; J: CSR m x n); the numeric calls
 * take host or device pointers, and the device output of gmrf_assemble_precision is the `nzval`
 * array of the pattern gmrf_assemble_pattern returns -- feed it to gmrf_bt_factor_csc once and to
 * gmrf_bt_refactor_values afterwards.  device = -1: symbolic only (pattern queries). */
typedef struct gmrf_assembler gmrf_assembler;
gmrf_status gmrf_assemble_create(int32_t device, void* stream, int64_t n, const int64_t* q_colptr,
                                 const int64_t* q_rowval, int64_t m, const int64_t* j_rowptr,
                                 const int64_t* j_colidx, int32_t index_base, gmrf_assembler** out);
gmrf_status gmrf_assemble_destroy(gmrf_assembler* as);
/* nnz of the result, number of J[k,i] * J[k,j] products, and (if not NULL) its CSC pattern. */
gmrf_status gmrf_assemble_pattern(const gmrf_assembler* as, int64_t* nnz_out, int64_t* n_products,
                                  int64_t* colptr, int64_t* rowval, int32_t index_base);
gmrf_status gmrf_assemble_precision(gmrf_assembler* as, const double* q_nzval, const double* j_vals,
                                    double noise, double* out_nzval);
/* obs_diff and base may be NULL (zero). */
gmrf_status gmrf_assemble_rhs(gmrf_assembler* as, const double* base, const double* j_vals,
                              const double* x, const double* obs_diff, double noise, double* out);

/* ------------------------------------------------------------------ FEM block assembly (first piece)
 * Darcy stiffness matrix  G[i][j] = int a grad(phi_i).grad(phi_j),  f[i] = beta int phi_i  with the
 * homogeneous Dirichlet rows / columns applied -- assemble_darcy_diff_matrix,
 * /root/reference/src/problems/darcy.jl:5-63 (cell loop :27-60, coefficient looked up at the quadrature
 * point by nearest grid point, src/datasets/darcy.jl:30-34; `apply!` :61) -- on the structured P1 mesh
 * of the BASELINE Darcy configs: nx x ny nodes on the unit square, x fastest, quads cut by the diagonal
 * n00 - n11, one quadrature point per cell.  The values come out in CSR order of the 7-point
 * pattern gmrf_darcy_p1_pattern returns, which is the `J` of gmrf_assemble_precision
 * (Q_post = Q + Q_eps G'G): per problem only the ng x ng coefficient table crosses the bus.
 * coeff_table[ix * ng + iy] = a at grid point (x = ix / (ng-1), y = iy / (ng-1)); host or device
 * pointers; device = -1: pattern only. */
typedef struct gmrf_darcy_p1 gmrf_darcy_p1;
gmrf_status gmrf_darcy_p1_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_darcy_p1** out);
/* The same assembler with the reference's own element (src/utils.jl:32-33): Lagrange{RefTriangle,2} and the 4-point rule of
 * QuadratureRule{RefTriangle}(3) on the nx x ny vertex mesh; dofs = the (2 nx - 1) x (2 ny - 1) lattice of vertices and edge
 * midpoints, x fastest; the coefficient is looked up at every quadrature point.  Same handle type: gmrf_darcy_p1_pattern /
 * _assemble / _destroy serve both (n = (2 nx - 1)(2 ny - 1) rows, up to 19 entries per row). */
gmrf_status gmrf_darcy_p2_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_darcy_p1** out);
gmrf_status gmrf_darcy_p1_destroy(gmrf_darcy_p1* d);
gmrf_status gmrf_darcy_p1_pattern(const gmrf_darcy_p1* d, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx,
                                  int32_t index_base);
gmrf_status gmrf_darcy_p1_assemble(gmrf_darcy_p1* d, const double* coeff_table, int64_t ng, double beta,
                                   double* vals_out, double* f_out);

/* Residual and tangent of the implicit-Euler Burgers space-time system (FEM block assembly, second piece):
 *     J(w) = J_static + dt J_adv(w),     f(w) = J_static w + dt v_adv(w)
 * -- f_and_J / nonlinear_primal_tangent, /root/reference/scripts/burgers/solve_burgers_gmrf-fem.jl:118-149, with
 * J_static = M_{t+1} - M_t + dt nu G_{t+1} (:123-130; assemble_burgers_mass_diffusion_matrices,
 * src/problems/burgers.jl:60-98) and assemble_burgers_advection_matrix (src/problems/burgers.jl:5-59, cell loop
 * :22-51) per time slice -- on the periodic P1 line of the BASELINE Burgers configs: ns nodes on [0,1), nt time
 * slices, time-major index (t-1) ns + s, 3-point Gauss rule.  J has (nt-1) ns rows (slices 2 .. nt), nt ns columns
 * and 6 entries per row; the values come out in the CSR order of gmrf_burgers_p1_pattern, which is the `J` that
 * gmrf_assemble_precision / gmrf_assemble_rhs take: a Gauss-Newton iteration (scripts/solve_burger.jl:143-149)
 * moves only w across the bus, or nothing at all with device pointers.  w, vals_out ((nt-1) ns 6), f_out
 * ((nt-1) ns): host or device pointers.  device -1: pattern only. */
typedef struct gmrf_burgers_p1 gmrf_burgers_p1;
gmrf_status gmrf_burgers_p1_create(int32_t device, void* stream, int64_t ns, int64_t nt, double dt, double nu,
                                   gmrf_burgers_p1** out);
/* The same on the QUADRATIC periodic line, the element the reference's Burgers scripts run on
 * (periodic_unit_interval_discretization, /root/reference/src/utils.jl:42-49: QuadraticLine cells, Lagrange{RefLine,2},
 * 3-point Gauss rule): ns = 2 N_x dofs numbered by position, even = cell boundaries, odd = midpoints; vertex rows of J hold
 * 10 entries (columns i-2 .. i+2 of slices t-1 and t), midpoint rows 6.  The handle is a gmrf_burgers_p1: _pattern,
 * _tangent and _destroy serve both element orders. */
gmrf_status gmrf_burgers_p2_create(int32_t device, void* stream, int64_t ns, int64_t nt, double dt, double nu,
                                   gmrf_burgers_p1** out);
gmrf_status gmrf_burgers_p1_destroy(gmrf_burgers_p1* b);
gmrf_status gmrf_burgers_p1_pattern(const gmrf_burgers_p1* b, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx,
                                    int32_t index_base);
gmrf_status gmrf_burgers_p1_tangent(gmrf_burgers_p1* b, const double* w, double* vals_out, double* f_out);

/* Linear shallow-water SPDE (FEM block assembly, third piece): the element loops of `assemble_system!`,
 * /root/reference/src/spdes/shallow_water.jl:17-122 -- coupling matrix K (h-u, h-v: -H grad(phi_i) phi_j; u-h, v-h:
 * -g grad(phi_i) phi_j; u-u, v-v: k phi_i phi_j; u-v / v-u: -+ f phi_i phi_j), element-lumped mass M (`lump_matrix`, :116),
 * stiffness S, `apply!` of the constraint handler to all three (:119-121) -- and the per-step operators `discretize` forms
 * from them (:170-217), on the structured P1 triangle mesh of the Darcy configs (nx x ny nodes, x fastest, quads cut by the
 * diagonal n00 - n11; cell numbering: all lower triangles (n00, n10, n11), then all upper (n00, n11, n01)) with
 *   dof = 3 * node + field, fields (h, u, v) = (0, 1, 2)        (Ferrite numbers dofs in cell-visit order: a permutation)
 *   the symmetric 3-point quadrature rule of order 2; H enters through H_q[cell][q] = H(x_q), x_q from _qpoints.
 * pattern 0: K, 3 nn rows with the full field coupling over the 7-point node stencil (create_sparsity_pattern(dh, ch),
 * :140; explicit zeros kept); pattern 1: S (and M~ / the Matern factor), block-diagonal coupling (:141-150).  M is lumped and
 * comes back as a vector.  prescribed: 3 nn bytes (non-zero = dof of the constraint handler) or NULL.
 * _operators, for one time step dt: G_vals = (M~ + dt K) with the constraints applied (K's pattern, :212-217), J_vals =
 * sqrt(ratio) M~^-1/2 (kappa^2 M~ + G) in S's pattern -- the transposed square root of the initial precision, Q_matern = J'J
 * (:178-189; feed J to gmrf_assemble_precision with a zero Q), M_tilde (:172-174), beta(dt) = sqrt(dt) tau (1e-2 on prescribed
 * dofs, :198-211).  Values: host or device pointers; device -1: patterns and quadrature points only. */
typedef struct gmrf_swe_p1 gmrf_swe_p1;
gmrf_status gmrf_shallow_water_p1_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_swe_p1** out);
gmrf_status gmrf_shallow_water_p1_destroy(gmrf_swe_p1* w);
gmrf_status gmrf_shallow_water_p1_pattern(const gmrf_swe_p1* w, int32_t which, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx,
                                          int32_t index_base);
gmrf_status gmrf_shallow_water_p1_qpoints(const gmrf_swe_p1* w, double* xy);   /* xy: [cells][3][2] doubles in host memory */
gmrf_status gmrf_shallow_water_p1_assemble(gmrf_swe_p1* w, const double* H_q, double k, double f, double g, const uint8_t* prescribed,
                                           double* K_vals, double* M_lumped, double* S_vals);
gmrf_status gmrf_shallow_water_p1_operators(gmrf_swe_p1* w, const double* K_vals, const double* M_lumped, const double* S_vals,
                                            const uint8_t* prescribed, double kappa_matern, double tau, double dt, double* G_vals,
                                            double* J_vals, double* M_tilde, double* beta);

/* ------------------------------------------------------------------ test hooks
 * Direct access to the dense device kernels for the parity tests (row-major operands on
 * the HOST; not part of the drop-in surface). */
gmrf_status gmrf_test_gemm(int32_t device, int64_t M, int64_t N, int64_t K, int32_t transA,
                           int32_t transB, int32_t tri_flags, int32_t lower_only,
                           double alpha, const double* A, int64_t lda, const double* B,
                           int64_t ldb, double beta, double* C, int64_t ldc);
/* Device-resident timing of one GEMM shape (random operands): batch problems, big = 1 takes the
 * 128 x 128 kernel, 0 the 64 x 64 one, 2 the launcher's own choice. */
gmrf_status gmrf_test_gemm_rate(int32_t device, int64_t M, int64_t N, int64_t K, int32_t transB,
                                int32_t tri_flags, int32_t lower_only, int32_t batch,
                                int32_t big, int32_t reps, double* ms_per_launch);
/* GEMM launches of the profiled steps of a handle by shape (gmrf_bt_set_profiling on): rows of 11 doubles = class, M, N, K,
 * tri flags, lower_only, problems, per-tile K bounds?, launches, ms (the dispatches' own time stamps), flops as booked. */
gmrf_status gmrf_test_gemm_shapes(gmrf_handle* h, double* rows, int64_t cap_rows, int64_t* n_rows);
gmrf_status gmrf_test_potrf_tile(int32_t device, double* tile64 /* in: SPD, out: L */,
                                 double* inv64, int32_t* info);
gmrf_status gmrf_test_tile_timing(double* out, int32_t n);
gmrf_status gmrf_test_potrf_block(int32_t device, int64_t bs, double* S /* in/out: L */,
                                  double* Linv, int32_t* info);
/* s_memtime stamps of the chain workgroup of the last gmrf_test_potrf_block that took the persistent form
 * (csrc/potrf_persist.hpp): out[0] = tile 0 done, then eight per step (tools/persist_stamps.py names them); cycles relative to out[0], -1 = not written. */
gmrf_status gmrf_test_persist_stamps(double* out, int32_t n);
/* block ranges of this handle that were repeated with the launch-per-step form because a bounded wait inside a persistent
 * launch gave up (GMRF_PERSIST_SPIN_MS: default 2000 for one problem, 200 for batches); also in gmrf_stats.persist_aborts */
gmrf_status gmrf_test_persist_aborts(gmrf_handle* h, int32_t* n);
/* The per-device budget of CUs for persistent launches, host only (no GPU needed): `n` handles ask for demands[i] CUs one after
 * the other on a device of `cus` CUs; granted[i] = 1 if the claim fitted beside the earlier ones, 0 if it was refused (the
 * handle would take the launch-per-step routes up front instead of meeting a bounded wait). */
gmrf_status gmrf_test_persist_budget(int32_t cus, int32_t n, const int32_t* demands, int32_t* granted);
gmrf_status gmrf_test_mfma_f64_rate(int32_t device, double* tflops);
/* Shader clock under load: _start launches a bounded probe (8 waves stamping s_memtime / s_memrealtime every ~3.4 us x sleeps,
 * n samples) on a stream of its own and returns; run the load under test; _finish waits and returns n - 1 interval clocks in GHz
 * (median over the 8 waves) and the intervals' start times in ms. */
gmrf_status gmrf_test_clock_probe_start(int32_t device, int32_t n, int32_t sleeps, void** probe);
gmrf_status gmrf_test_clock_probe_finish(void* probe, double* ghz, double* t_ms);
gmrf_status gmrf_test_hbm_rate(int32_t device, int64_t bytes, double* gbps);
gmrf_status gmrf_test_microbench(int32_t device, double* out, int32_t n);
/* Host-only part of the symbolic phase of gmrf_bt_factor_csc (needs no device): driven by the sanitizer build of the host
 * side.  out8 = {cmin, rmax, entries, longest lower-block row, sparse route?, tile plan?, plan capacity, checksum}. */
gmrf_status gmrf_test_symbolic_csc(int64_t n, int64_t n_blocks, const int64_t* colptr, const int64_t* rowval,
                                   int32_t index_base, int64_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* GMRF_HIP_H */
