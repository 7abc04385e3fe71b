// libgmrf_hip.so -- C ABI (include/gmrf_hip.h) over the gfx950 kernels of this directory.
//
// Host orchestration of the block-tridiagonal Cholesky path of DiffEqGMRFs.jl
// (/root/reference/src/tridiagonal_cholesky.jl): symbolic analysis of the CSC matrix into
// per-block entry lists, then per block  scatter -> C = B Linv^T -> S = D - C C^T ->
// potrf(S) by 64-wide panels -> Linv by recursive doubling;  sweeps as chains of
// matrix-panel products.  Launch sequences are captured once into HIP graphs and replayed
// (the chains are launch-latency bound: ~60 dependent launches per block).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <iterator>
#include <cmath>
#include <map>
#include <chrono>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/gmrf_hip.h"
#include "assemble.hpp"
#include "fem_assemble.hpp"
#include "gemm_f64.hpp"
#include "gemm_f64_dma.hpp"
#include "microbench.hpp"
#include "misc_kernels.hpp"
#include "potrf_step.hpp"
#include "potrf_persist.hpp"
#include "sweep.hpp"
#include "sweep_persist.hpp"
#include "swe_assemble.hpp"
#include "fem_assemble_p2.hpp"

using namespace gmrf;

static thread_local std::string g_last_error;
static double g_tile_us = 0.0;
static unsigned long long g_tile_stamps[64];
static unsigned long long g_persist_stamps[128];    // chain workgroup of the last gmrf_test_potrf_block (potrf_persist.hpp)

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
            return GMRF_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define GCHK(expr)                                                                          \
    do {                                                                                    \
        gmrf_status _s = (expr);                                                            \
        if (_s != GMRF_OK) return _s;                                                       \
    } while (0)

static gmrf_status bad_shape(const char* msg) {
    g_last_error = msg;
    return GMRF_ERR_BAD_SHAPE;
}

static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // unregistered host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static int64_t next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------ csr
struct gmrf_csr {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int64_t* d_rowptr = nullptr;
    int32_t* d_colidx = nullptr;
    double* d_vals = nullptr;
    float* d_vals32 = nullptr;
    double* d_diag = nullptr;
    bool tiles128_ok = false;          // the same for 128-row tiles and 2 * SPMV_CAP
    bool tiles_ok = false;             // every SPMV_ROWS-row tile has at most SPMV_CAP entries (csr_spmv_tiles)
    double* d_stage_x = nullptr;
    double* d_stage_y = nullptr;
    int64_t stage_cap = 0;
    // tile plan of csr_spmm_tiles (node-major right-hand sides), built on first use
    int plan_state = 0;                // 0: not built, 1: usable, -1: a tile exceeds the LDS image
    int64_t* d_tile_uptr = nullptr;    // [tiles + 1] ranges into d_ucols
    int32_t* d_ucols = nullptr;        // distinct columns of every tile, ascending
    uint16_t* d_lidx = nullptr;        // per entry: index of its column in the tile's list
    int64_t n_ucols = 0;
    int plan_rows = 0, plan_ucap = 0, plan_ecap = 0, plan_ecap_pad = 0, plan_umax = 0;   // rows per tile, LDS capacities (distinct columns, entries)
};

// ------------------------------------------------------------------------------------ handle
struct EvPair {
    hipEvent_t a, b;
    int kind;       // kernel class, see gmrf_stats
    double work;    // flops or bytes
    int shape = -1; // GEMM launches: index into gmrf_handle::gemm_shapes
};

// one distinct GEMM launch shape of a profiled run (gmrf_test_gemm_shapes): the roofline table by shape
struct GemmShapeStat {
    int64_t key[8];     // class, M, N, K, tri, lower_only, problems, per-tile K bounds?
    double launches = 0, ms = 0, work = 0;
};

struct gmrf_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, N = 0, bs = 0, bsp = 0, n_pad = 0;
    int64_t B = 1;                     // independent problems factored / solved in lock step
    int64_t sel = 0;                   // problem the accessors (get_block, logdet, ...) address
    int64_t alloc_B = 0, vals_B = 0;
    // symbolic
    int64_t nnz_in = 0, n_entries = 0;
    uint64_t* d_keys = nullptr;
    double* d_vals = nullptr;
    int64_t* d_src = nullptr;          // entry -> index into the caller's nzval
    double* d_nz_stage = nullptr;      // staging for host nzval
    std::vector<int64_t> diag_first, diag_count, low_first, low_count;
    bool analyzed = false;
    // Layout of the coupling blocks C_i of the CURRENT factor (set by the symbolic phase, by
    // gmrf_bt_adopt_layout for a factor received by broadcast, "dense" for an imported image): C_i is
    // zero left of column cmin and below row rmax (64-aligned), and inside that window row tile t
    // (64 rows) is zero left of column cmin + kst[t] -- the staircase a FEM coupling block has
    // (kst: monotone envelope over all blocks).  C is STORED as the window only: [rmax][bsp - cmin].
    int64_t cmin = 0, rmax = 0;
    std::vector<int> kst, mend;        // mend[u]: column tile u (64 columns from cmin) is zero below row mend[u]
    int* d_kst = nullptr;              // device copies: [bsp / 64] each
    int* d_mend = nullptr;
    int* d_kbx = nullptr;              // [bsp / 64]: max(0, 64 u - cmin): first non-zero row of column tile u of Linv_i[cmin:, :] (selected inversion)
    int kst_cap = 0;
    double c_streamed = 0.0;           // doubles of one C_i inside the staircase (what a k = 1 sweep reads)
    double g2_tile_k = 0.0;            // sum over lower tiles of the K extent of S = -C C^T (flop accounting)
    bool c_dirty = false;              // C must be re-zeroed (new pattern)
    int* d_lo_rowptr = nullptr;
    int* d_dg_rowptr = nullptr;       // [N][bsp + 1]: the diagonal blocks' entries row by row (absolute positions in d_keys)
    // tile plan of the sparse C = B X^T (spmm_bxt_tiles): per lower block and 64-row tile the distinct columns, per entry its index into them
    int* d_bxt_uptr = nullptr; int* d_bxt_ucols = nullptr; uint16_t* d_bxt_lidx = nullptr; int* d_bxt_gtiles = nullptr;
    std::vector<int> bxt_ng;           // groups of row tiles per lower block (host copy: the launch's grid)
    int bxt_ecap = 0; int bxt_nrt = 0; bool bxt_plan_ok = false;        // [N][bsp + 1] row-wise view of the lower blocks' entry lists
    int64_t lo_row_max = 0;            // most entries in one row of a lower block
    bool sparse_b = false;             // lower blocks are sparse enough for C = B X^T by spmm_bxt
    bool dense_g1 = false;             // force the dense GEMM for C = B X^T (comparison)
    // factor storage
    double *d_L = nullptr, *d_C = nullptr, *d_Linv = nullptr;
    bool external_storage = false;
    bool keep_l = true;                // false: L_i lives in a one-block work buffer (sweeps need Linv and C only)
    bool l_valid = false;              // d_L holds the blocks of the current factor (get_block / export / logdet from L)
    bool logdet_valid = false;         // d_logdet holds every block's log-determinant part of the current factor
    std::vector<char> got_block;       // blocks received through gmrf_bt_unpack_blocks_async since the last adopt / commit
    int64_t alloc_rm = 0, alloc_wc = 0;
    bool alloc_keep_l = true;
    double *d_S = nullptr, *d_B = nullptr, *d_T = nullptr, *d_W = nullptr;
    double* d_V = nullptr;             // two more work blocks per problem, allocated by the first exact-variance call (var_exact)
    int64_t v_elems = 0;
    int* d_info = nullptr;
    double* d_logdet = nullptr;
    int64_t alloc_N = 0, alloc_bsp = 0;
    bool factored = false;
    // right-hand-side panels
    double *d_P = nullptr, *d_Y = nullptr, *d_Tp = nullptr;
    int64_t kp_cap = 0;
    double* d_stage = nullptr;
    int64_t stage_cap = 0;
    double* d_mean = nullptr;
    double* d_acc = nullptr;           // variance accumulator / output ([acc_B][n])
    int64_t acc_B = 0;
    // graphs
    bool eager = false;
    bool split_step = false;           // use the three-launch panel step also for batch 1 (experiment)
    bool sweep_no_gemm = false;        // keep 64-multiples of right-hand sides on sweep_mm (comparison)
    unsigned long long* dbg_stamps = nullptr;   // test hook: phase stamps of the fused panel step
    bool fork_graph = false;           // second branch in the captured factor graph (experiment, see potrf_block)
    bool no_staircase = false;         // treat the coupling window as dense (comparison; takes effect at the next analysis)
    bool doubling_x = false;           // one problem: assemble Linv by recursive doubling after the steps (comparison) instead of row by row inside them
    // Split representation of the block inverses (batches whose coupling blocks are zero left of column cmin >= 256):
    // with p = xsplit, a = [0, p), b = [p, bsp), the storage of Linv_i holds X_aa, X_bb and -- in the place of
    // X_ba = -X_bb L_ba X_aa -- the factor's own L_ba.  C_i = B_i Linv_{i-1}^T reads X_bb only; the sweeps apply
    // y_a = X_aa t_a, y_b = X_bb (t_b - L_ba y_a) (and the transposed order backward).  0: the full inverse.
    int xsplit = 0;                    // state of the stored factor
    int adopt_xsplit = 0;              // what the layout record of an adopted factor said (committed by gmrf_bt_adopt_commit)
    bool no_xsplit = false;            // set_eager bit 12: always assemble the full inverse (comparison)
    bool no_lookahead = false;         // one problem: every fused step re-factors its diagonal tile (comparison) instead of the look-ahead chain
    // One problem: the in-block Cholesky as ONE persistent launch per block / per 256-column panel (potrf_persist.hpp) instead of a
    // launch per 64-column step.  Needs every workgroup resident (1 + tiles <= CUs); a wait that gives up sets d_info[1], and
    // factor_finish then repeats the factorisation with the launch-per-step form and keeps it for this handle.
    bool no_persist = false;           // set_eager bit 13 / GMRF_PERSIST=0
    bool persist_gave_up = false;      // a bounded wait inside a persistent launch gave up: this handle keeps the launch-per-step forms for good (set_eager cannot clear it)
    bool persist_launched = false;     // a persistent launch was enqueued since the abort word was last looked at
    int persist_cus = 0;               // CUs held of the device's budget (persist_plan); 0: no persistent launches
    int info_checked = 0;              // d_info[0] as the host last saw it (restored when an aborted range is repeated)
    int factor_graph_route = 0;        // stats.persist_route of the captured factor graph
    int persist_aborts = 0;
    int cu_count = 0;
    unsigned* d_pflags = nullptr;      // flag words of the persistent launches (zero between launches: potrf_persist cleans up after itself)
    int64_t pflags_words = 0;
    // One problem: a whole sweep as ONE persistent launch (sweep_persist.hpp) instead of two dependent launches per block.  Every
    // workgroup must be resident: the handle then claims the whole chip (persist_plan).  A wait that gives up sets the mapped host
    // word; gmrf_bt_solve / gmrf_bt_sample see it at their synchronisation and repeat the call with the launch-per-product form.
    bool no_sweep_persist = false;     // set_eager bit 16 / GMRF_SWEEP_PERSIST=0
    bool no_scatter_fold = false;      // set_eager bit 17: one problem keeps scatter_block as a launch of its own behind S = -C C^T (comparison)
    bool sweep_persist_planned = false;   // the claim covers the whole chip (persist_plan)
    bool sweep_persist_launched = false;  // such a launch was enqueued since the abort word was last looked at
    unsigned* d_sweep_flags = nullptr;    // [0] the device's abort word
    double* d_Tsw = nullptr;              // intermediate panel of the persistent sweeps (right-hand side minus coupling product)
    int64_t t_elems = 0;
    unsigned* h_sweep_abort = nullptr;    // mapped host word
    int sweep_nw = 0;
    double *d_P2 = nullptr, *d_Y2 = nullptr, *d_T2 = nullptr;     // gmrf_bt_posterior: the samples' panels (their sweep runs beside the mean's)
    int64_t p2_elems = 0;
    bool sweep_persist_hold = false;      // this call must not use it (its input would be lost if the launch gave up: in-place solve)
    bool no_persist_panels = false;    // batches small enough for it keep potrf_diag128 + the 128^3 products instead of one persistent launch per panel (set_eager bit 15)
    // second branch of the captured factor graph: the inverse assembly of a block's first half runs
    // beside the panel chain of its second half (see potrf_block)
    hipStream_t aux = nullptr;
    bool aux_distinct = false;         // `aux` was chosen on a hardware queue of its own (gmrf_bt_posterior)
    hipStream_t gemm_stream = nullptr; // stream gemm() launches on (h->stream unless inside the aux branch)
    bool capturing = false;
    std::vector<hipEvent_t> fork_events;
    size_t fork_next = 0;
    hipGraphExec_t factor_graph = nullptr;
    int64_t factor_graph_i0 = -1, factor_graph_i1 = -1;
    std::map<int64_t, hipGraphExec_t> sweep_graphs;   // key = mode * 4096 + kp
    // profiling
    int profiling = 0;
    std::vector<EvPair> events;
    std::vector<hipEvent_t> ev_pool;
    std::vector<GemmShapeStat> gemm_shapes;
    gmrf_stats stats;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static void destroy_graphs(gmrf_handle* h) {
    if (h->factor_graph) { (void)hipGraphExecDestroy(h->factor_graph); h->factor_graph = nullptr; }
    for (auto& kv : h->sweep_graphs) (void)hipGraphExecDestroy(kv.second);
    h->sweep_graphs.clear();
}

static hipEvent_t ev_get(gmrf_handle* h) {
    if (!h->ev_pool.empty()) {
        hipEvent_t e = h->ev_pool.back();
        h->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    gmrf_handle* h;
    EvPair p;
    bool on;
    ProfScope(gmrf_handle* hh, int kind, double work) : h(hh), on(hh->profiling > 0) {
        if (on) {
            p.kind = kind; p.work = work;
            p.a = ev_get(h); p.b = ev_get(h);
            (void)hipEventRecord(p.a, h->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(p.b, h->stream);
            h->events.push_back(p);
        }
    }
};

static void prof_collect(gmrf_handle* h) {
    for (auto& p : h->events) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        h->stats.kernel_ms[p.kind] += ms; h->stats.kernel_work[p.kind] += p.work; h->stats.kernel_launches[p.kind]++;
        if (p.shape >= 0) { auto& g = h->gemm_shapes[(size_t)p.shape]; g.launches += 1; g.ms += ms; g.work += p.work; }
        h->ev_pool.push_back(p.a);
        h->ev_pool.push_back(p.b);
    }
    h->events.clear();
}

// ------------------------------------------------------------------------------------ helpers
static gmrf_status gemm(gmrf_handle* h, bool a_t, bool b_n, int M, int N, int K, int tri, int lower_only,
                        double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                        double beta, double* C, int64_t ldc, int64_t pA, int64_t pB, int64_t pC,
                        int batch = 1, int64_t sA = 0, int64_t sB = 0, int64_t sC = 0, const double* D = nullptr,
                        int64_t ldd = 0, int64_t pD = 0, int pclass = 0, double pwork = -1.0,
                        const int* kb_m = nullptr, const int* kb_n = nullptr, const int* ke_n = nullptr) {
    GemmArgs g;
    g.kb_m = kb_m; g.kb_n = kb_n; g.ke_n = ke_n;
    g.D = D; g.ldd = ldd; g.pD = pD;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.strideA = sA; g.strideB = sB; g.strideC = sC;
    g.pA = pA; g.pB = pB; g.pC = pC; g.nb1 = batch;
    g.M = M; g.N = N; g.K = K; g.tri = tri; g.lower_only = lower_only;
    g.alpha = alpha; g.beta = beta; g.stamps = nullptr;
    double flops = 2.0 * M * N * (double)K * batch * (double)h->B;
    if (lower_only) flops *= 0.5 * (1.0 + 64.0 / std::max(M, 64));
    if (tri) flops *= 0.5 * (1.0 + 64.0 / std::max(K, 64));
    // statistics: launches of the 128 x 128 kernel are their own classes (6: B stored [n][k],
    // 7: B stored [k][n]) whoever calls, so that a class is one kernel symbol of a rocprof trace
    if (gemm_uses_big(a_t, g, batch * (int)h->B)) { pclass = b_n ? 7 : 6; pwork = -1.0; }
    else if (pclass == 0) {
        if (gemm_uses_ll(g, batch * (int)h->B)) pclass = 13;
        else if (gemm_uses_dma(a_t, g, batch * (int)h->B)) pclass = a_t ? 18 : (b_n ? 15 : 14);   // gemm_f64_dma<.., B [n][k]> / <.., B [k][n]> / <.., A [k][m]>
        else pclass = a_t ? 12 : (b_n ? 11 : 0);
    }
    if (h->profiling > 0) {
        // GEMM classes: the dispatch's own begin / end time stamps (see launch_gemm), not an event pair around the launch
        EvPair p;
        p.kind = pclass; p.work = pwork >= 0.0 ? pwork : flops;
        p.a = ev_get(h); p.b = ev_get(h);
        const int64_t key[8] = {pclass, M, N, K, tri, lower_only, batch * h->B, (kb_m || kb_n || ke_n) ? 1 : 0};
        for (size_t i = 0; i < h->gemm_shapes.size() && p.shape < 0; ++i)
            if (!memcmp(h->gemm_shapes[i].key, key, sizeof(key))) p.shape = (int)i;
        if (p.shape < 0) {
            GemmShapeStat st; memcpy(st.key, key, sizeof(key));
            h->gemm_shapes.push_back(st); p.shape = (int)h->gemm_shapes.size() - 1;
        }
        HIPCHK(launch_gemm(h->gemm_stream ? h->gemm_stream : h->stream, a_t, b_n, g, batch * (int)h->B, p.a, p.b));
        h->events.push_back(p);
        return GMRF_OK;
    }
    HIPCHK(launch_gemm(h->gemm_stream ? h->gemm_stream : h->stream, a_t, b_n, g, batch * (int)h->B));
    return GMRF_OK;
}

static void free_dev(void* p) {
    if (p) (void)hipFree(p);
}

// element counts / strides of the factor arrays (doubles)
static inline int64_t blk_elems(const gmrf_handle* h) { return h->bsp * h->bsp; }
static inline int64_t c_ld(const gmrf_handle* h) { return h->bsp - h->cmin; }                      // row stride of a stored C_i
static inline int64_t c_blk(const gmrf_handle* h) { return h->rmax * (h->bsp - h->cmin); }         // one stored C_i
static inline int64_t stride_pL(const gmrf_handle* h) { return blk_elems(h) * (h->keep_l ? h->N : 1); }
static inline int64_t stride_pX(const gmrf_handle* h) { return blk_elems(h) * h->N; }
static inline int64_t stride_pC(const gmrf_handle* h) { return c_blk(h) * std::max<int64_t>(h->N - 1, 1); }
static inline double* l_block(const gmrf_handle* h, int64_t i) { return h->d_L + (h->keep_l ? i * blk_elems(h) : 0); }

// Layout of the coupling blocks (see gmrf_handle): validates, derives mend / the accounting sums and
// uploads the per-tile bounds.  kst_abs[t]: first non-zero column of row tile t (absolute, any value;
// it is rounded down to 64 and replaced by its monotone envelope).  Needs set_shape first.
static gmrf_status set_layout(gmrf_handle* h, int64_t cmin, int64_t rmax, const std::vector<int64_t>& kst_abs) {
    const int64_t bsp = h->bsp;
    if (cmin < 0 || cmin >= bsp || cmin % 64 || rmax <= 0 || rmax > bsp || rmax % 64) return bad_shape("bad coupling-block layout");
    const int nrt = (int)(rmax / 64), nct = (int)((bsp - cmin) / 64), ntile = (int)(bsp / 64);
    if ((int)kst_abs.size() != nrt) return bad_shape("bad coupling-block layout (row tiles)");
    (void)hipStreamSynchronize(h->stream);
    destroy_graphs(h);
    h->cmin = cmin; h->rmax = rmax;
    h->kst.assign((size_t)ntile, 0);
    int64_t run = bsp - 64;                                   // envelope from the bottom: kst[t] = min over t' >= t
    for (int t = nrt - 1; t >= 0; --t) {
        int64_t v = std::min(std::max(kst_abs[t], cmin), bsp - 64);
        run = std::min(run, (v / 64) * 64);
        h->kst[t] = (int)(run - cmin);
    }
    if (nrt > 0) h->kst[0] = 0;                               // cmin is the smallest start by definition
    for (int t = nrt; t < ntile; ++t) h->kst[t] = (int)(bsp - cmin);      // rows below rmax: empty
    h->mend.assign((size_t)ntile, (int)rmax);
    for (int u = 0; u < nct; ++u) {
        int m = 0;
        for (int t = 0; t < nrt; ++t) if (h->kst[t] <= 64 * u) m = 64 * (t + 1);
        h->mend[u] = m;
    }
    h->c_streamed = 0.0; h->g2_tile_k = 0.0;
    for (int t = 0; t < nrt; ++t) {
        h->c_streamed += 64.0 * (double)(bsp - cmin - h->kst[t]);
        h->g2_tile_k += (double)(t + 1) * (double)(bsp - cmin - h->kst[t]);    // tiles (t, 0..t) start at kst[t]
    }
    if (!h->d_kst || !h->d_mend || h->kst_cap < ntile) {
        free_dev(h->d_kst); free_dev(h->d_mend); free_dev(h->d_kbx); h->d_kst = h->d_mend = h->d_kbx = nullptr; h->kst_cap = 0;
        HIPCHK(hipMalloc(&h->d_kst, sizeof(int) * ntile));
        HIPCHK(hipMalloc(&h->d_mend, sizeof(int) * ntile));
        HIPCHK(hipMalloc(&h->d_kbx, sizeof(int) * ntile));
        h->kst_cap = ntile;
    }
    {
        std::vector<int> kbx((size_t)ntile);
        for (int u = 0; u < ntile; ++u) kbx[(size_t)u] = (int)std::max<int64_t>(0, 64 * (int64_t)u - cmin);
        HIPCHK(hipMemcpyAsync(h->d_kbx, kbx.data(), sizeof(int) * ntile, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));        // (kbx is a local)
    }
    HIPCHK(hipMemcpyAsync(h->d_kst, h->kst.data(), sizeof(int) * ntile, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_mend, h->mend.data(), sizeof(int) * ntile, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->c_dirty = true;
    return GMRF_OK;
}

static gmrf_status set_layout_dense(gmrf_handle* h) {
    return set_layout(h, 0, h->bsp, std::vector<int64_t>((size_t)(h->bsp / 64), 0));
}

static gmrf_status alloc_work(gmrf_handle* h);
static gmrf_status alloc_factor(gmrf_handle* h) {
    if (h->alloc_N == h->N && h->alloc_bsp == h->bsp && h->alloc_B == h->B && h->alloc_rm == h->rmax &&
        h->alloc_wc == c_ld(h) && h->alloc_keep_l == h->keep_l && h->d_Linv)
        return GMRF_OK;
    destroy_graphs(h);
    h->logdet_valid = false;
    if (h->external_storage) {
        // caller-owned L / C / Linv (sized by gmrf_bt_storage_bytes for this shape and batch): only the
        // work buffers follow the layout
        if (h->alloc_N != h->N || h->alloc_bsp != h->bsp || h->alloc_B != h->B || h->alloc_keep_l != h->keep_l) {
            g_last_error = "external factor storage does not match the shape / batch";
            return GMRF_ERR_BAD_SHAPE;
        }
        h->alloc_rm = h->rmax; h->alloc_wc = c_ld(h);
        h->l_valid = false;
        return GMRF_OK;
    }
    free_dev(h->d_L); free_dev(h->d_C); free_dev(h->d_Linv);
    free_dev(h->d_S); free_dev(h->d_B); free_dev(h->d_T); free_dev(h->d_W);
    free_dev(h->d_logdet);
    h->d_L = h->d_C = h->d_Linv = h->d_S = h->d_B = h->d_T = h->d_W = h->d_logdet = nullptr;
    h->l_valid = false;
    const size_t blk = (size_t)blk_elems(h) * sizeof(double) * (size_t)h->B;
    HIPCHK(hipMalloc(&h->d_L, blk * (h->keep_l ? h->N : 1)));
    HIPCHK(hipMalloc(&h->d_Linv, blk * h->N));
    HIPCHK(hipMalloc(&h->d_C, (size_t)stride_pC(h) * sizeof(double) * (size_t)h->B));
    // tiles strictly above the block diagonal of L / Linv are never written: keep them zero
    HIPCHK(hipMemsetAsync(h->d_L, 0, blk * (h->keep_l ? h->N : 1), h->stream));
    HIPCHK(hipMemsetAsync(h->d_Linv, 0, blk * h->N, h->stream));
    HIPCHK(hipMemsetAsync(h->d_C, 0, (size_t)stride_pC(h) * sizeof(double) * (size_t)h->B, h->stream));
    h->c_dirty = false;
    GCHK(alloc_work(h));
    h->alloc_rm = h->rmax; h->alloc_wc = c_ld(h); h->alloc_keep_l = h->keep_l;
    return GMRF_OK;
}

static gmrf_status alloc_work(gmrf_handle* h) {
    const size_t blk = (size_t)blk_elems(h) * sizeof(double) * (size_t)h->B;
    HIPCHK(hipMalloc(&h->d_S, blk));
    HIPCHK(hipMalloc(&h->d_B, blk));
    HIPCHK(hipMalloc(&h->d_T, blk));
    HIPCHK(hipMalloc(&h->d_W, blk));
    HIPCHK(hipMalloc(&h->d_logdet, sizeof(double) * h->N * h->B));
    {   // flag words of the persistent in-block launches (potrf_persist.hpp), one set per problem
        const int64_t words = (int64_t)persist_flag_words((int)(h->bsp / 64)) * h->B;
        if (!h->d_pflags || h->pflags_words < words) {
            free_dev(h->d_pflags); h->d_pflags = nullptr; h->pflags_words = 0;
            HIPCHK(hipMalloc(&h->d_pflags, sizeof(unsigned) * (size_t)words));
            HIPCHK(hipMemsetAsync(h->d_pflags, 0, sizeof(unsigned) * (size_t)words, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));     // (whatever stream the first launch is on later sees the zeros)
            h->pflags_words = words;
        }
    }
    h->alloc_N = h->N; h->alloc_bsp = h->bsp; h->alloc_B = h->B;
    h->stats.factor_bytes = (int64_t)(blk * ((h->keep_l ? 2 : 1) * h->N) + (size_t)stride_pC(h) * sizeof(double) * (size_t)h->B);
    return GMRF_OK;
}

// Buffers sized by n / n_pad / the batch: dropped whenever the shape or the batch changes
// (they are re-created on demand with the new sizes).
static void release_shape_buffers(gmrf_handle* h) {
    for (auto& kv : h->sweep_graphs) (void)hipGraphExecDestroy(kv.second);
    h->sweep_graphs.clear();
    free_dev(h->d_P); free_dev(h->d_Y); free_dev(h->d_Tp);
    h->d_P = h->d_Y = h->d_Tp = nullptr; h->kp_cap = 0;
    free_dev(h->d_mean); h->d_mean = nullptr;
    free_dev(h->d_acc); h->d_acc = nullptr; h->acc_B = 0;
    free_dev(h->d_stage); h->d_stage = nullptr; h->stage_cap = 0;
}

static gmrf_status set_shape(gmrf_handle* h, int64_t n, int64_t N) {
    if (n <= 0 || N <= 0 || n % N != 0) return bad_shape("n must be a positive multiple of N_blocks");
    const int64_t bs = n / N;
    if (bs > (1 << 20)) return bad_shape("block size too large");
    if (h->n != n || h->N != N) {
        (void)hipStreamSynchronize(h->stream);
        destroy_graphs(h);
        release_shape_buffers(h);
        h->factored = false;
        h->analyzed = false;
    }
    h->n = n; h->N = N; h->bs = bs;
    h->bsp = 64 * next_pow2((bs + 63) / 64);
    h->n_pad = h->bsp * N;
    h->stats.n = n; h->stats.n_blocks = N; h->stats.block_size = bs; h->stats.block_size_padded = h->bsp;
    h->stats.factor_flops = ((double)N * bs * bs * bs / 3.0 + (double)(N - 1) * 2.0 * bs * bs * bs) * (double)h->B;
    return GMRF_OK;
}

static gmrf_status ensure_panels(gmrf_handle* h, int64_t kp) {
    if (kp <= h->kp_cap && h->d_P) return GMRF_OK;
    for (auto& kv : h->sweep_graphs) (void)hipGraphExecDestroy(kv.second);
    h->sweep_graphs.clear();
    free_dev(h->d_P); free_dev(h->d_Y); free_dev(h->d_Tp);
    h->d_P = h->d_Y = h->d_Tp = nullptr;
    HIPCHK(hipMalloc(&h->d_P, sizeof(double) * kp * h->n_pad * h->B));
    HIPCHK(hipMalloc(&h->d_Y, sizeof(double) * kp * h->n_pad * h->B));
    HIPCHK(hipMalloc(&h->d_Tp, sizeof(double) * kp * h->bsp * h->B));
    h->kp_cap = kp;
    return GMRF_OK;
}

static gmrf_status ensure_stage(gmrf_handle* h, int64_t elems) {
    if (elems <= h->stage_cap && h->d_stage) return GMRF_OK;
    free_dev(h->d_stage);
    h->d_stage = nullptr;
    HIPCHK(hipMalloc(&h->d_stage, sizeof(double) * elems));
    h->stage_cap = elems;
    return GMRF_OK;
}

// ------------------------------------------------------------------------------------ symbolic
struct HostEntry { uint64_t key; int64_t src; };

// Everything the symbolic phase derives on the HOST from the blocks' entry lists (no HIP call: the sanitizer build of the
// host side reaches it without a GPU through gmrf_test_symbolic_csc): the flat key / source arrays in block order (lower
// blocks row by row), the zero structure shared by all lower blocks (first non-zero column, last non-zero row, per 64-row
// tile the first non-zero column: the staircase), the row pointers into the lower blocks' entry lists, and the tile plan
// of the sparse C = B X^T (spmm_bxt_tiles: per lower block and 64-row tile the distinct columns, per entry its index).
struct SymbolicPlan {
    std::vector<int64_t> diag_first, diag_count, low_first, low_count;
    std::vector<uint64_t> keys;
    std::vector<int64_t> src;
    int64_t cmin = 0, rmax = 0;
    std::vector<int64_t> first;            // per 64-row tile of the window: first non-zero column (absolute)
    std::vector<int> rowptr;               // [N][bsp + 1]
    int64_t max_row = 0;
    bool sparse_b = false, bxt_ok = false;
    // tile plan of spmm_bxt_tiles, by GROUPS of up to three 64-row tiles of a lower block whose entries meet (mostly) the same
    // columns of X (round 4): per block the groups' tiles gtiles[N][nrt][3] (-1: none) and their count ng[N]; per group the
    // distinct columns (uptr[N][nrt + 1] into ucols), per entry its index into its group's list
    std::vector<int> uptr, ucols, gtiles, ng;
    std::vector<uint16_t> lidx;
    int ecap = 0, nrt = 0;
    std::vector<int> drowptr;          // [N][bsp + 1]: a diagonal block's entries row by row
};

static void build_symbolic(int64_t N, int64_t bsp, const std::vector<std::vector<HostEntry>>& dg,
                           const std::vector<std::vector<HostEntry>>& lo, bool no_staircase, SymbolicPlan& sp, bool group_tiles = true) {
    sp.diag_first.assign((size_t)N, 0); sp.diag_count.assign((size_t)N, 0);
    sp.low_first.assign((size_t)N, 0); sp.low_count.assign((size_t)N, 0);
    auto& keys = sp.keys; auto& src = sp.src;
    keys.clear(); src.clear();
    for (int64_t i = 0; i < N; ++i) {
        sp.diag_first[i] = (int64_t)keys.size(); sp.diag_count[i] = (int64_t)dg[i].size();
        {   // row by row too (round 5): the workgroups of spmm_bxt_tiles that zero rows of the next Schur block scatter D_i's rows into them
            std::vector<HostEntry> drow(dg[i]);
            std::stable_sort(drow.begin(), drow.end(), [](const HostEntry& x, const HostEntry& y) { return x.key < y.key; });
            for (auto& e : drow) { keys.push_back(e.key); src.push_back(e.src); }
        }
        sp.low_first[i] = (int64_t)keys.size(); sp.low_count[i] = (int64_t)lo[i].size();
        std::vector<HostEntry> byrow(lo[i]);           // row by row (key = row << 32 | col): spmm_bxt walks rows
        std::stable_sort(byrow.begin(), byrow.end(), [](const HostEntry& x, const HostEntry& y) { return x.key < y.key; });
        for (auto& e : byrow) { keys.push_back(e.key); src.push_back(e.src); }
    }
    // zero structure shared by all lower blocks (set_layout takes the monotone envelope of `first`)
    {
        int64_t cmin = bsp, rmax = 0;
        std::vector<int64_t> first((size_t)(bsp / 64), bsp);
        for (int64_t i = 1; i < N; ++i)
            for (auto& e : lo[i]) {
                const int64_t r = (int64_t)(e.key >> 32), c = (int64_t)(e.key & 0xffffffffu);
                cmin = std::min(cmin, c); rmax = std::max(rmax, r + 1);
                first[(size_t)(r / 64)] = std::min(first[(size_t)(r / 64)], c);
            }
        if (rmax == 0) { cmin = 0; rmax = 64; first[0] = 0; }
        cmin = (cmin / 64) * 64;
        rmax = std::min<int64_t>(bsp, (rmax + 63) / 64 * 64);
        first.resize((size_t)(rmax / 64));
        if (no_staircase) std::fill(first.begin(), first.end(), cmin);
        sp.cmin = cmin; sp.rmax = rmax; sp.first = first;
    }
    // row pointers into the lower blocks' entry lists for the sparse C = B X^T (spmm_bxt)
    sp.rowptr.assign((size_t)(N * (bsp + 1)), 0);
    int64_t max_row = 0;
    for (int64_t i = 1; i < N; ++i) {
        int* rp = sp.rowptr.data() + i * (bsp + 1);
        const int64_t first = sp.low_first[i], cnt = sp.low_count[i];
        for (int64_t k = 0; k < cnt; ++k) rp[(keys[first + k] >> 32) + 1]++;
        for (int64_t r = 0; r < bsp; ++r) max_row = std::max<int64_t>(max_row, rp[r + 1]);
        rp[0] = (int)first;
        for (int64_t r = 0; r < bsp; ++r) rp[r + 1] += rp[r];
    }
    // row pointers into the diagonal blocks' entry lists (absolute positions in `keys`)
    sp.drowptr.assign((size_t)(N * (bsp + 1)), 0);
    if (keys.size() < ((size_t)1 << 31)) {
        for (int64_t i = 0; i < N; ++i) {
            int* rp = sp.drowptr.data() + i * (bsp + 1);
            const int64_t first = sp.diag_first[i], cnt = sp.diag_count[i];
            for (int64_t k = 0; k < cnt; ++k) rp[(keys[first + k] >> 32) + 1]++;
            rp[0] = (int)first;
            for (int64_t r = 0; r < bsp; ++r) rp[r + 1] += rp[r];
        }
    }
    // dense GEMM: 2 bs^3 flop at ~50 TF/s; sparse: one pass over C.  Rows of up to 32 entries go the sparse way,
    // anything denser keeps the GEMM.
    sp.max_row = max_row;
    sp.sparse_b = keys.size() < ((size_t)1 << 31) && max_row <= 32 && bsp >= 64;
    // tile plan for spmm_bxt_tiles (rows of the lower blocks are sorted by (row, column): a 64-row tile is a contiguous
    // range of entries)
    sp.bxt_ok = false; sp.uptr.clear(); sp.ucols.clear(); sp.lidx.clear(); sp.gtiles.clear(); sp.ng.clear();
    if (sp.sparse_b && N > 1) {
        // Row tiles whose entries meet the same columns share ONE gathered chunk of X inside a workgroup (a stencil block: the
        // tiles of mesh rows 2 and 3 of a block meet subsets of the columns the tile of mesh row 1 above them meets -- darcy256:
        // 195 + 129 + 64 gathered columns become 195).  Greedy: a tile joins the group of an earlier tile when at least half of
        // its columns are already there and the union still fits one thread per column.
        static const bool no_groups = [] { const char* e = getenv("GMRF_BXT_GROUPS"); return e && atoi(e) == 0; }();   // tuning aid: every tile alone
        const int nrt = (int)(sp.rmax / 64);
        sp.uptr.assign((size_t)(N * (nrt + 1)), 0);
        sp.gtiles.assign((size_t)(N * nrt * 3), -1);
        sp.ng.assign((size_t)N, 0);
        sp.lidx.assign(std::max<size_t>(keys.size(), 1), 0);
        std::vector<std::vector<int>> cols((size_t)nrt);
        std::vector<int> uni, tmp;
        bool ok = true;
        int64_t pmax = 0;
        for (int64_t i = 1; i < N && ok; ++i) {
            const int* rp = sp.rowptr.data() + i * (bsp + 1);
            for (int rt = 0; rt < nrt; ++rt) {
                auto& c = cols[(size_t)rt];
                c.clear();
                for (int e = rp[rt * 64]; e < rp[rt * 64 + 64]; ++e) c.push_back((int)(keys[(size_t)e] & 0xffffffffu));
                std::sort(c.begin(), c.end());
                c.erase(std::unique(c.begin(), c.end()), c.end());
                if ((int)c.size() > BXT_UCAP) ok = false;
            }
            if (!ok) break;
            std::vector<char> taken((size_t)nrt, 0);
            int g = 0;
            for (int rt = 0; rt < nrt; ++rt) {
                if (taken[(size_t)rt]) continue;
                int* gt = sp.gtiles.data() + ((size_t)i * nrt + g) * 3;
                int cnt = 0;
                gt[cnt++] = rt; taken[(size_t)rt] = 1;
                uni = cols[(size_t)rt];
                for (int u = rt + 1; u < nrt && cnt < 3 && !no_groups && group_tiles; ++u) {
                    if (taken[(size_t)u] || cols[(size_t)u].empty()) continue;
                    tmp.clear();
                    std::set_union(uni.begin(), uni.end(), cols[(size_t)u].begin(), cols[(size_t)u].end(), std::back_inserter(tmp));
                    const size_t common = uni.size() + cols[(size_t)u].size() - tmp.size();
                    if (2 * common >= cols[(size_t)u].size() && tmp.size() <= (size_t)BXT_UCAP) {
                        gt[cnt++] = u; taken[(size_t)u] = 1;
                        uni.swap(tmp);
                    }
                }
                int64_t padded = 0;
                for (int q = 0; q < cnt; ++q) {
                    const int t = gt[q];
                    for (int e = rp[t * 64]; e < rp[t * 64 + 64]; ++e)
                        sp.lidx[(size_t)e] = (uint16_t)(std::lower_bound(uni.begin(), uni.end(), (int)(keys[(size_t)e] & 0xffffffffu)) - uni.begin());
                    for (int r = t * 64; r < t * 64 + 64; ++r) padded += (rp[r + 1] - rp[r] + 3) / 4 * 4;     // (the kernel pads a row to turns of 4 entries)
                }
                pmax = std::max(pmax, padded);
                sp.uptr[(size_t)(i * (nrt + 1) + g)] = (int)sp.ucols.size();
                sp.ucols.insert(sp.ucols.end(), uni.begin(), uni.end());
                ++g;
            }
            sp.uptr[(size_t)(i * (nrt + 1) + g)] = (int)sp.ucols.size();
            for (int q = g + 1; q <= nrt; ++q) sp.uptr[(size_t)(i * (nrt + 1) + q)] = (int)sp.ucols.size();
            sp.ng[(size_t)i] = g;
        }
        const int ecap = (int)std::max<int64_t>(64, (pmax + 63) / 64 * 64);
        if (ok && bxt_tile_lds_bytes(ecap) <= 53 * 1024) { sp.bxt_ok = true; sp.ecap = ecap; sp.nrt = nrt; }
    }
}

static gmrf_status upload_entries(gmrf_handle* h, const std::vector<std::vector<HostEntry>>& dg,
                                  const std::vector<std::vector<HostEntry>>& lo, int64_t nnz_in) {
    const int64_t N = h->N;
    SymbolicPlan sp;
    // (tile groups for batches: one problem alone has too few workgroups as it is -- 48 chunks x 12 tiles = 576 of 768 slots -- and
    //  a third of them doing three tiles' rows each is slower: 7.1 -> 9.7 us per block on darcy256)
    build_symbolic(N, h->bsp, dg, lo, h->no_staircase, sp, h->B >= 8);
    h->diag_first = sp.diag_first; h->diag_count = sp.diag_count; h->low_first = sp.low_first; h->low_count = sp.low_count;
    const auto& keys = sp.keys; const auto& src = sp.src;
    GCHK(set_layout(h, sp.cmin, sp.rmax, sp.first));
    free_dev(h->d_keys); free_dev(h->d_vals); free_dev(h->d_src); free_dev(h->d_nz_stage);
    h->d_keys = nullptr; h->d_vals = nullptr; h->d_src = nullptr; h->d_nz_stage = nullptr;
    h->n_entries = (int64_t)keys.size();
    h->nnz_in = nnz_in;
    const size_t ne = std::max<size_t>(keys.size(), 1);
    HIPCHK(hipMalloc(&h->d_keys, ne * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&h->d_vals, ne * sizeof(double) * h->B));
    HIPCHK(hipMalloc(&h->d_src, ne * sizeof(int64_t)));
    HIPCHK(hipMalloc(&h->d_nz_stage, std::max<int64_t>(nnz_in, 1) * sizeof(double) * h->B));
    h->vals_B = h->B;
    if (!keys.empty()) {
        // (stream-ordered copies + a stream wait: a synchronous hipMemcpy goes through the legacy stream,
        //  which HIP refuses while another host thread is capturing a graph)
        HIPCHK(hipMemcpyAsync(h->d_keys, keys.data(), keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_src, src.data(), src.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    free_dev(h->d_lo_rowptr);
    h->d_lo_rowptr = nullptr;
    HIPCHK(hipMalloc(&h->d_lo_rowptr, sp.rowptr.size() * sizeof(int)));
    HIPCHK(hipMemcpyAsync(h->d_lo_rowptr, sp.rowptr.data(), sp.rowptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    free_dev(h->d_dg_rowptr);
    h->d_dg_rowptr = nullptr;
    HIPCHK(hipMalloc(&h->d_dg_rowptr, sp.drowptr.size() * sizeof(int)));
    HIPCHK(hipMemcpyAsync(h->d_dg_rowptr, sp.drowptr.data(), sp.drowptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->lo_row_max = sp.max_row;
    h->sparse_b = sp.sparse_b;
    free_dev(h->d_bxt_uptr); free_dev(h->d_bxt_ucols); free_dev(h->d_bxt_lidx); free_dev(h->d_bxt_gtiles);
    h->d_bxt_uptr = nullptr; h->d_bxt_ucols = nullptr; h->d_bxt_lidx = nullptr; h->d_bxt_gtiles = nullptr; h->bxt_plan_ok = false;
    if (sp.bxt_ok) {
        HIPCHK(hipMalloc(&h->d_bxt_gtiles, sp.gtiles.size() * sizeof(int)));
        HIPCHK(hipMemcpyAsync(h->d_bxt_gtiles, sp.gtiles.data(), sp.gtiles.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        h->bxt_ng = sp.ng;
        HIPCHK(hipMalloc(&h->d_bxt_uptr, sp.uptr.size() * sizeof(int)));
        HIPCHK(hipMalloc(&h->d_bxt_ucols, std::max<size_t>(sp.ucols.size(), 1) * sizeof(int)));
        HIPCHK(hipMalloc(&h->d_bxt_lidx, sp.lidx.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpyAsync(h->d_bxt_uptr, sp.uptr.data(), sp.uptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (!sp.ucols.empty()) HIPCHK(hipMemcpyAsync(h->d_bxt_ucols, sp.ucols.data(), sp.ucols.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_bxt_lidx, sp.lidx.data(), sp.lidx.size() * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->bxt_ecap = sp.ecap; h->bxt_nrt = sp.nrt; h->bxt_plan_ok = true;
    }
    h->analyzed = true;
    h->factored = false;
    return GMRF_OK;
}

// The CSC walk of the symbolic phase (host only): which stored entries belong to which block; band check.
static gmrf_status split_csc(int64_t n, int64_t N, int64_t bs, const int64_t* colptr, const int64_t* rowval, int32_t base,
                             std::vector<std::vector<HostEntry>>& dg, std::vector<std::vector<HostEntry>>& lo) {
    dg.assign((size_t)N, {}); lo.assign((size_t)N, {});
    if (colptr[0] != base) return bad_shape("colptr must start at the index base");
    for (int64_t c = 0; c < n; ++c) {
        const int64_t bj = c / bs;
        if (colptr[c + 1] < colptr[c]) return bad_shape("colptr must be non-decreasing");
        for (int64_t p = colptr[c] - base; p < colptr[c + 1] - base; ++p) {
            const int64_t r = rowval[p] - base;
            if (r < 0 || r >= n) return bad_shape("row index out of range");
            const int64_t bi = r / bs;
            if (bi == bj) {
                if (r >= c) dg[bi].push_back({((uint64_t)(r - bi * bs) << 32) | (uint64_t)(c - bj * bs), p});
            } else if (bi == bj + 1) {
                lo[bi].push_back({((uint64_t)(r - bi * bs) << 32) | (uint64_t)(c - bj * bs), p});
            } else if (bi + 1 == bj) {
                // upper block: never read by the reference (src/tridiagonal_cholesky.jl:73,76)
            } else {
                g_last_error = "entry outside the block tri-band of the partition";
                return GMRF_ERR_BAND;
            }
        }
    }
    return GMRF_OK;
}

static gmrf_status analyze_csc(gmrf_handle* h, int64_t n, int64_t N, const int64_t* colptr,
                               const int64_t* rowval, int32_t base) {
    if (!colptr || !rowval) return bad_shape("null CSC arrays");
    GCHK(set_shape(h, n, N));
    std::vector<std::vector<HostEntry>> dg, lo;
    GCHK(split_csc(n, N, h->bs, colptr, rowval, base, dg, lo));
    return upload_entries(h, dg, lo, colptr[n] - base);
}

// blockIdx.y = problem: nzval of problem p starts at nz + p * nnz
__global__ void gather_values(const double* __restrict__ nz, const int64_t* __restrict__ src,
                              int64_t count, double* __restrict__ vals, int64_t nnz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) vals[(int64_t)blockIdx.y * count + i] = nz[(int64_t)blockIdx.y * nnz + src[i]];
}

static gmrf_status load_values(gmrf_handle* h, const double* nzval) {
    if (!nzval) return bad_shape("null nzval");
    const double* d_nz = nzval;
    if (h->vals_B != h->B) {            // batch size changed after the analysis: resize the value buffers
        free_dev(h->d_vals); free_dev(h->d_nz_stage);
        h->d_vals = nullptr; h->d_nz_stage = nullptr;
        HIPCHK(hipMalloc(&h->d_vals, std::max<int64_t>(h->n_entries, 1) * sizeof(double) * h->B));
        HIPCHK(hipMalloc(&h->d_nz_stage, std::max<int64_t>(h->nnz_in, 1) * sizeof(double) * h->B));
        h->vals_B = h->B;
    }
    if (!is_device_ptr(nzval)) {
        HIPCHK(hipMemcpyAsync(h->d_nz_stage, nzval, sizeof(double) * h->nnz_in * h->B, hipMemcpyHostToDevice, h->stream));
        d_nz = h->d_nz_stage;
    }
    if (h->n_entries > 0) {
        const int bl = 256;
        hipLaunchKernelGGL(gather_values, dim3((unsigned)((h->n_entries + bl - 1) / bl), (unsigned)h->B), dim3(bl), 0,
                           h->stream, d_nz, h->d_src, h->n_entries, h->d_vals, h->nnz_in);
        HIPCHK(hipGetLastError());
    }
    return GMRF_OK;
}

// ------------------------------------------------------------------------------------ numeric factor
// Recursive doubling  X21 = -X22 (L21 X11)  over pairs of hh-wide diagonal blocks, levels hh = lo .. hi.
// half: -1 all pairs, 0 / 1 only the pairs inside the first / second half of the block.
static gmrf_status doubling_levels(gmrf_handle* h, double* L, double* X, double* T, int lo, int hi, int half, int split = 0) {
    const int bsp = (int)h->bsp;
    const int64_t ld = bsp;
    const int64_t pL = stride_pL(h), pX = stride_pX(h), pW = (int64_t)bsp * bsp;
    if (split > 0 && half >= 0) return bad_shape("internal: the split inverse is not assembled half by half");
    for (int hh = lo; hh <= hi && hh < bsp; hh *= 2) {
        int pairs = bsp / (2 * hh), first = 0;
        if (half >= 0) { pairs /= 2; first = half * pairs; }
        if (split > 0 && hh >= split) {
            // split representation (split = p, a power of two): the first block column [0, p) of the rows >= p is not
            // assembled.  Level p: the first pair IS that part; above: the first pair keeps its columns [p, hh) --
            // T[:, p:hh] = L21[:, p:hh] X11[p:hh, p:hh],  X21[:, p:hh] = -X22 T[:, p:hh]
            if (hh > split) {
                const int q = split, w = hh - q;
                GCHK(gemm(h, false, true, hh, w, w, TRI_B_LOWER, 0, 1.0, L + (int64_t)hh * ld + q, ld, X + (int64_t)q * ld + q, ld, 0.0,
                          T + (int64_t)hh * ld + q, ld, pL, pX, pW));
                GCHK(gemm(h, false, true, hh, w, hh, TRI_A_LOWER, 0, -1.0, X + (int64_t)hh * ld + hh, ld, T + (int64_t)hh * ld + q, ld, 0.0,
                          X + (int64_t)hh * ld + q, ld, pX, pW, pX));
            }
            first = 1; pairs -= 1;
        }
        if (pairs <= 0) continue;
        const int64_t st = (int64_t)2 * hh * ld + 2 * hh, o = first * st;
        // T21 = L21 * X11
        GCHK(gemm(h, false, true, hh, hh, hh, TRI_B_LOWER, 0, 1.0, L + o + (int64_t)hh * ld, ld, X + o, ld, 0.0,
                  T + o + (int64_t)hh * ld, ld, pL, pX, pW, pairs, st, st, st));
        // X21 = -X22 * T21
        GCHK(gemm(h, false, true, hh, hh, hh, TRI_A_LOWER, 0, -1.0, X + o + (int64_t)hh * ld + hh, ld,
                  T + o + (int64_t)hh * ld, ld, 0.0, X + o + (int64_t)hh * ld, ld, pX, pW, pX, pairs, st, st, st));
    }
    return GMRF_OK;
}

// The top level (hh = bsp / 2, one pair) in its two halves: T21 = L21 X11, then X21 = -X22 T21.
static gmrf_status doubling_top(gmrf_handle* h, double* L, double* X, double* T, bool first_product, bool second_product) {
    const int bsp = (int)h->bsp, hh = bsp / 2;
    const int64_t ld = bsp;
    const int64_t pL = stride_pL(h), pX = stride_pX(h), pW = (int64_t)bsp * bsp;
    if (first_product)
        GCHK(gemm(h, false, true, hh, hh, hh, TRI_B_LOWER, 0, 1.0, L + (int64_t)hh * ld, ld, X, ld, 0.0, T + (int64_t)hh * ld, ld,
                  pL, pX, pW));
    if (second_product)
        GCHK(gemm(h, false, true, hh, hh, hh, TRI_A_LOWER, 0, -1.0, X + (int64_t)hh * ld + hh, ld, T + (int64_t)hh * ld, ld, 0.0,
                  X + (int64_t)hh * ld, ld, pX, pW, pX));
    return GMRF_OK;
}

static gmrf_status fork_event(gmrf_handle* h, hipEvent_t* out) {
    if (h->fork_next == h->fork_events.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->fork_events.push_back(e);
    }
    *out = h->fork_events[h->fork_next++];
    return GMRF_OK;
}

// Which in-block Cholesky the next factorisation of this handle takes: the ONE route predicate (potrf_block follows it,
// planned_xsplit asks it; round 5 -- the comparison routes that lost twice, left-looking panels, rank-64 panel steps of batches,
// 128-column panels and the one-workgroup potrf_panel256, are gone, and with them four flags both places had to agree on by hand).
enum PotrfRoute {
    ROUTE_FUSED = 1,       // one problem, blocks of up to 16 tiles: ONE persistent launch per block, else one launch per 64-column step
    ROUTE_BIG_ONE = 2,     // one problem, larger blocks: 256-column panels, each a persistent launch (or fused steps) + a rank-256 GEMM
    ROUTE_PANELS256 = 3,   // batches: 256-column panels -- potrf_diag128 x 2 + 128^3 GEMMs, or one persistent launch per diagonal block -- + GEMM
    ROUTE_STEPS = 4        // what is left (blocks of fewer than 4 tiles, tile counts not divisible by 4, the forked graph): tile, panel, update
};
static int potrf_route(const gmrf_handle* h) {
    const int nt = (int)(h->bsp / 64);
    if (h->B == 1 && !h->split_step) return nt <= 16 ? ROUTE_FUSED : ROUTE_BIG_ONE;
    if (h->fork_graph && nt >= 8) return ROUTE_STEPS;
    return (nt >= 4 && nt % 4 == 0) ? ROUTE_PANELS256 : ROUTE_STEPS;
}

// The split the next factorisation of this handle will use (0: none): only the 256-column panel route of batches
// assembles the inverse level by level, and only a coupling window that starts at cmin >= 256 leaves a first block
// column nobody multiplies with.  p = the largest power of two <= cmin (the doubling tree splits at powers of two).
static int planned_xsplit(const gmrf_handle* h) {
    const int bsp = (int)h->bsp, nt = bsp / 64;
    if (h->no_xsplit || h->N <= 0 || nt < 8 || potrf_route(h) != ROUTE_PANELS256) return 0;
    int p = 256;
    while (2 * p <= (int)h->cmin) p *= 2;
    return (h->cmin >= 256 && p < bsp) ? p : 0;
}

// ---- the device's budget of CUs for persistent launches (round 5) --------------------------------------------------------
// Every workgroup of a potrf_persist launch must be resident at once (140 KB of LDS: one per CU).  One handle alone checks its
// launch against the CU count; several handles' launches in flight together -- a StreamSet of batched handles, two one-problem
// handles on two threads -- can together ask for more than the chip has, end up partially resident and starve each other until
// their bounded waits give up (0.2 - 2 s each).  So a handle CLAIMS the CUs its widest persistent launch needs when it plans a
// factorisation, the claims of a device never exceed its CU count, and a handle whose claim does not fit takes the launch-per-
// step routes up front (stats: persist_refused).  A claim is held until the handle is destroyed, gives a launch up, or
// re-plans with another shape: the route of a handle does not depend on what other handles happen to be doing at the moment.
static std::mutex g_persist_mu;
static std::map<int, std::map<const void*, int>> g_persist_claims;     // device -> handle -> CUs

// `who` asks for `want` CUs of a device with `cus`: true and recorded when it fits beside the others' claims (want = 0 releases)
static bool persist_budget_claim(std::map<const void*, int>& claims, const void* who, int want, int cus, int margin) {
    int others = 0;
    for (const auto& kv : claims) if (kv.first != who) others += kv.second;
    if (want <= 0 || others + want + margin > cus) { claims.erase(who); return false; }
    claims[who] = want;
    return true;
}

static int potrf_route(const gmrf_handle* h);
static const int& persist_margin() {
    static const int margin = [] { const char* e = getenv("GMRF_PERSIST_CU_MARGIN"); return e ? atoi(e) : 0; }();   // tuning aid
    return margin;
}
// CUs the widest persistent launch of this handle's next factorisation needs at once (0: its route launches none)
static int persist_demand(const gmrf_handle* h) {
    if (h->no_persist || h->persist_gave_up || h->cu_count <= 0 || h->bsp < 128 || h->fork_graph) return 0;
    const int nt = (int)(h->bsp / 64);
    int64_t wgs = 0;
    switch (potrf_route(h)) {
        case 1 /* ROUTE_FUSED */: if (!h->no_lookahead && !h->doubling_x) wgs = 1 + persist_tiles(nt, 0, nt, 1); break;
        case 2 /* ROUTE_BIG_ONE */: if (!h->no_lookahead) wgs = 1 + persist_tiles(nt, 0, std::min(nt, 4), 0); break;      // (the first panel is the widest)
        case 3 /* ROUTE_PANELS256 */: {
            static const bool no_small = [] { const char* e = getenv("GMRF_PERSIST_PANELS"); return e && atoi(e) == 0; }();   // tuning aid
            if (!no_small && !h->no_persist_panels) wgs = 1 + persist_tiles(4, 0, 4, 2);
            break;
        }
        default: break;
    }
    wgs *= h->B;
    return wgs > 0 && wgs <= h->cu_count ? (int)wgs : 0;
}
// One problem with blocks of 512 .. 1024: its sweeps run as one persistent launch each (sweep_persist.hpp), whose workgroups --
// one per CU, up to 256 -- must all be resident: the handle then asks for the whole chip.  0: the sweeps keep a launch per product.
static int sweep_persist_demand(const gmrf_handle* h) {
    if (h->no_persist || h->no_sweep_persist || h->persist_gave_up || h->cu_count < 64 || h->B != 1) return 0;
    if (!h->d_kst || !h->d_mend) return 0;             // (a handle that adopted a factor without analysing a pattern)
    if (h->bsp < 512 || h->bsp > SWEEP_PERSIST_XMAX || h->bsp % 64 != 0 || h->cmin % 16 != 0 || h->rmax % 16 != 0) return 0;      // (the bodies' tilings: 16 rows, 8 / 16 columns)
    return h->cu_count;
}
// plan the persistent launches of the next factorisation: claim (or release) this handle's CUs
static void persist_plan(gmrf_handle* h) {
    const int want_f = persist_demand(h), want_s = sweep_persist_demand(h);
    int want = std::max(want_f, want_s);
    bool ok;
    {
        std::lock_guard<std::mutex> lock(g_persist_mu);
        ok = persist_budget_claim(g_persist_claims[h->device], h, want, h->cu_count, want == h->cu_count ? 0 : persist_margin());
        if (!ok && want_f > 0 && want_f < want) {          // (the whole chip is not free: the factorisation's launches alone)
            want = want_f;
            ok = persist_budget_claim(g_persist_claims[h->device], h, want, h->cu_count, persist_margin());
        }
    }
    h->sweep_persist_planned = ok && want_s > 0 && want >= want_s;
    const int got = ok ? want : 0;
    if (got != h->persist_cus) destroy_graphs(h);              // (a captured factor graph holds the launches of the old plan)
    h->persist_cus = got;
    h->stats.persist_cus = got;
    h->stats.persist_refused = (want > 0 && !ok) ? 1 : 0;
}
static void persist_release(gmrf_handle* h) {
    std::lock_guard<std::mutex> lock(g_persist_mu);
    auto it = g_persist_claims.find(h->device);
    if (it != g_persist_claims.end()) it->second.erase(h);
    h->persist_cus = 0;
    h->stats.persist_cus = 0;
    h->sweep_persist_planned = false;
}

// One persistent launch (potrf_persist.hpp) over the column tiles [j0, j1) of a block of nt tiles (its flag words are zero
// between launches: the last workgroup out of a launch leaves them so).  false: the handle holds no claim that covers this
// shape (it does not fit the chip beside the other handles' launches, or the form is switched off).
static bool persist_fits(const gmrf_handle* h, int nt, int j0, int j1, int xrows) {
    if (h->no_persist || h->persist_gave_up || h->persist_cus <= 0) return false;
    // (a batch: every problem brings its own set of workgroups and flag words; all of them must fit the claim at once)
    return (int64_t)(1 + persist_tiles(nt, j0, j1, xrows)) * h->B <= h->persist_cus;
}

static gmrf_status launch_persist(gmrf_handle* h, double* S, double* L, double* X, int nt, int j0, int j1, int xrows, int blk_id,
                                  double flops) {
    const int words = persist_flag_words(nt);
    if (!h->d_pflags || h->pflags_words < (int64_t)words * h->B) return bad_shape("internal: flag words of the persistent launches not allocated");
    // (no memset node: the words are zero -- zeroed when allocated, left zero by the last workgroup out of every launch)
    PersistArgs pa;
    pa.S = S; pa.L = L; pa.X = X; pa.ld = h->bsp; pa.nt = nt; pa.j0 = j0; pa.j1 = j1; pa.xrows = xrows;
    pa.tail_panel = (!xrows && j1 < nt) ? 1 : 0;
    pa.info = h->d_info; pa.blk = blk_id;
    pa.pS = h->bsp * h->bsp; pa.pL = stride_pL(h); pa.pX = stride_pX(h); pa.blk_per_problem = (int)h->N;
    pa.flags = h->d_pflags; pa.flag_stride = words;
    pa.abort_word = reinterpret_cast<unsigned*>(h->d_info + 1);
    // Bound of every wait, in ticks of the 100 MHz clock: 2 s for one problem; 200 ms for a batch -- several handles' launches
    // can ask for more workgroups than the chip has CUs (each must be resident as a whole), and if they ever hold each other's
    // CUs the handles should learn it soon, drain, and go on with the other route (a launch lasts 60 us; a wait that is
    // honest ends within a few GEMM launches of another stream).
    static const int limit_ms = [] { const char* e = getenv("GMRF_PERSIST_SPIN_MS"); return e ? atoi(e) : -1; }();
    pa.spin_limit = (unsigned)(limit_ms >= 0 ? limit_ms : (h->B > 1 ? 200 : 2000)) * 100000u;
    pa.stamps = h->dbg_stamps;
    h->persist_launched = true;
    h->stats.persist_route = xrows == 1 ? 1 : (xrows == 0 ? 2 : 3);
    ProfScope ps(h, 17, flops);                        // (its own class: the rocprof symbol is potrf_persist, not potrf_step)
    hipLaunchKernelGGL(potrf_persist<false>, dim3(1 + persist_tiles(nt, j0, j1, xrows), (unsigned)h->B), dim3(POTRF_PERSIST_THREADS), POTRF_PERSIST_LDS,
                       h->stream, pa);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

static gmrf_status potrf_block(gmrf_handle* h, double* S, double* L, double* X, double* T, int blk_id) {
    const int bsp = (int)h->bsp;
    const int64_t ld = bsp;
    const int nt = bsp / 64;
    // Batches: two-level blocking.  Inside a 256-column panel the 64-column steps update only the
    // panel's own columns; the rest of the trailing block gets ONE rank-256 update per panel from
    // the GEMM kernel (a quarter of the read-modify-write traffic of four rank-64 updates, and
    // one rounding at |S| instead of four).  A lone problem with blocks up to 1024 keeps the fused
    // one-launch step (43 ms against 48 ms on darcy256); beyond that the fused step's redundant
    // tile factorisations lose (bs = 4096: 4.45 s fused, 2.93 s two-level).
    const int route = potrf_route(h);
    const bool fused = route == ROUTE_FUSED;
    if (h->xsplit > 0 && route != ROUTE_PANELS256) return bad_shape("internal: split inverse planned off the 256-column panel route");
    static const int pw_env = [] { const char* e = getenv("GMRF_PANEL_TILES"); return e ? atoi(e) : 4; }();   // tuning aid
    const int pw = (!fused && nt >= 8) ? ((pw_env == 2 || pw_env == 8) ? pw_env : 4) : nt;           // panel width in tiles
    // a lone problem with larger blocks: two-level, with the fused kernel restricted to the panel's
    // own columns as the in-panel step (one launch instead of three where the panel has columns left)
    const bool fused_in_panel = route == ROUTE_BIG_ONE;
    // Opt-in (set_eager bit 4): inside a captured graph the block forks once its first half is
    // factored: a second branch assembles the inverse of that half and the top-level product
    // T21 = L21 X11 (everything they read is final) while this branch runs the latency-bound panel
    // chain of the second half.  Measured on darcy256: the branches do overlap (3.4 ms of a 42 ms
    // factor) but the two cross-queue dependencies per block cost as much, and with several handles
    // the extra streams share hardware queues (3 x 32: 22.6 k -> 19.3 k solves/s).  Off by default.
    const bool overlap = h->capturing && h->fork_graph && nt >= 8;
    if (overlap && !h->aux) HIPCHK(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (overlap) { GCHK(fork_event(h, &ev_fork)); GCHK(fork_event(h, &ev_join)); }
    if (fused && !h->no_lookahead && !h->doubling_x && !overlap && nt >= 2 && persist_fits(h, nt, 0, nt, 1)) {
        // One problem, ONE launch per block: the look-ahead chain below with flags in place of its launch boundaries
        // (potrf_persist.hpp)
        const double t3 = 64.0 * 64.0 * 64.0;
        double fl = t3 / 3.0 * nt;
        for (int j = 0; j + 1 < nt; ++j) {
            const int m = nt - j - 1;
            fl += (double)m * t3 + 2.0 * t3 * (m * (m + 1) / 2) + 2.0 * t3 * (0.5 * (double)(j + 1) * (j + 2) + 0.625 * (j + 1));
        }
        return launch_persist(h, S, L, X, nt, 0, nt, 1, blk_id, fl);
    }
    if (fused && !h->no_lookahead && !h->doubling_x && !overlap && nt >= 2) {
        // One problem, look-ahead chain (potrf_step, `lookahead`): tile 0 alone, then per step j ONE launch in which
        // workgroup 0 forms L[j+1,j], updates tile (j+1,j+1) in LDS and factors it for the next launch while the other
        // workgroups do panel + update of their tiles with the X_jj the previous launch left; the tiles of row j of
        // the inverse ride along (xrow_strip), its last row gets the launch at the end.
        StepArgs sa;
        sa.S = S; sa.L = L; sa.X = X; sa.ld = ld; sa.nt = nt; sa.cend = nt;
        sa.info = h->d_info; sa.blk = blk_id; sa.dbg = nullptr;
        sa.pS = (int64_t)bsp * bsp; sa.pL = stride_pL(h); sa.pX = stride_pX(h); sa.blk_per_problem = (int)h->N;
        const double t3 = 64.0 * 64.0 * 64.0;
        {
            sa.j = 0; sa.lookahead = 0; sa.xrow = -1;
            ProfScope ps(h, 1, t3 / 3.0);
            hipLaunchKernelGGL(potrf_step<false>, dim3(1, 1), dim3(256), POTRF_TILE_LDS, h->stream, sa);
        }
        for (int j = 0; j + 1 < nt; ++j) {
            const int m = nt - j - 1, ntile_wg = m * (m + 1) / 2;
            sa.j = j; sa.lookahead = 1;
            sa.xrow = (j >= 1) ? j : -1; sa.xrow_first = ntile_wg;
            const int nx_wg = (j >= 1) ? 4 * j : 0;
            const double f_x = (j >= 1) ? 2.0 * t3 * (0.5 * (double)j * (j + 1) + 0.625 * j) : 0.0;
            ProfScope ps(h, 1, t3 / 3.0 + (double)m * t3 + 2.0 * t3 * ntile_wg + f_x);
            hipLaunchKernelGGL(potrf_step<false>, dim3(ntile_wg + nx_wg, 1), dim3(256), POTRF_STEP_LDS, h->stream, sa);
        }
        {
            sa.j = nt - 1; sa.lookahead = 0; sa.xrow = nt - 1; sa.xrow_first = 0;
            ProfScope ps(h, 1, 2.0 * t3 * (0.5 * (double)(nt - 1) * nt + 0.625 * (nt - 1)));
            hipLaunchKernelGGL(potrf_step<false>, dim3(4 * (nt - 1), 1), dim3(256), 64 * 18 * sizeof(double), h->stream, sa);
        }
        HIPCHK(hipGetLastError());
        return GMRF_OK;
    }
    // Batches (round 3): 256-column panels whose diagonal block is two potrf_diag128 launches (see potrf_step.hpp); the
    // rows below a panel meet the 256 x 256 inverse of its diagonal block in ONE product on the GEMM kernel (K = 256):
    //   potrf_diag128(A)                              A, B: the panel's two 128 x 128 diagonal blocks
    //   L_BA = S_BA X_A^T,  S_BB -= L_BA L_BA^T       (two 128^3 products)
    //   potrf_diag128(B)
    //   X_BA = -X_B (L_BA X_A)                        (the level-128 doubling step of this pair: X_P = [X_A 0; X_BA X_B])
    //   L[below, P] = S[below, P] X_P^T               (K = 256, X_P lower triangular)
    //   S[below, below] -= L[below, P] L[below, P]^T  (rank-256 update)
    if (route == ROUTE_PANELS256 && !overlap) {
        StepArgs sa;
        sa.S = S; sa.L = L; sa.X = X; sa.ld = ld; sa.nt = nt; sa.cend = nt;
        sa.info = h->d_info; sa.blk = blk_id; sa.dbg = nullptr;
        sa.pS = (int64_t)bsp * bsp; sa.pL = stride_pL(h); sa.pX = stride_pX(h); sa.blk_per_problem = (int)h->N;
        const int64_t pW = (int64_t)bsp * bsp;
        const double t3 = 64.0 * 64.0 * 64.0, nb = (double)h->B;
        auto diag128 = [&](int j) -> gmrf_status {
            sa.j = j;
            // two tile Choleskys + four triangular 64^3 products (L10, S11 update, L10 X00, X11 W)
            ProfScope ps(h, 16, (2.0 * t3 / 3.0 + 4.0 * 2.0 * t3 * 0.625) * nb);
            static const size_t lds_pad = [] { const char* e = getenv("GMRF_DIAG128_LDS_PAD_KB"); return (size_t)(e ? atoi(e) : 0) * 1024; }();   // tuning aid
            static const bool fat = [] { const char* e = getenv("GMRF_DIAG128_FAT"); return e && atoi(e) != 0; }();      // tuning aid: the three-tile form
            if (fat) hipLaunchKernelGGL(potrf_diag128, dim3(1, (unsigned)h->B), dim3(256), POTRF_DIAG128_LDS + lds_pad, h->stream, sa);
            else hipLaunchKernelGGL(potrf_diag128_slim, dim3(1, (unsigned)h->B), dim3(256), POTRF_DIAG128_SLIM_LDS + lds_pad, h->stream, sa);
            HIPCHK(hipGetLastError());
            return GMRF_OK;
        };
        // Small batches (round 4): 7 workgroups per problem fit the chip up to a batch of 36, and then the panel's whole
        // 256 x 256 diagonal block -- two potrf_diag128 launches and the four 128^3 products -- is ONE persistent launch
        // (potrf_persist on the 4 x 4 tiles of the block, inverse rows included): 62 us instead of ~100 us per panel, on
        // 7 B CUs instead of B (C4 elliptic512 at batch 8: 8 of 256 CUs were busy in the diagonal chain).  Larger batches keep
        // the one-workgroup kernels: their chain hides behind the other problems, and 7 B workgroups of 140 KB would not fit.
        static const bool no_small = [] { const char* e = getenv("GMRF_PERSIST_PANELS"); return e && atoi(e) == 0; }();   // tuning aid
        const bool persist_panels = !no_small && !h->no_persist_panels && persist_fits(h, 4, 0, 4, 2);
        for (int j = 0; j < nt; j += 4) {
            const int64_t oa = (int64_t)j * 64, ob = oa + 128, oc = oa + 256;
            if (persist_panels) {
                const double fl = (4.0 * t3 / 3.0 + 2.0 * t3 * (6.0 * 0.625 + 10.0 + 10.0 + 6.0 * 0.625)) * nb;
                GCHK(launch_persist(h, S + oa * ld + oa, L + oa * ld + oa, X + oa * ld + oa, 4, 0, 4, 2, blk_id, fl));
            } else {
                GCHK(diag128(j));
                // The four 128^3 products of the panel -- L_BA = S_BA X_A^T, S_BB -= L_BA L_BA^T before the second diagonal block,
                // X_BA = -X_B (L_BA X_A) after it -- are four GEMM launches of 256 workgroups (13 % of a step's GEMM time at 20.7 TF/s).
                // (Round 4 built and measured the alternative, ONE workgroup per problem in two launches, bitwise the same results:
                //  the time-weighted GEMM fraction rose to 0.66 and the job got SLOWER, 45.6 k against 48.1 k solves/s -- twelve
                //  dependent 64^3 products take one CU 57 us, the four launches 39 us.  Removed in round 5.)
                // L_BA = S_BA X_A^T (X_A lower triangular, stored [n][k]);  S_BB -= L_BA L_BA^T (lower tiles)
                GCHK(gemm(h, false, false, 128, 128, 128, TRI_B_UPPER, 0, 1.0, S + ob * ld + oa, ld, X + oa * ld + oa, ld, 0.0, L + ob * ld + oa, ld,
                          sa.pS, sa.pX, sa.pL, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * t3 * 3.0 * 2.0 * nb));
                GCHK(gemm(h, false, false, 128, 128, 128, 0, 1, -1.0, L + ob * ld + oa, ld, L + ob * ld + oa, ld, 1.0, S + ob * ld + ob, ld,
                          sa.pL, sa.pL, sa.pS, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * t3 * 2.0 * 3.0 * nb));
                GCHK(diag128(j + 2));
                // X_BA = -X_B (L_BA X_A): the level-128 doubling step of this pair (T is the work block doubling_levels uses)
                GCHK(gemm(h, false, true, 128, 128, 128, TRI_B_LOWER, 0, 1.0, L + ob * ld + oa, ld, X + oa * ld + oa, ld, 0.0, T + ob * ld + oa, ld,
                          sa.pL, sa.pX, pW));
                GCHK(gemm(h, false, true, 128, 128, 128, TRI_A_LOWER, 0, -1.0, X + ob * ld + ob, ld, T + ob * ld + oa, ld, 0.0, X + ob * ld + oa, ld,
                          sa.pX, pW, sa.pX));
            }
            const int m3 = nt - j - 4;                             // row tiles below the panel
            if (m3 <= 0) continue;
            // Split representation with p = the panel width and no L blocks kept: L[p:, 0:p] is read by this panel's own
            // update only, so it is formed directly in the place it takes in Linv's storage (see gmrf_handle::xsplit)
            const bool in_slot = (j == 0 && h->xsplit == 256 && !h->keep_l);
            double* Lb = in_slot ? X + oc * ld + oa : L + oc * ld + oa;
            const int64_t pLb = in_slot ? sa.pX : sa.pL;
            // L[below, P] = S[below, P] X_P^T: b(k, n) = X_P[n][k], zero for k > n (tile-K units: 1 + 2 + 3 + 4 of 16)
            GCHK(gemm(h, false, false, 64 * m3, 256, 256, TRI_B_UPPER, 0, 1.0, S + oc * ld + oa, ld, X + oa * ld + oa, ld, 0.0, Lb, ld,
                      sa.pS, sa.pX, pLb, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * t3 * 10.0 * m3 * nb));
            // S[r,c] -= L[r,P] L[c,P]^T for the tiles right of / below the panel
            GCHK(gemm(h, false, false, m3 * 64, m3 * 64, 256, 0, 1, -1.0, Lb, ld, Lb, ld, 1.0, S + oc * ld + oc, ld, pLb, pLb, sa.pS,
                      1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * t3 * 4.0 * (m3 * (m3 + 1) / 2) * nb));
        }
        // X = L^-1 by recursive doubling over the 256-wide diagonal inverses -- or, split representation, everything
        // but its first block column below row p, whose place L[p:, 0:p] takes (see gmrf_handle::xsplit)
        const int p = h->xsplit;
        GCHK(doubling_levels(h, L, X, T, 256, bsp / 2, -1, p));
        if (p > 0 && !(p == 256 && !h->keep_l)) {
            hipLaunchKernelGGL(copy_rect, dim3((unsigned)((bsp - p) / 4), (unsigned)h->B), dim3(256), 0, h->stream,
                               L + (int64_t)p * ld, ld, sa.pL, X + (int64_t)p * ld, ld, sa.pX, bsp - p, p);
            HIPCHK(hipGetLastError());
        }
        return GMRF_OK;
    }
    // (only the 256-column panel route above leaves the split form: planned_xsplit asks the same predicate)
    if (h->xsplit > 0) return bad_shape("internal: split inverse planned off the 256-column panel route");
    for (int j = 0; j < nt; ++j) {
        StepArgs sa;
        sa.S = S; sa.L = L; sa.X = X; sa.ld = ld; sa.j = j; sa.nt = nt;
        sa.info = h->d_info; sa.blk = blk_id; sa.dbg = h->dbg_stamps;
        sa.pS = (int64_t)bsp * bsp; sa.pL = stride_pL(h); sa.pX = stride_pX(h); sa.blk_per_problem = (int)h->N;
        const int m = nt - j - 1;
        const int cend = std::min(nt, (j / pw + 1) * pw);  // first column tile outside this panel
        sa.cend = cend;
        int utiles = 0;
        for (int c = j + 1; c < cend; ++c) utiles += nt - c;
        const double rem = 64.0 * m, nb = (double)h->B;
        const double f_tile = 64.0 * 64.0 * 64.0 / 3.0 * nb, f_panel = rem * 64.0 * 64.0 * nb;
        const double f_upd = 2.0 * 64.0 * 64.0 * 64.0 * utiles * nb;
        if (fused_in_panel && !h->no_lookahead && !overlap && j % pw == 0 && cend - j >= 2 && persist_fits(h, nt, j, cend, 0)) {
            // One problem, larger blocks: the panel's column tiles [j, cend) as ONE persistent launch (potrf_persist.hpp): the
            // look-ahead chain over the panel's own columns, the last column's rows below included (the launch's last-column
            // workgroups form them: `tail_panel`); the rank-256 update of the rest of the block (GEMM) follows as before
            const int npc = cend - j;
            double fl = 64.0 * 64.0 * 64.0 / 3.0 * npc;
            for (int jj = j; jj + 1 < cend; ++jj) {
                int ut = 0;
                for (int c = jj + 1; c < cend; ++c) ut += nt - c;
                fl += 64.0 * (nt - jj - 1) * 64.0 * 64.0 + 2.0 * 64.0 * 64.0 * 64.0 * ut;
            }
            fl += 64.0 * (nt - cend) * 64.0 * 64.0;          // (the last column's rows below the panel: formed inside the launch too)
            GCHK(launch_persist(h, S, L, X, nt, j, cend, 0, blk_id, fl));
            j = cend - 1;                                     // the loop continues behind the panel's last column
            sa.j = j;
        } else if (fused || m == 0) {
            // one problem, fused steps: the tiles of row j - 1 of the inverse ride in the launch of step j (4 workgroups
            // per tile on CUs the step leaves idle; see xrow_strip), the last row gets a launch of its own below
            const bool xrows = fused && !h->doubling_x && !overlap && j >= 2;
            const int ntile_wg = 1 + m * (m + 1) / 2;
            sa.xrow = xrows ? j - 1 : -1; sa.xrow_first = ntile_wg;
            const int nx_wg = xrows ? 4 * (j - 1) : 0;
            const double f_x = xrows ? 2.0 * 64.0 * 64.0 * 64.0 * (0.5 * (double)(j - 1) * j + 0.625 * (j - 1)) : 0.0;
            ProfScope ps(h, 1, f_tile + f_panel + f_upd + f_x);
            hipLaunchKernelGGL(potrf_step<false>, dim3(ntile_wg + nx_wg, (unsigned)h->B), dim3(256),
                               m == 0 ? POTRF_TILE_LDS : POTRF_STEP_LDS, h->stream, sa);
            sa.xrow = -1;
        } else if (fused_in_panel && utiles > 0) {
            // the workgroups of column j+1 write the whole panel L[j+1.., j]
            ProfScope ps(h, 1, f_tile + f_panel + f_upd);
            hipLaunchKernelGGL(potrf_step<false>, dim3(1 + utiles, 1), dim3(256), POTRF_STEP_LDS, h->stream, sa);
        } else {
            // factor the B diagonal tiles once, then panel and update without the redundant tile work
            {
                ProfScope ps(h, 1, f_tile);
                hipLaunchKernelGGL(potrf_step<false>, dim3(1, (unsigned)h->B), dim3(256), POTRF_TILE_LDS, h->stream, sa);
            }
            {
                ProfScope ps(h, 8, f_panel);
                hipLaunchKernelGGL(potrf_panel, dim3(m, (unsigned)h->B), dim3(256), 0, h->stream, sa);
            }
            if (utiles > 0) {
                ProfScope ps(h, 9, f_upd);
                hipLaunchKernelGGL(potrf_update, dim3(utiles, (unsigned)h->B), dim3(256), 0, h->stream, sa);
            }
        }
        HIPCHK(hipGetLastError());
        if (j + 1 == cend && cend < nt) {
            // S[r,c] -= L[r,P] L[c,P]^T for the tiles right of / below the finished panel P
            const int j0 = cend - pw, mr = nt - cend;
            const double* Lp = L + (int64_t)cend * 64 * ld + (int64_t)j0 * 64;
            double* Sr = S + (int64_t)cend * 64 * ld + (int64_t)cend * 64;
            GCHK(gemm(h, false, false, mr * 64, mr * 64, pw * 64, 0, 1, -1.0, Lp, ld, Lp, ld, 1.0, Sr, ld, sa.pL, sa.pL,
                      sa.pS, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * 64.0 * 64.0 * (pw * 64.0) * (mr * (mr + 1) / 2) * (double)h->B));
        }
        if (overlap && j + 1 == nt / 2) {
            HIPCHK(hipEventRecord(ev_fork, h->stream));
            HIPCHK(hipStreamWaitEvent(h->aux, ev_fork, 0));
            h->gemm_stream = h->aux;
            gmrf_status st = doubling_levels(h, L, X, T, 64, bsp / 4, 0);
            if (st == GMRF_OK) st = doubling_top(h, L, X, T, true, false);
            h->gemm_stream = nullptr;
            GCHK(st);
        }
    }
    if (fused && !h->doubling_x && !overlap) {
        // the inverse was assembled row by row inside the step launches; its last row is what is left
        if (nt >= 2) {
            StepArgs sa;
            sa.S = S; sa.L = L; sa.X = X; sa.ld = ld; sa.j = nt - 1; sa.nt = nt;
            sa.info = h->d_info; sa.blk = blk_id; sa.dbg = nullptr;
            sa.pS = (int64_t)bsp * bsp; sa.pL = stride_pL(h); sa.pX = stride_pX(h); sa.blk_per_problem = (int)h->N;
            sa.cend = nt; sa.xrow = nt - 1; sa.xrow_first = 0;
            ProfScope ps(h, 1, 2.0 * 64.0 * 64.0 * 64.0 * (0.5 * (double)(nt - 1) * nt + 0.625 * (nt - 1)));
            hipLaunchKernelGGL(potrf_step<false>, dim3(4 * (nt - 1), 1), dim3(256), 64 * 18 * sizeof(double), h->stream, sa);
            HIPCHK(hipGetLastError());
        }
        return GMRF_OK;
    }
    // X = L^-1 by recursive doubling over the 64-wide diagonal inverses (the part not done beside
    // the panel chain above)
    if (overlap) {
        GCHK(doubling_levels(h, L, X, T, 64, bsp / 4, 1));       // second half, levels below the top
        HIPCHK(hipEventRecord(ev_join, h->aux));
        HIPCHK(hipStreamWaitEvent(h->stream, ev_join, 0));       // T21 of the top level is ready
        GCHK(doubling_top(h, L, X, T, false, true));
    } else {
        GCHK(doubling_levels(h, L, X, T, 64, bsp / 2, -1));
    }
    return GMRF_OK;
}

static gmrf_status factor_blocks_range(gmrf_handle* h, int64_t i0, int64_t i1) {
    const int bsp = (int)h->bsp;
    h->xsplit = planned_xsplit(h);                     // representation of the inverses this factorisation leaves
    const int64_t ld = bsp;
    const size_t blk_bytes = (size_t)bsp * bsp * sizeof(double) * (size_t)h->B;
    const int64_t bstride = (int64_t)bsp * bsp;
    const int64_t pX = stride_pX(h), pC = stride_pC(h);
    const int64_t ldc = c_ld(h), cstride = c_blk(h);
    const unsigned nb = (unsigned)h->B;
    for (int64_t i = i0; i < i1; ++i) {
        double* L = l_block(h, i);
        double* X = h->d_Linv + i * bstride;
        // S = D_i - C C^T is built as  S := -C C^T (GEMM, beta = 0)  then  S += D_i (scatter): same
        // single rounding as D - acc, and only the rows the product does not write (>= rmax) need
        // zeroing -- a quarter of the block on darcy instead of all of it.  The first block has no product.
        const int rm_s = (i > 0) ? (int)h->rmax : 0;
        // (round 5: where the coupling product of this block runs on spmm_bxt_tiles, that launch zeroes the rows as well)
        static const bool no_tiles_z = [] { const char* e = getenv("GMRF_BXT_TILES"); return e && atoi(e) == 0; }();
        const bool scatter_in_bxt = i > 0 && nb == 1 && h->sparse_b && !h->dense_g1 && h->bxt_plan_ok && !no_tiles_z && h->bxt_nrt == (int)h->rmax / 64
                                    && h->d_dg_rowptr && h->bsp == h->bs && !h->no_scatter_fold;
        const bool zero_in_bxt = scatter_in_bxt || (i > 0 && h->sparse_b && !h->dense_g1 && h->bxt_plan_ok && !no_tiles_z && h->bxt_nrt == (int)h->rmax / 64
                                                    && ((int64_t)(bsp - rm_s) * bsp) % 2 == 0);
        if (rm_s < bsp && !zero_in_bxt) {
            const int64_t cnt = (int64_t)(bsp - rm_s) * bsp;
            hipLaunchKernelGGL(zero_rows, dim3((unsigned)((cnt / 2 + 255) / 256), nb), dim3(256), 0, h->stream,
                               h->d_S + (int64_t)rm_s * ld, cnt, bstride);
            HIPCHK(hipGetLastError());
        }
        if (i > 0) {
            double* C = h->d_C + (i - 1) * cstride;              // stored window: rows 0 .. rm, columns cm ..
            const double* Xp = h->d_Linv + (i - 1) * bstride;
            const bool sparse_g1 = h->sparse_b && !h->dense_g1;
            if (!sparse_g1) {          // dense image of B for the GEMM route
                HIPCHK(hipMemsetAsync(h->d_B, 0, blk_bytes, h->stream));
                if (h->low_count[i] > 0) {
                    hipLaunchKernelGGL(scatter_block, dim3((unsigned)((h->low_count[i] + 255) / 256), nb), dim3(256), 0,
                                       h->stream, h->d_keys, h->d_vals, h->low_first[i], h->low_count[i], h->d_B, ld,
                                       h->n_entries, bstride, 0);
                    HIPCHK(hipGetLastError());
                }
            }
            // C = B * Linv_{i-1}^T      (src/tridiagonal_cholesky.jl:74).  B is zero left of column
            // cmin and below row rmax, hence C is zero there too and only the rest is computed; inside that
            // window row tile t of B -- and of C, Linv being lower triangular -- is zero left of kst[t].
            const int cm = (int)h->cmin, rm = (int)h->rmax;
            const int W = bsp - cm;
            if (sparse_g1) {
                BxtArgs ba;
                ba.rowptr = h->d_lo_rowptr + i * (h->bsp + 1); ba.keys = h->d_keys;
                ba.vals = h->d_vals; ba.n_entries = h->n_entries;
                ba.X = Xp; ba.C = C; ba.ld = ld; ba.ldc = ldc; ba.pX = pX; ba.pC = pC; ba.cm = cm; ba.rm = rm;
                ba.kst = h->d_kst;
                const int rch = (rm + 255) / 256;
                ba.cw = ((int64_t)(W / 64) * rch * nb >= 512) ? 64 : (((int64_t)(W / 32) * rch * nb >= 512) ? 32 : 16);
                ba.bsp = bsp;
                const dim3 grid((unsigned)((W / ba.cw) * rch * (int)nb));
                // streamed bytes: C inside the staircase (written) + the rows of Linv from the first gathered column on (read)
                ProfScope ps(h, 10, 8.0 * (h->c_streamed + 0.5 * (double)W * W) * (double)h->B);
                static const bool no_tiles = [] { const char* e = getenv("GMRF_BXT_TILES"); return e && atoi(e) == 0; }();   // tuning aid
                if (h->bxt_plan_ok && !no_tiles && h->bxt_nrt == rm / 64) {
                    BxtTileArgs ta;
                    ta.rowptr = ba.rowptr; ta.lidx = h->d_bxt_lidx; ta.vals = h->d_vals; ta.n_entries = h->n_entries;
                    ta.uptr = h->d_bxt_uptr + i * (h->bxt_nrt + 1); ta.ucols = h->d_bxt_ucols;
                    ta.gtiles = h->d_bxt_gtiles + i * h->bxt_nrt * 3; ta.ng = h->bxt_ng[(size_t)i];
                    ta.X = Xp; ta.C = C; ta.ld = ld; ta.ldc = ldc; ta.pX = pX; ta.pC = pC; ta.kst = h->d_kst;
                    ta.cm = cm; ta.rm = rm; ta.bsp = bsp; ta.ecap = h->bxt_ecap;
                    static const bool live_skip = [] { const char* e = getenv("GMRF_BXT_LIVE"); return e && atoi(e) != 0; }();   // tuning aid (measured: 4.1 vs 3.9 ms per step, off)
                    ta.skip_dead = live_skip ? 1 : 0;
                    const int chunks = W / 16, ng = ta.ng;
                    // 16-column chunks per workgroup: the launch should be a whole number of rounds over the 3 workgroups a CU
                    // holds (measured, darcy256 x 64, 4 groups x 48 chunks: 16 chunks = 768 workgroups = one round 85 us; 8 =
                    // two rounds 96 us; 12 = 1.33 rounds 108 us; 24 = 2/3 round 99 us): rounds x (chunks + ~2 for the prologue)
                    static const int nch_env = [] { const char* e = getenv("GMRF_BXT_NCH"); return e ? atoi(e) : 0; }();   // tuning aid
                    {
                        const int64_t slots = 3 * (int64_t)std::max(h->cu_count, 1);
                        int best = 1; int64_t best_cost = INT64_MAX;
                        for (int c : {1, 2, 3, 4, 6, 8, 12, 16, 24, 48}) {
                            if (c > chunks) break;
                            const int64_t wgs = (int64_t)((chunks + c - 1) / c) * ng * nb;
                            const int64_t cost = ((wgs + slots - 1) / slots) * (c + 2);
                            if (cost < best_cost || (cost == best_cost && c < best)) { best = c; best_cost = cost; }
                        }
                        ta.nch = nch_env > 0 ? std::min(nch_env, chunks) : best;
                    }
                    const int ncg = (chunks + ta.nch - 1) / ta.nch;
                    ta.nprob = (int)nb;
                    ta.zdst = nullptr; ta.zcount = 0; ta.zpdst = 0; ta.zwgs = 0;
                    ta.srowptr = nullptr; ta.skeys = nullptr; ta.svals = nullptr;
                    if (scatter_in_bxt) {
                        // one problem: the launch zeroes the WHOLE Schur block and scatters D_i into it row range by row range; the
                        // product then accumulates, S = 1.0 * S - C C^T -- the same single rounding fl(D - acc) as "S := -C C^T, then
                        // S += D_i", without the scatter launch (4.9 us + a boundary) between the product and the block's Cholesky
                        ta.zdst = h->d_S; ta.zcount = (int64_t)bsp * bsp; ta.zpdst = bstride; ta.zwgs = 64;
                        ta.srowptr = h->d_dg_rowptr + i * (h->bsp + 1); ta.skeys = h->d_keys; ta.svals = h->d_vals;
                    } else if (zero_in_bxt && rm_s < bsp) {
                        ta.zdst = h->d_S + (int64_t)rm_s * ld; ta.zcount = (int64_t)(bsp - rm_s) * bsp; ta.zpdst = bstride;
                        ta.zwgs = (int)std::min<int64_t>(32, std::max<int64_t>(1, ta.zcount / 16384));      // >= 128 KB per workgroup
                    }
                    hipLaunchKernelGGL(spmm_bxt_tiles, dim3((unsigned)((ncg * ng + ta.zwgs) * (int)nb)), dim3(256), bxt_tile_lds_bytes(h->bxt_ecap),
                                       h->stream, ta);
                }
                else if (h->lo_row_max <= 8) hipLaunchKernelGGL(spmm_bxt<8>, grid, dim3(256), 0, h->stream, ba);
                else if (h->lo_row_max <= 16) hipLaunchKernelGGL(spmm_bxt<16>, grid, dim3(256), 0, h->stream, ba);
                else hipLaunchKernelGGL(spmm_bxt<32>, grid, dim3(256), 0, h->stream, ba);
                HIPCHK(hipGetLastError());
            } else {
                GCHK(gemm(h, false, false, rm, W, W, TRI_B_UPPER, 0, 1.0, h->d_B + cm, ld,
                          Xp + (int64_t)cm * ld + cm, ld, 0.0, C, ldc, bstride, pX, pC, 1, 0, 0, 0, nullptr, 0, 0, 0, -1.0,
                          h->d_kst, nullptr, nullptr));
            }
            // S = D - C C^T             (src/tridiagonal_cholesky.jl:77): the product part.  Tile (R, R') sums over
            // the columns from max(kst[R], kst[R']) on -- the rest of the two row tiles is structurally zero.
            GCHK(gemm(h, false, false, rm, rm, W, 0, 1, -1.0, C, ldc, C, ldc, scatter_in_bxt ? 1.0 : 0.0, h->d_S, ld, pC, pC,
                      bstride, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * 64.0 * 64.0 * h->g2_tile_k * (double)h->B,
                      h->d_kst, h->d_kst, nullptr));
        }
        if (h->diag_count[i] > 0 && !scatter_in_bxt) {
            hipLaunchKernelGGL(scatter_block, dim3((unsigned)((h->diag_count[i] + 255) / 256), nb), dim3(256), 0,
                               h->stream, h->d_keys, h->d_vals, h->diag_first[i], h->diag_count[i], h->d_S, ld,
                               h->n_entries, bstride, 1);
            HIPCHK(hipGetLastError());
        }
        if (h->bsp > h->bs) {
            hipLaunchKernelGGL(pad_identity, dim3((unsigned)((h->bsp - h->bs + 255) / 256), nb), dim3(256), 0,
                               h->stream, h->d_S, ld, (int)h->bs, bsp, bstride);
            HIPCHK(hipGetLastError());
        }
        GCHK(potrf_block(h, h->d_S, L, X, h->d_T, (int)(i + 1)));
        if (!h->keep_l) {
            // L_i lives in a work buffer that the next block overwrites: its log-determinant part is taken now
            hipLaunchKernelGGL(logdet_blocks, dim3(1, nb), dim3(256), 0, h->stream, L, bstride, ld, (int)h->bs,
                               h->d_logdet + i, stride_pL(h), h->N);
            HIPCHK(hipGetLastError());
        }
    }
    return GMRF_OK;
}

// Split representation -> full inverse, in place, every block and problem: X_ba = -X_bb (L_ba X_aa) takes L_ba's place.
// Needed by whoever reads Linv_i as a matrix (selected inversion, gmrf_bt_get_block / export); the sweeps and the
// coupling products never are.  The factor stays full until the next factorisation.
static gmrf_status ensure_full_inverse(gmrf_handle* h) {
    if (h->xsplit <= 0) return GMRF_OK;
    const int bsp = (int)h->bsp, p = h->xsplit, q = bsp - p;
    const int64_t ld = bsp, bstride = (int64_t)bsp * bsp, pX = stride_pX(h), pW = bstride;
    for (int64_t i = 0; i < h->N; ++i) {
        double* X = h->d_Linv + i * bstride;
        double* T = h->d_T + (int64_t)p * ld;
        GCHK(gemm(h, false, true, q, p, p, TRI_B_LOWER, 0, 1.0, X + (int64_t)p * ld, ld, X, ld, 0.0, T, ld, pX, pX, pW));
        GCHK(gemm(h, false, true, q, p, q, TRI_A_LOWER, 0, -1.0, X + (int64_t)p * ld + p, ld, T, ld, 0.0, X + (int64_t)p * ld, ld, pX, pW, pX));
    }
    h->xsplit = 0;
    return GMRF_OK;
}

// Stream captures of different handles (host threads) are taken one at a time: capture + instantiate is a
// once-per-shape set-up step, and concurrent captures are the one place where handles would meet inside the runtime.
// A PRECAUTION, not a diagnosed fix: no fault of the runtime under concurrent thread-local captures was ever reproduced or
// traced to a cause here (round 2 saw one unexplained profiler-side crash and kept no log of it); the mutex costs nothing
// in steady state (graphs are replayed, not re-captured) and can go once concurrent captures have been shown to be safe.
static std::mutex g_capture_mu;

static gmrf_status run_factor(gmrf_handle* h, int64_t i0, int64_t i1) {
    if (h->eager || h->profiling) { h->stats.persist_route = 0; return factor_blocks_range(h, i0, i1); }
    if (!h->factor_graph || h->factor_graph_i0 != i0 || h->factor_graph_i1 != i1) {
        std::lock_guard<std::mutex> capture_lock(g_capture_mu);
        if (h->factor_graph) { (void)hipGraphExecDestroy(h->factor_graph); h->factor_graph = nullptr; }
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        h->capturing = true; h->fork_next = 0;
        h->stats.persist_route = 0;
        gmrf_status s = factor_blocks_range(h, i0, i1);
        h->capturing = false;
        h->factor_graph_route = h->stats.persist_route;
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (s != GMRF_OK) { if (graph) (void)hipGraphDestroy(graph); return s; }
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(&h->factor_graph, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        h->factor_graph_i0 = i0; h->factor_graph_i1 = i1;
    }
    h->xsplit = planned_xsplit(h);                     // (what the captured launches produce)
    h->stats.persist_route = h->factor_graph_route;
    if (h->factor_graph_route) h->persist_launched = true;
    HIPCHK(hipGraphLaunch(h->factor_graph, h->stream));
    return GMRF_OK;
}

// The blocks [i0, i1) have been enqueued; if a persistent launch was among them, wait for the range and look at the abort word:
// a bounded wait inside such a launch gave up (its workgroups were not all resident, or starved) -- every such launch has drained
// and left garbage.  The range is then repeated at once with the launch-per-step forms (its inputs, the value lists and the
// previous block's inverse, are untouched), which the handle keeps for good; the event is counted in stats.persist_aborts.
// Called per range by gmrf_bt_factor_step_async, BEFORE the caller packs / shares the range (a range shared before
// gmrf_bt_factor_end looked at the word carried garbage to the other ranks), and for the whole chain by factor_finish.
static gmrf_status persist_check_range(gmrf_handle* h, int64_t i0, int64_t i1, int* hinfo_out) {
    int hinfo2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(hinfo2, h->d_info, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->persist_launched = false;
    if (hinfo2[1] != 0) {
        h->persist_gave_up = true; h->persist_aborts++;
        h->stats.persist_aborts = h->persist_aborts;
        persist_release(h);
        destroy_graphs(h);
        const int restore[4] = {h->info_checked, 0, 0, 0};     // (what the aborted range wrote into the info word means nothing)
        HIPCHK(hipMemcpyAsync(h->d_info, restore, 4 * sizeof(int), hipMemcpyHostToDevice, h->stream));
        if (h->d_pflags) HIPCHK(hipMemsetAsync(h->d_pflags, 0, sizeof(unsigned) * (size_t)h->pflags_words, h->stream));   // (belt and braces: the drained launches cleaned up themselves)
        h->stats.persist_route = 0;
        GCHK(factor_blocks_range(h, i0, i1));
        HIPCHK(hipMemcpyAsync(hinfo2, h->d_info, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->persist_launched = false;
    }
    h->info_checked = hinfo2[0];
    if (hinfo_out) *hinfo_out = hinfo2[0];
    return GMRF_OK;
}

static gmrf_status factor_finish(gmrf_handle* h, int32_t* info) {
    // (stepwise factorisations have looked at every range that held a persistent launch already; the monolithic one is one range)
    int hinfo = 0;
    GCHK(persist_check_range(h, 0, h->N, &hinfo));
    if (info) *info = hinfo;
    if (hinfo != 0) {
        h->factored = false;
        g_last_error = "matrix is not positive definite; failed block " + std::to_string((hinfo - 1) % h->N + 1) +
                       " of problem " + std::to_string((hinfo - 1) / h->N);
        if (info) *info = (hinfo - 1) % (int)h->N + 1;
        return GMRF_ERR_NOT_SPD;
    }
    h->factored = true;
    h->l_valid = h->keep_l;
    h->logdet_valid = !h->keep_l;      // without the L blocks the factorisation left every block's part in d_logdet
    return GMRF_OK;
}

static gmrf_status numeric_factor(gmrf_handle* h, const double* nzval, int32_t* info) {
    if (!h->analyzed) { g_last_error = "no sparsity pattern analysed"; return GMRF_ERR_NO_FACTOR; }
    GCHK(alloc_factor(h));
    if (h->c_dirty) {
        HIPCHK(hipMemsetAsync(h->d_C, 0, sizeof(double) * stride_pC(h) * h->B, h->stream));
        h->c_dirty = false;
    }
    persist_plan(h);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    GCHK(load_values(h, nzval));
    HIPCHK(hipMemsetAsync(h->d_info, 0, 4 * sizeof(int), h->stream));
    h->info_checked = 0; h->persist_launched = false;
    GCHK(run_factor(h, 0, h->N));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    gmrf_status s = factor_finish(h, info);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    h->stats.factor_ms = ms;
    if (h->profiling) prof_collect(h);
    return s;
}

// ------------------------------------------------------------------------------------ streams on distinct hardware queues
// spins for `ticks` of the constant 100 MHz clock (always terminates: the clock advances)
__global__ void gmrf_spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
}

static double spin_pair_ms(hipStream_t a, hipStream_t b, unsigned long long ticks) {
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(gmrf_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    if (b) hipLaunchKernelGGL(gmrf_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
    (void)hipStreamSynchronize(a);
    if (b) (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

gmrf_status gmrf_streams_create(int32_t device, int32_t n, void** streams, int32_t* n_distinct) {
    if (n <= 0 || n > 32 || !streams) return bad_shape("gmrf_streams_create: 1 <= n <= 32 streams");
    HIPCHK(hipSetDevice(device));
    const int ncand = std::min(3 * n + 4, 48);
    std::vector<hipStream_t> cand((size_t)ncand, nullptr);
    for (int i = 0; i < ncand; ++i) {
        if (hipStreamCreateWithFlags(&cand[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipStreamDestroy(cand[(size_t)j]);
            g_last_error = "hipStreamCreateWithFlags failed"; return GMRF_ERR_HIP;
        }
        // first use binds the stream to its hardware queue
        hipLaunchKernelGGL(gmrf_spin_kernel, dim3(1), dim3(64), 0, cand[(size_t)i], 1ull);
    }
    HIPCHK(hipDeviceSynchronize());
    const unsigned long long ticks = 100000ull;                       // 1 ms at 100 MHz
    double one = 1e30;
    for (int r = 0; r < 3; ++r) one = std::min(one, spin_pair_ms(cand[0], nullptr, ticks));
    std::vector<int> chosen;
    for (int i = 0; i < ncand && (int)chosen.size() < n; ++i) {
        bool ok = true;
        for (int c : chosen) {
            // serialised pairs take 2 x one, overlapped ones ~1 x one (best of two tries against host jitter)
            const double t = std::min(spin_pair_ms(cand[(size_t)c], cand[(size_t)i], ticks), spin_pair_ms(cand[(size_t)c], cand[(size_t)i], ticks));
            if (t > 1.5 * one) { ok = false; break; }
        }
        if (ok) chosen.push_back(i);
    }
    const int nd = (int)chosen.size();
    std::vector<char> used((size_t)ncand, 0);
    for (int c : chosen) used[(size_t)c] = 1;
    for (int i = 0; i < ncand && (int)chosen.size() < n; ++i)         // not enough distinct queues: fill up with the others
        if (!used[(size_t)i]) { chosen.push_back(i); used[(size_t)i] = 1; }
    for (int k = 0; k < n; ++k) streams[k] = (void*)cand[(size_t)chosen[(size_t)k]];
    for (int i = 0; i < ncand; ++i)
        if (!used[(size_t)i]) (void)hipStreamDestroy(cand[(size_t)i]);
    if (n_distinct) *n_distinct = nd;
    return GMRF_OK;
}

gmrf_status gmrf_streams_destroy(int32_t device, int32_t n, void** streams) {
    if (n < 0 || (n > 0 && !streams)) return bad_shape("gmrf_streams_destroy");
    HIPCHK(hipSetDevice(device));
    for (int k = 0; k < n; ++k)
        if (streams[k]) { (void)hipStreamSynchronize((hipStream_t)streams[k]); (void)hipStreamDestroy((hipStream_t)streams[k]); streams[k] = nullptr; }
    return GMRF_OK;
}

// ------------------------------------------------------------------------------------ sweeps
static double sweep_bytes(const gmrf_handle* h, int64_t k) {
    const double bs = (double)h->bs, N = (double)h->N;
    return 8.0 * (N * bs * (bs + 1) / 2.0 + (N - 1) * bs * bs) + 16.0 * (double)h->n * (double)k;
}

// One problem: may this sweep run as one persistent launch?  (the claim is planned with the factorisation: persist_plan)
static bool sweep_persist_ok(const gmrf_handle* h, int kp) {
    if (!h->sweep_persist_planned || h->sweep_persist_hold || h->no_sweep_persist || h->no_persist || h->persist_gave_up || h->B != 1 || h->xsplit != 0) return false;
    if (h->persist_cus < h->cu_count || sweep_persist_demand(h) <= 0) return false;
    if (kp != 1 && (kp % 16 != 0 || (int64_t)kp * h->n_pad * 8 >= ((int64_t)1 << 31))) return false;
    if (h->n_pad * 8 >= ((int64_t)1 << 31)) return false;
    const bool via_gemm = (kp % 64 == 0) && !h->sweep_no_gemm && (int64_t)(h->bsp / 64) * (kp / 64) >= 128;
    return !via_gemm;
}

// the words and the intermediate panel of the persistent sweeps (before any capture: allocations are not allowed inside one)
static gmrf_status sweep_persist_prepare(gmrf_handle* h, int kp) {
    if (!h->d_sweep_flags) {
        HIPCHK(hipMalloc(&h->d_sweep_flags, sizeof(unsigned) * 16));
        HIPCHK(hipMemsetAsync(h->d_sweep_flags, 0, sizeof(unsigned) * 16, h->stream));
    }
    h->sweep_nw = std::min(256, h->cu_count);
    if (!h->h_sweep_abort) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_sweep_abort), 64, hipHostMallocMapped));
        *h->h_sweep_abort = 0u;
    }
    const int64_t need = (int64_t)kp * h->n_pad;
    if (!h->d_Tsw || h->t_elems < need) {
        HIPCHK(hipStreamSynchronize(h->stream));
        free_dev(h->d_Tsw); h->d_Tsw = nullptr; h->t_elems = 0;
        destroy_graphs(h);                                  // (captured sweeps hold the old pointer)
        HIPCHK(hipMalloc(&h->d_Tsw, sizeof(double) * (size_t)need));
        h->t_elems = need;
    }
    return GMRF_OK;
}

static gmrf_status launch_sweep_persist(gmrf_handle* h, bool backward, int kp, double* Pin, double* Yout, hipStream_t st_over = nullptr,
                                        double* T_over = nullptr) {
    const int nw = h->sweep_nw;
    const int64_t elems = (int64_t)kp * h->n_pad;
    hipStream_t st = st_over ? st_over : h->stream;
    double* Tp = T_over ? T_over : h->d_Tsw;
    if (!h->d_sweep_flags || !h->h_sweep_abort || nw <= 0 || !Tp || (!T_over && h->t_elems < elems))
        return bad_shape("internal: words / panel of the persistent sweeps not allocated");
    SweepPersistArgs a;
    a.C = h->d_C; a.Linv = h->d_Linv; a.Pin = Pin; a.T = Tp; a.Yout = Yout;
    a.N = (int)h->N; a.bsp = (int)h->bsp; a.cm = (int)h->cmin; a.rm = (int)h->rmax; a.kp = kp; a.backward = backward ? 1 : 0; a.nw = nw;
    a.npad = h->n_pad; a.ldc = c_ld(h); a.cstride = c_blk(h); a.bstride = (int64_t)h->bsp * h->bsp;
    a.kst = h->d_kst; a.mend = h->d_mend;
    a.abort_w = h->d_sweep_flags;
    void* dev_abort = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dev_abort, h->h_sweep_abort, 0));
    a.host_abort = reinterpret_cast<unsigned*>(dev_abort);
    static const int limit_ms = [] {
        const char* e = getenv("GMRF_SWEEP_SPIN_MS");              // (tests: the sweeps' own bound, so that the factorisation's launches pass)
        if (!e) e = getenv("GMRF_PERSIST_SPIN_MS");
        return e ? atoi(e) : -1;
    }();
    a.spin_limit = (unsigned)(limit_ms >= 0 ? limit_ms : 2000) * 100000u;
    static const int dbg = [] { const char* e = getenv("GMRF_SWEEP_DBG"); return e ? atoi(e) : 0; }();      // tuning aid
    a.dbg = dbg;
    // Pause between two looks at an input that has not arrived: k = 1 looks again at once (8 units = 0.2 us; a longer pause only
    // adds to every product: +0.28 us per 0.43 us measured).  The k >= 16 bodies look with 4 waves x 16 KB per workgroup, and
    // their looks load the fabric the OTHER workgroups' stores and looks travel on: 64 units (1.7 us) -- darcy256's 64-sample
    // sweep 0.96 -> 0.91 ms, and beside the mean's two sweeps (gmrf_bt_posterior) 1.52 -> 1.39 ms for the three; 128 units: 1.05 / 1.35.
    static const int pause_env = [] { const char* e = getenv("GMRF_SWEEP_PAUSE"); return e ? atoi(e) : -1; }();      // tuning aid
    a.pause = (kp == 1) ? 8 : (pause_env >= 0 ? pause_env : 64);
    const double N = (double)h->N, bsp = (double)h->bsp;
    const double work = (kp == 1) ? 8.0 * ((N - 1) * h->c_streamed + N * 0.5 * bsp * (bsp + 1))
                                  : kp * (2.0 * (N - 1) * h->c_streamed + N * bsp * (bsp + 1));
    {
        // the panels the products read from each other: sentinel everywhere (elems is even: n_pad is a multiple of 64)
        ProfScope ps(h, 5, 16.0 * (double)elems);
        hipLaunchKernelGGL(sweep_fill_sentinel, dim3((unsigned)((elems / 2 + 255) / 256)), dim3(256), 0, st, Tp, Yout, elems);
        HIPCHK(hipGetLastError());
    }
    ProfScope ps(h, kp == 1 ? 19 : 20, work);
    if (kp == 1) hipLaunchKernelGGL(sweep_persist<true>, dim3((unsigned)nw), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sweep_persist<false>, dim3((unsigned)nw), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

// behind a synchronisation of the stream: did a persistent sweep give up?  Then its panels hold garbage: the words are zeroed,
// the handle keeps the launch-per-product form for good (stats.persist_aborts), and the caller repeats its call.
static gmrf_status sweep_persist_check(gmrf_handle* h, bool* repeat) {
    *repeat = false;
    if (!h->sweep_persist_launched) return GMRF_OK;
    h->sweep_persist_launched = false;
    if (!h->h_sweep_abort || *reinterpret_cast<volatile unsigned*>(h->h_sweep_abort) == 0u) return GMRF_OK;
    *h->h_sweep_abort = 0u;
    h->persist_gave_up = true; h->persist_aborts++;
    h->stats.persist_aborts = h->persist_aborts;
    persist_release(h);
    destroy_graphs(h);
    HIPCHK(hipMemsetAsync(h->d_sweep_flags, 0, sizeof(unsigned) * 16, h->stream));
    h->stats.sweep_persist = 0;
    *repeat = true;
    return GMRF_OK;
}

static gmrf_status sweep_launches(gmrf_handle* h, bool backward, int kp, double* Pin, double* Yout) {
    if (sweep_persist_ok(h, kp)) return launch_sweep_persist(h, backward, kp, Pin, Yout);
    const int bsp = (int)h->bsp;
    const int64_t ld = bsp, bstride = (int64_t)bsp * bsp, npad = h->n_pad;
    const int64_t N = h->N;
    SweepArgs s;
    s.ld = ld;
    const int nprob = (int)h->B;
    s.narrow = (nprob < 8 && bsp >= 512 && bsp <= SWEEP_PERSIST_XMAX) ? 1 : 0;      // (what sweep_persist's k = 1 bodies sum: 8-column blocks)
    const int64_t pPanel = (int64_t)kp * npad;
    const int64_t pX = stride_pX(h), pCm = stride_pC(h);
    const int64_t ldc = c_ld(h), cstride = c_blk(h);
    // C_i is stored as its non-zero window (rows 0 .. rm, columns cm ..); inside it row tile t is zero
    // left of kst[t] (staircase).  forward: rows m < rm take sums over k >= cm + kst; backward (C^T):
    // outputs k >= cm take sums over the rows m < mend.
    const int cm = (int)h->cmin, rm = (int)h->rmax, wc = bsp - cm;
    // class 3 (k = 1): bytes the kernels stream (C inside the staircase, the lower triangle of Linv);
    // class 2: flops of the panel product
    const int pclass = (kp == 1) ? 3 : 2;
    const double blk_work_c = ((kp == 1) ? 8.0 * h->c_streamed : 2.0 * h->c_streamed * kp) * nprob;
    const double blk_work_t = ((kp == 1) ? 4.0 * bsp * (double)(bsp + 1) : 1.0 * bsp * (double)(bsp + 1) * kp) * nprob;
    // 64-multiples of right-hand sides go through the GEMM kernel (panel = the [m][k] operand) when
    // the batch gives it enough 64 x 64 tiles; a lone problem stays on sweep_mm (256 workgroups)
    const bool via_gemm = (kp % 64 == 0) && !h->sweep_no_gemm && (int64_t)nprob * (bsp / 64) * (kp / 64) >= 128;
    for (int64_t step = 0; step < N; ++step) {
        const int64_t i = backward ? (N - 1 - step) : step;
        double* rhs = Pin + i * bsp;               // the input panel is consumed: P_i becomes P_i - C y_prev
        if (step > 0) {
            // forward: P_i -= C_{i-1} y_{i-1};  backward: P_i -= C_i^T x_{i+1}      (in place)
            const int64_t ci = backward ? i : (i - 1);
            const int64_t prev = backward ? (i + 1) : (i - 1);
            const double* Cs = h->d_C + ci * cstride;
            const double* xin = Yout + prev * bsp + (backward ? 0 : cm);
            double* out = rhs + (backward ? cm : 0);
            const int rows = backward ? wc : rm, kdim = backward ? rm : wc;
            if (via_gemm) {
                // T[r][m] = P[r][m] - sum_k y[r][k] c(k,m): the panel is the [m][k] operand, the block the other
                GCHK(gemm(h, false, backward, kp, rows, kdim, 0, 0, -1.0, xin, npad, Cs, ldc, 1.0, out, npad, pPanel, pCm,
                          pPanel, 1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * h->c_streamed * kp * nprob,
                          nullptr, backward ? nullptr : h->d_kst, backward ? h->d_mend : nullptr));
            } else {
                s.Mat = Cs; s.ld = ldc; s.Xin = xin; s.ldx = npad; s.Bin = out; s.ldb = npad; s.Out = out; s.ldo = npad;
                s.rows = rows; s.kdim = kdim; s.sub = 1;
                s.pMat = pCm; s.pXin = pPanel; s.pBin = pPanel; s.pOut = pPanel;
                s.kst = h->d_kst; s.mend = h->d_mend;
                ProfScope ps(h, pclass, blk_work_c);
                HIPCHK(launch_sweep(h->stream, backward, false, kp, s, nprob));
            }
        }
        // y_i = Linv_i T   /   x_i = Linv_i^T T
        const double* X = h->d_Linv + i * bstride;
        double* yout = Yout + i * bsp;
        if (h->xsplit > 0) {
            // split representation: [X_aa 0; L_ba X_bb] in the block's storage.  forward: y_a = X_aa t_a, t_b -= L_ba y_a,
            // y_b = X_bb t_b;  backward: x_b = X_bb^T t_b, t_a -= L_ba^T x_b, x_a = X_aa^T t_a.  Same bytes, same flops.
            const int p = h->xsplit, q = bsp - p;
            const double* Xbb = X + (int64_t)p * ld + p;
            const double* Lba = X + (int64_t)p * ld;
            for (int part = 0; part < 3; ++part) {
                const int which = backward ? 2 - part : part;          // 0: aa, 1: ba, 2: bb
                if (via_gemm) {
                    if (which == 0)
                        GCHK(gemm(h, false, backward, kp, p, p, backward ? TRI_B_LOWER : TRI_B_UPPER, 0, 1.0, rhs, npad, X, ld, 0.0, yout, npad,
                                  pPanel, pX, pPanel));
                    else if (which == 2)
                        GCHK(gemm(h, false, backward, kp, q, q, backward ? TRI_B_LOWER : TRI_B_UPPER, 0, 1.0, rhs + p, npad, Xbb, ld, 0.0,
                                  yout + p, npad, pPanel, pX, pPanel));
                    else if (!backward)
                        GCHK(gemm(h, false, false, kp, q, p, 0, 0, -1.0, yout, npad, Lba, ld, 1.0, rhs + p, npad, pPanel, pX, pPanel));
                    else
                        GCHK(gemm(h, false, true, kp, p, q, 0, 0, -1.0, yout + p, npad, Lba, ld, 1.0, rhs, npad, pPanel, pX, pPanel));
                } else {
                    s.ld = ld; s.ldx = npad; s.ldo = npad; s.pMat = pX; s.pXin = pPanel; s.pOut = pPanel; s.kst = nullptr; s.mend = nullptr;
                    if (which == 0) { s.Mat = X; s.Xin = rhs; s.Bin = nullptr; s.ldb = 0; s.Out = yout; s.rows = p; s.kdim = p; s.sub = 0; s.pBin = 0; }
                    else if (which == 2) { s.Mat = Xbb; s.Xin = rhs + p; s.Bin = nullptr; s.ldb = 0; s.Out = yout + p; s.rows = q; s.kdim = q; s.sub = 0; s.pBin = 0; }
                    else if (!backward) { s.Mat = Lba; s.Xin = yout; s.Bin = rhs + p; s.ldb = npad; s.Out = rhs + p; s.rows = q; s.kdim = p; s.sub = 1; s.pBin = pPanel; }
                    else { s.Mat = Lba; s.Xin = yout + p; s.Bin = rhs; s.ldb = npad; s.Out = rhs; s.rows = p; s.kdim = q; s.sub = 1; s.pBin = pPanel; }
                    const double frac = which == 0 ? (double)p * (p + 1) : (which == 2 ? (double)q * (q + 1) : 2.0 * p * q);
                    ProfScope ps(h, pclass, blk_work_t * frac / ((double)bsp * (bsp + 1)));
                    HIPCHK(launch_sweep(h->stream, backward, which != 1, kp, s, nprob));
                }
            }
        } else if (via_gemm) {
            // forward: Linv stored [m][k], zero for k > m; backward: Linv^T, stored [k][m], zero for k < m
            GCHK(gemm(h, false, backward, kp, bsp, bsp, backward ? TRI_B_LOWER : TRI_B_UPPER, 0, 1.0, rhs, npad, X, ld, 0.0,
                      yout, npad, pPanel, pX, pPanel, 1, 0, 0, 0, nullptr, 0, 0, 0, -1.0));
        } else {
            s.Mat = X; s.ld = ld; s.Xin = rhs; s.ldx = npad; s.Bin = nullptr; s.ldb = 0; s.Out = yout; s.ldo = npad;
            s.rows = bsp; s.kdim = bsp; s.sub = 0;
            s.pMat = pX; s.pXin = pPanel; s.pBin = 0; s.pOut = pPanel;
            s.kst = nullptr; s.mend = nullptr;
            ProfScope ps(h, pclass, blk_work_t);
            HIPCHK(launch_sweep(h->stream, backward, true, kp, s, nprob));
        }
    }
    return GMRF_OK;
}

// mode: 1 forward P->Y, 2 backward P->Y, 0 full P->Y->P (result in P)
static gmrf_status run_sweeps(gmrf_handle* h, int mode, int kp) {
    auto body = [&]() -> gmrf_status {
        if (mode == GMRF_SOLVE_FORWARD) return sweep_launches(h, false, kp, h->d_P, h->d_Y);
        if (mode == GMRF_SOLVE_BACKWARD) return sweep_launches(h, true, kp, h->d_P, h->d_Y);
        GCHK(sweep_launches(h, false, kp, h->d_P, h->d_Y));
        return sweep_launches(h, true, kp, h->d_Y, h->d_P);
    };
    const bool persistent = sweep_persist_ok(h, kp);
    if (persistent) {
        GCHK(sweep_persist_prepare(h, kp));
        h->sweep_persist_launched = true;                  // (here, not at the launch: a graph replay launches it too)
        h->stats.sweep_persist = 1;
        h->stats.sweep_persist_launches += (mode == GMRF_SOLVE_FULL) ? 2 : 1;
    }
    if (h->eager || h->profiling) return body();
    const int64_t key = ((int64_t)h->xsplit << 32) | ((int64_t)mode * 4096 + kp) |       // (the representation decides the launches)
                        (persistent ? ((int64_t)1 << 60) : 0);
    auto it = h->sweep_graphs.find(key);
    if (it == h->sweep_graphs.end()) {
        std::lock_guard<std::mutex> capture_lock(g_capture_mu);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        gmrf_status s = body();
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (s != GMRF_OK) { if (graph) (void)hipGraphDestroy(graph); return s; }
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        it = h->sweep_graphs.emplace(key, exec).first;
    }
    HIPCHK(hipGraphLaunch(it->second, h->stream));
    return GMRF_OK;
}

static int pad_k(int64_t k) { return k == 1 ? 1 : (int)((k + 15) / 16 * 16); }

static gmrf_status launch_pack(gmrf_handle* h, const double* d_src, int64_t ld, int k, int kp) {
    const int64_t total = (int64_t)kp * h->n_pad;
    hipLaunchKernelGGL(pack_panel, dim3((unsigned)((total + 255) / 256), (unsigned)h->B), dim3(256), 0, h->stream, d_src, ld,
                       h->d_P, h->n_pad, (int)h->bs, (int)h->bsp, (int)h->N, k, kp);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

static gmrf_status launch_unpack(gmrf_handle* h, const double* panel, double* d_dst, int64_t ld, int k,
                                 const double* d_mean) {
    const int64_t total = h->n * (int64_t)k;
    hipLaunchKernelGGL(unpack_panel, dim3((unsigned)((total + 255) / 256), (unsigned)h->B), dim3(256), 0, h->stream, panel,
                       h->n_pad, d_dst, ld, (int)h->bs, (int)h->bsp, h->n, k, pad_k(k), d_mean);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

constexpr int64_t KP_CHUNK = 128;

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

int32_t gmrf_version(void) { return 100; }
const char* gmrf_last_error(void) { return g_last_error.c_str(); }

gmrf_status gmrf_bt_create(int32_t device, void* stream, gmrf_handle** out) {
    if (!out) return bad_shape("null out pointer");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        (void)hipGetLastError();
        g_last_error = "no HIP device visible (libgmrf_hip needs an MI355X / gfx950 GPU)";
        return GMRF_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    gmrf_handle* h = new gmrf_handle();
    memset(&h->stats, 0, sizeof(h->stats));
    h->device = device;
    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    HIPCHK(hipMalloc(&h->d_info, 4 * sizeof(int)));           // [0] info, [1] abort word of the persistent kernel
    HIPCHK(hipMemsetAsync(h->d_info, 0, 4 * sizeof(int), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_STEP_LDS));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_diag128, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)potrf_persist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_PERSIST_LDS));
    HIPCHK(hipDeviceGetAttribute(&h->cu_count, hipDeviceAttributeMultiprocessorCount, device));
    { const char* e = getenv("GMRF_PERSIST"); if (e && atoi(e) == 0) h->no_persist = true; }      // tuning aid
    { const char* e = getenv("GMRF_SWEEP_PERSIST"); if (e && atoi(e) == 0) h->no_sweep_persist = true; }      // tuning aid
    HIPCHK(hipFuncSetAttribute((const void*)potrf_diag128_slim, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(gemm_init());
    HIPCHK(gemm_dma_init());
    *out = h;
    return GMRF_OK;
}

gmrf_status gmrf_bt_destroy(gmrf_handle* h) {
    if (!h) return GMRF_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    persist_release(h);
    destroy_graphs(h);
    free_dev(h->d_keys); free_dev(h->d_vals); free_dev(h->d_src); free_dev(h->d_nz_stage);
    free_dev(h->d_lo_rowptr); free_dev(h->d_dg_rowptr); free_dev(h->d_kst); free_dev(h->d_mend);
    free_dev(h->d_bxt_uptr); free_dev(h->d_bxt_ucols); free_dev(h->d_bxt_lidx); free_dev(h->d_bxt_gtiles);
    if (!h->external_storage) { free_dev(h->d_L); free_dev(h->d_C); free_dev(h->d_Linv); }
    else if (!h->keep_l) free_dev(h->d_L);               // the one-block work buffer is ours
    free_dev(h->d_S); free_dev(h->d_B); free_dev(h->d_T); free_dev(h->d_W);
    free_dev(h->d_sweep_flags); free_dev(h->d_Tsw); free_dev(h->d_P2); free_dev(h->d_Y2); free_dev(h->d_T2);
    if (h->h_sweep_abort) { (void)hipHostFree(h->h_sweep_abort); h->h_sweep_abort = nullptr; }
    free_dev(h->d_info); free_dev(h->d_logdet); free_dev(h->d_pflags); free_dev(h->d_kbx); free_dev(h->d_V);
    free_dev(h->d_P); free_dev(h->d_Y); free_dev(h->d_Tp);
    free_dev(h->d_stage); free_dev(h->d_mean); free_dev(h->d_acc);
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (auto e : h->fork_events) (void)hipEventDestroy(e);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GMRF_OK;
}

gmrf_status gmrf_bt_set_batch(gmrf_handle* h, int64_t batch) {
    if (!h || batch < 1 || batch > 4096) return bad_shape("batch must be in [1, 4096]");
    if (batch != h->B) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        destroy_graphs(h);
        h->B = batch;
        h->sel = 0;
        h->factored = false;
        release_shape_buffers(h);
        if (h->n > 0) (void)set_shape(h, h->n, h->N);      // flop accounting follows the batch
    }
    return GMRF_OK;
}

gmrf_status gmrf_bt_select_problem(gmrf_handle* h, int64_t p) {
    if (!h || p < 0 || p >= h->B) return bad_shape("problem index out of range");
    h->sel = p;
    return GMRF_OK;
}

gmrf_status gmrf_bt_set_profiling(gmrf_handle* h, int32_t level) {
    if (!h) return bad_shape("null handle");
    h->profiling = level;
    if (level > 0 && h->events.empty()) h->gemm_shapes.clear();
    for (int i = 0; i < GMRF_KERNEL_CLASSES; ++i) { h->stats.kernel_ms[i] = h->stats.kernel_work[i] = 0; h->stats.kernel_launches[i] = 0; }
    return GMRF_OK;
}

gmrf_status gmrf_bt_set_eager(gmrf_handle* h, int32_t eager) {
    if (!h) return bad_shape("null handle");
    if (((eager & 2) != 0) != h->split_step) { destroy_graphs(h); h->split_step = (eager & 2) != 0; }
    if (((eager & 4) != 0) != h->sweep_no_gemm) { destroy_graphs(h); h->sweep_no_gemm = (eager & 4) != 0; }
    if (((eager & 8) != 0) != h->dense_g1) { destroy_graphs(h); h->dense_g1 = (eager & 8) != 0; }
    if (((eager & 16) != 0) != h->fork_graph) { destroy_graphs(h); h->fork_graph = (eager & 16) != 0; }
    h->no_staircase = (eager & 32) != 0;
    if (((eager & 128) != 0) != h->doubling_x) { destroy_graphs(h); h->doubling_x = (eager & 128) != 0; }
    if (((eager & 256) != 0) != h->no_lookahead) { destroy_graphs(h); h->no_lookahead = (eager & 256) != 0; }
    if (((eager & 4096) != 0) != h->no_xsplit) { destroy_graphs(h); h->no_xsplit = (eager & 4096) != 0; }
    if (((eager & 8192) != 0) != h->no_persist) { destroy_graphs(h); h->no_persist = (eager & 8192) != 0; }
    if (((eager & 32768) != 0) != h->no_persist_panels) { destroy_graphs(h); h->no_persist_panels = (eager & 32768) != 0; }
    if (((eager & 65536) != 0) != h->no_sweep_persist) { destroy_graphs(h); h->no_sweep_persist = (eager & 65536) != 0; }
    if (((eager & 131072) != 0) != h->no_scatter_fold) { destroy_graphs(h); h->no_scatter_fold = (eager & 131072) != 0; }
    h->eager = (eager & 1) != 0;
    return GMRF_OK;
}

gmrf_status gmrf_bt_synchronize(gmrf_handle* h) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMRF_OK;
}

gmrf_status gmrf_bt_stats(gmrf_handle* h, gmrf_stats* out) {
    if (!h || !out) return bad_shape("null pointer");
    *out = h->stats;
    return GMRF_OK;
}

gmrf_status gmrf_bt_factor_csc(gmrf_handle* h, int64_t n, int64_t n_blocks, const int64_t* colptr,
                               const int64_t* rowval, const double* nzval, int32_t index_base,
                               int32_t* info) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (info) *info = 0;
    GCHK(analyze_csc(h, n, n_blocks, colptr, rowval, index_base));
    return numeric_factor(h, nzval, info);
}

gmrf_status gmrf_bt_factor_begin_csc(gmrf_handle* h, int64_t n, int64_t n_blocks, const int64_t* colptr,
                                     const int64_t* rowval, const double* nzval, int32_t index_base) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (colptr) GCHK(analyze_csc(h, n, n_blocks, colptr, rowval, index_base));
    if (!h->analyzed) { g_last_error = "no sparsity pattern analysed"; return GMRF_ERR_NO_FACTOR; }
    GCHK(alloc_factor(h));
    if (h->c_dirty) {
        HIPCHK(hipMemsetAsync(h->d_C, 0, sizeof(double) * stride_pC(h) * h->B, h->stream));
        h->c_dirty = false;
    }
    persist_plan(h);
    GCHK(load_values(h, nzval));
    HIPCHK(hipMemsetAsync(h->d_info, 0, 4 * sizeof(int), h->stream));
    h->info_checked = 0; h->persist_launched = false; h->stats.persist_route = 0;
    h->factored = false;
    return GMRF_OK;
}

gmrf_status gmrf_bt_factor_step_async(gmrf_handle* h, int64_t i0, int64_t i1) {
    if (!h) return bad_shape("null handle");
    if (i0 < 0 || i1 > h->N || i0 > i1) return bad_shape("bad block range");
    HIPCHK(hipSetDevice(h->device));
    // block ranges are launched directly (a graph per range would have to be re-captured)
    GCHK(factor_blocks_range(h, i0, i1));
    // A range that holds persistent launches is waited for here and repeated launch-per-step if one of them gave up, so that
    // what the caller packs / broadcasts next is the factor (the call is asynchronous only for handles without such launches)
    if (h->persist_launched) GCHK(persist_check_range(h, i0, i1, nullptr));
    return GMRF_OK;
}

gmrf_status gmrf_bt_factor_end(gmrf_handle* h, int32_t* info) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    gmrf_status s = factor_finish(h, info);
    if (h->profiling) prof_collect(h);
    return s;
}

gmrf_status gmrf_bt_refactor_values(gmrf_handle* h, const double* nzval, int32_t* info) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (info) *info = 0;
    return numeric_factor(h, nzval, info);
}

gmrf_status gmrf_bt_factor_blocks(gmrf_handle* h, int64_t n, int64_t n_blocks, const gmrf_sparse_block* diag,
                                  const gmrf_sparse_block* lower, int32_t index_base,
                                  int32_t compressed_by_column, int32_t* info) {
    if (!h || !diag || (n_blocks > 1 && !lower)) return bad_shape("null pointer");
    HIPCHK(hipSetDevice(h->device));
    if (info) *info = 0;
    GCHK(set_shape(h, n, n_blocks));
    const int64_t bs = h->bs;
    std::vector<std::vector<HostEntry>> dg(n_blocks), lo(n_blocks);
    std::vector<double> vals;
    auto add_block = [&](const gmrf_sparse_block& b, std::vector<HostEntry>& dst, bool lower_tri_only) -> bool {
        for (int64_t o = 0; o < bs; ++o) {
            for (int64_t p = b.ptr[o] - index_base; p < b.ptr[o + 1] - index_base; ++p) {
                const int64_t in = b.idx[p] - index_base;
                if (in < 0 || in >= bs) return false;
                const int64_t r = compressed_by_column ? in : o, c = compressed_by_column ? o : in;
                if (lower_tri_only && r < c) continue;
                dst.push_back({((uint64_t)r << 32) | (uint64_t)c, (int64_t)vals.size()});
                vals.push_back(b.val[p]);
            }
        }
        return true;
    };
    for (int64_t i = 0; i < n_blocks; ++i) {
        if (!add_block(diag[i], dg[i], true)) return bad_shape("block index out of range");
        if (i > 0 && !add_block(lower[i - 1], lo[i], false)) return bad_shape("block index out of range");
    }
    GCHK(upload_entries(h, dg, lo, (int64_t)vals.size()));
    return numeric_factor(h, vals.data(), info);
}


gmrf_status gmrf_bt_storage_bytes(int64_t n, int64_t n_blocks, int64_t batch, int64_t* bytes_L, int64_t* bytes_C,
                                  int64_t* bytes_Linv) {
    if (n <= 0 || n_blocks <= 0 || n % n_blocks != 0 || batch < 1 || !bytes_L || !bytes_C || !bytes_Linv)
        return bad_shape("n must be a positive multiple of N_blocks, batch >= 1");
    const int64_t bs = n / n_blocks, bsp = 64 * next_pow2((bs + 63) / 64);
    const int64_t blk = bsp * bsp * (int64_t)sizeof(double) * batch;
    *bytes_L = blk * n_blocks; *bytes_Linv = blk * n_blocks; *bytes_C = blk * std::max<int64_t>(n_blocks - 1, 1);
    return GMRF_OK;
}

gmrf_status gmrf_bt_set_storage(gmrf_handle* h, int64_t n, int64_t n_blocks, int64_t batch, void* dev_L, void* dev_C,
                                void* dev_Linv) {
    if (!h || !dev_C || !dev_Linv) return bad_shape("null pointer");
    if (batch != h->B) return bad_shape("storage batch differs from the handle's batch (gmrf_bt_set_batch first)");
    if (!dev_L && h->keep_l) return bad_shape("dev_L may be NULL only after gmrf_bt_set_keep_l(h, 0)");
    HIPCHK(hipSetDevice(h->device));
    GCHK(set_shape(h, n, n_blocks));
    HIPCHK(hipStreamSynchronize(h->stream));
    destroy_graphs(h);
    if (!h->external_storage) { free_dev(h->d_L); free_dev(h->d_C); free_dev(h->d_Linv); }
    else if (!h->keep_l) free_dev(h->d_L);             // our one-block work buffer of an earlier set_storage
    free_dev(h->d_S); free_dev(h->d_B); free_dev(h->d_T); free_dev(h->d_W); free_dev(h->d_logdet);
    h->d_S = h->d_B = h->d_T = h->d_W = h->d_logdet = nullptr;
    h->d_L = (double*)dev_L; h->d_C = (double*)dev_C; h->d_Linv = (double*)dev_Linv;
    h->external_storage = true;
    h->factored = false; h->l_valid = false; h->logdet_valid = false;   // whatever was factored lived in the old buffers
    if (!h->keep_l) {                                  // one-block work buffer for L_i
        HIPCHK(hipMalloc(&h->d_L, (size_t)blk_elems(h) * sizeof(double) * (size_t)h->B));
        HIPCHK(hipMemsetAsync(h->d_L, 0, (size_t)blk_elems(h) * sizeof(double) * (size_t)h->B, h->stream));
    }
    if (!h->analyzed) GCHK(set_layout_dense(h));      // until a pattern is analysed / a layout adopted
    // tiles strictly above the block diagonal of L / Linv are never written: keep them zero
    const size_t blk = (size_t)blk_elems(h) * sizeof(double) * (size_t)h->B;
    if (h->keep_l) HIPCHK(hipMemsetAsync(h->d_L, 0, blk * h->N, h->stream));
    HIPCHK(hipMemsetAsync(h->d_Linv, 0, blk * h->N, h->stream));
    HIPCHK(hipMemsetAsync(h->d_C, 0, blk * std::max<int64_t>(h->N - 1, 1), h->stream));
    h->c_dirty = false;
    GCHK(alloc_work(h));
    h->alloc_rm = h->rmax; h->alloc_wc = c_ld(h); h->alloc_keep_l = h->keep_l;
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMRF_OK;
}

gmrf_status gmrf_bt_set_keep_l(gmrf_handle* h, int32_t keep) {
    if (!h) return bad_shape("null handle");
    if ((keep != 0) == h->keep_l) return GMRF_OK;
    if (h->external_storage) return bad_shape("choose gmrf_bt_set_keep_l before gmrf_bt_set_storage");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    destroy_graphs(h);
    h->keep_l = keep != 0;
    h->factored = false; h->l_valid = false; h->logdet_valid = false;
    return GMRF_OK;
}

// Layout of the stored coupling blocks: out = {cmin, rmax, n_row_tiles, kst[0 .. n_row_tiles)} (kst relative to cmin).
gmrf_status gmrf_bt_get_layout(gmrf_handle* h, int64_t* out, int64_t cap, int64_t* count) {
    if (!h || !count) return bad_shape("null pointer");
    if (h->N <= 0 || h->kst.empty()) { g_last_error = "no shape"; return GMRF_ERR_NO_FACTOR; }
    const int64_t nrt = h->rmax / 64;
    *count = 4 + nrt;
    if (!out) return GMRF_OK;
    if (cap < 4 + nrt) return bad_shape("layout buffer too small");
    out[0] = h->cmin; out[1] = h->rmax; out[2] = nrt;
    for (int64_t t = 0; t < nrt; ++t) out[3 + t] = h->kst[(size_t)t];
    // representation of the block inverses (gmrf_handle::xsplit): of the stored factor, or what the next factorisation will leave
    out[3 + nrt] = h->factored ? h->xsplit : planned_xsplit(h);
    return GMRF_OK;
}

// A rank that receives the factor by broadcast: shape + the root's layout (NULL: dense coupling blocks),
// storage allocated (or the caller's, gmrf_bt_set_storage), nothing factored yet.
gmrf_status gmrf_bt_adopt_layout(gmrf_handle* h, int64_t n, int64_t n_blocks, const int64_t* layout, int64_t count) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    GCHK(set_shape(h, n, n_blocks));
    h->analyzed = false;                               // whatever pattern was analysed does not describe this factor
    if (layout) {
        if (count < 3 || layout[2] < 1 || count < 3 + layout[2] || layout[1] != 64 * layout[2]) return bad_shape("bad layout record");
        std::vector<int64_t> first((size_t)layout[2]);
        for (int64_t t = 0; t < layout[2]; ++t) first[(size_t)t] = layout[0] + layout[3 + t];
        GCHK(set_layout(h, layout[0], layout[1], first));
        h->adopt_xsplit = 0;
        if (count >= 4 + layout[2]) {
            const int64_t xs = layout[3 + layout[2]];
            if (xs < 0 || xs >= h->bsp || xs % 64) return bad_shape("bad layout record (split)");
            h->adopt_xsplit = (int)xs;
        }
    } else {
        GCHK(set_layout_dense(h));
        h->adopt_xsplit = 0;
    }
    GCHK(alloc_factor(h));
    if (h->c_dirty) {
        HIPCHK(hipMemsetAsync(h->d_C, 0, sizeof(double) * stride_pC(h) * h->B, h->stream));
        h->c_dirty = false;
    }
    h->factored = false; h->l_valid = false; h->logdet_valid = false;
    h->got_block.assign((size_t)h->N, 0);
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMRF_OK;
}

gmrf_status gmrf_bt_adopt_shape(gmrf_handle* h, int64_t n, int64_t n_blocks) {
    return gmrf_bt_adopt_layout(h, n, n_blocks, nullptr, 0);
}

// l_blocks_valid != 0: the caller also filled the L buffer (gmrf_bt_factor_buffer kind L), so F.chos / logdet work
gmrf_status gmrf_bt_adopt_commit(gmrf_handle* h, int32_t l_blocks_valid) {
    if (!h || !h->d_Linv) return bad_shape("no factor storage");
    HIPCHK(hipSetDevice(h->device));
    // Representation of the block inverses: a packed image says which form ITS sender was in when it packed (the layout record
    // may be older than that: the sender converts to the full form for get_block / export / exact variances and goes back to the
    // split form at its next factorisation); raw buffer transfers carry no tag and follow the record.
    int seen = 0;
    HIPCHK(hipMemcpyAsync(&seen, h->d_info + 2, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (seen != 0) HIPCHK(hipMemsetAsync(h->d_info + 2, 0, sizeof(int), h->stream));
    if (seen < 0) { g_last_error = "the packed ranges of this factor carry different (or no) representation tags"; return GMRF_ERR_BAD_SHAPE; }
    const int xs = seen > 0 ? seen - 1 : h->adopt_xsplit;
    if (xs < 0 || xs >= h->bsp || xs % 64) { g_last_error = "bad representation tag in the packed image"; return GMRF_ERR_BAD_SHAPE; }
    h->factored = true;
    h->xsplit = xs;
    h->l_valid = l_blocks_valid != 0 && h->keep_l;
    // the log-determinant parts travel inside the packed transport image (gmrf_bt_unpack_blocks_async): valid once
    // every block of this factor came that way; a factor moved as raw buffers has none (logdet then needs the L blocks)
    bool all = !h->got_block.empty();
    for (char g : h->got_block) all = all && g;
    h->logdet_valid = all;
    h->got_block.assign((size_t)h->N, 0);
    persist_plan(h);                   // (a receiver's sweeps may be persistent launches too: the claim is planned where a factor becomes usable)
    return GMRF_OK;
}

gmrf_status gmrf_bt_factor_buffer(gmrf_handle* h, int32_t kind, void** dev_ptr, int64_t* bytes) {
    if (!h || !dev_ptr || !bytes) return bad_shape("null pointer");
    if (!h->d_Linv) { g_last_error = "no factor storage"; return GMRF_ERR_NO_FACTOR; }
    const int64_t blk = blk_elems(h) * (int64_t)sizeof(double) * h->B;
    if (kind == GMRF_BLOCK_L) {
        if (!h->keep_l) { g_last_error = "L blocks are not kept (gmrf_bt_set_keep_l)"; return GMRF_ERR_NO_FACTOR; }
        *dev_ptr = h->d_L; *bytes = blk * h->N;
    }
    else if (kind == GMRF_BLOCK_C) { *dev_ptr = h->d_C; *bytes = stride_pC(h) * (int64_t)sizeof(double) * h->B; }
    else if (kind == GMRF_BLOCK_LINV) { *dev_ptr = h->d_Linv; *bytes = blk * h->N; }
    else return bad_shape("bad block kind");
    return GMRF_OK;
}

// Element ranges (in doubles, per problem) of blocks [i0, i1) inside the three factor buffers: what a
// block-range broadcast moves.  C holds the coupling blocks i0-1 .. i1-2 (C_i couples blocks i and i+1).
gmrf_status gmrf_bt_block_range(gmrf_handle* h, int32_t kind, int64_t i0, int64_t i1, int64_t* first_elem,
                                int64_t* n_elems, int64_t* problem_stride) {
    if (!h || !first_elem || !n_elems || !problem_stride) return bad_shape("null pointer");
    if (h->N <= 0 || i0 < 0 || i1 > h->N || i0 > i1) return bad_shape("bad block range");
    if (kind == GMRF_BLOCK_C) {
        const int64_t c0 = std::max<int64_t>(i0 - 1, 0), c1 = std::max<int64_t>(i1 - 1, 0);
        *first_elem = c0 * c_blk(h); *n_elems = (c1 - c0) * c_blk(h); *problem_stride = stride_pC(h);
    } else if (kind == GMRF_BLOCK_LINV || kind == GMRF_BLOCK_L) {
        if (kind == GMRF_BLOCK_L && !h->keep_l) { g_last_error = "L blocks are not kept"; return GMRF_ERR_NO_FACTOR; }
        *first_elem = i0 * blk_elems(h); *n_elems = (i1 - i0) * blk_elems(h);
        *problem_stride = kind == GMRF_BLOCK_L ? stride_pL(h) : stride_pX(h);
    } else return bad_shape("bad block kind");
    return GMRF_OK;
}

// Packed transport image of the blocks [i0, i1) of every problem of the batch: per problem one segment of
//   (i1 - i0) * ntri * 4096   lower-triangular 64 x 64 tiles of Linv_i0 .. Linv_{i1-1} (linv_tiles_copy),
//   (c1 - c0) * c_blk          stored windows of the coupling blocks C_{i0-1} .. C_{i1-2},
//   2                          representation tag {xsplit of the sender, magic} (pack_tag_write: the receiver's commit follows it)
//   (i1 - i0) rounded to even  log-determinant parts of the blocks
// doubles: darcy256 0.56 GB per posterior instead of the 0.83 GB of the raw Linv / C buffers.
static void packed_counts(const gmrf_handle* h, int64_t i0, int64_t i1, int64_t* x_elems, int64_t* c_elems, int64_t* ld_elems,
                          int64_t* c0_out) {
    const int64_t nt = h->bsp / 64, ntri = nt * (nt + 1) / 2;
    const int64_t c0 = std::max<int64_t>(i0 - 1, 0), c1 = std::max<int64_t>(i1 - 1, 0);
    *x_elems = (i1 - i0) * ntri * 4096;
    *c_elems = (c1 - c0) * c_blk(h);
    *ld_elems = 2 + ((i1 - i0) + 1) / 2 * 2;
    if (c0_out) *c0_out = c0;
}

gmrf_status gmrf_bt_packed_size(gmrf_handle* h, int64_t i0, int64_t i1, int64_t* elems_per_problem) {
    if (!h || !elems_per_problem) return bad_shape("null pointer");
    if (h->N <= 0 || i0 < 0 || i1 > h->N || i0 >= i1) return bad_shape("bad block range");
    int64_t xe, ce, le;
    packed_counts(h, i0, i1, &xe, &ce, &le, nullptr);
    *elems_per_problem = xe + ce + le;
    return GMRF_OK;
}

static gmrf_status pack_blocks_on(gmrf_handle* h, hipStream_t st, int64_t i0, int64_t i1, double* buf, bool pack) {
    if (!h->d_Linv || h->N <= 0) { g_last_error = "no factor storage"; return GMRF_ERR_NO_FACTOR; }
    if (i0 < 0 || i1 > h->N || i0 >= i1) return bad_shape("bad block range");
    if (!buf || !is_device_ptr(buf)) return bad_shape("the transport image lives in device memory");
    int64_t xe, ce, le, c0;
    packed_counts(h, i0, i1, &xe, &ce, &le, &c0);
    const int64_t seg = xe + ce + le;
    const int nt = (int)(h->bsp / 64), ntri = nt * (nt + 1) / 2;
    const dim3 grid((unsigned)(ntri * (i1 - i0)), (unsigned)h->B);
    double* X = h->d_Linv + i0 * blk_elems(h);
    if (pack) hipLaunchKernelGGL(linv_tiles_copy<true>, grid, dim3(256), 0, st, X, h->bsp, blk_elems(h), stride_pX(h), buf, seg, ntri);
    else hipLaunchKernelGGL(linv_tiles_copy<false>, grid, dim3(256), 0, st, X, h->bsp, blk_elems(h), stride_pX(h), buf, seg, ntri);
    HIPCHK(hipGetLastError());
    if (ce > 0) {
        double* Cw = h->d_C + c0 * c_blk(h);
        if (pack) HIPCHK(hipMemcpy2DAsync(buf + xe, seg * sizeof(double), Cw, stride_pC(h) * sizeof(double), ce * sizeof(double),
                                          (size_t)h->B, hipMemcpyDeviceToDevice, st));
        else HIPCHK(hipMemcpy2DAsync(Cw, stride_pC(h) * sizeof(double), buf + xe, seg * sizeof(double), ce * sizeof(double),
                                     (size_t)h->B, hipMemcpyDeviceToDevice, st));
    }
    if (pack && h->keep_l) {
        // with the L blocks resident nobody has taken their log-determinant parts yet
        hipLaunchKernelGGL(logdet_blocks, dim3((unsigned)(i1 - i0), (unsigned)h->B), dim3(256), 0, st, h->d_L + i0 * blk_elems(h),
                           blk_elems(h), h->bsp, (int)h->bs, h->d_logdet + i0, stride_pL(h), h->N);
        HIPCHK(hipGetLastError());
    }
    if (pack) {
        if (h->B > 1024) return bad_shape("batch too large for the transport tag");
        hipLaunchKernelGGL(pack_tag_write, dim3(1), dim3((unsigned)h->B), 0, st, buf + xe + ce, seg, (double)h->xsplit);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(buf + xe + ce + 2, seg * sizeof(double), h->d_logdet + i0, h->N * sizeof(double),
                                (i1 - i0) * sizeof(double), (size_t)h->B, hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(pack_tag_read, dim3(1), dim3(1), 0, st, buf + xe + ce, seg, (int)h->B, h->d_info + 2);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(h->d_logdet + i0, h->N * sizeof(double), buf + xe + ce + 2, seg * sizeof(double),
                                (i1 - i0) * sizeof(double), (size_t)h->B, hipMemcpyDeviceToDevice, st));
        if (h->got_block.size() != (size_t)h->N) h->got_block.assign((size_t)h->N, 0);
        for (int64_t i = i0; i < i1; ++i) h->got_block[(size_t)i] = 1;
    }
    return GMRF_OK;
}

gmrf_status gmrf_bt_pack_blocks_async(gmrf_handle* h, int64_t i0, int64_t i1, double* dev_buf) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    return pack_blocks_on(h, h->stream, i0, i1, dev_buf, true);
}

gmrf_status gmrf_bt_unpack_blocks_async(gmrf_handle* h, int64_t i0, int64_t i1, const double* dev_buf) {
    if (!h) return bad_shape("null handle");
    HIPCHK(hipSetDevice(h->device));
    return pack_blocks_on(h, h->stream, i0, i1, const_cast<double*>(dev_buf), false);
}

gmrf_status gmrf_bt_get_block(gmrf_handle* h, int32_t kind, int64_t i, double* out, int64_t ld) {
    if (!h || !out) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "no factor"; return GMRF_ERR_NO_FACTOR; }
    HIPCHK(hipSetDevice(h->device));
    const int64_t bs = h->bs, bsp = h->bsp;
    if (ld < bs) return bad_shape("ld < block_size");
    // source window [rows][cols] with row stride lds, placed at (0, c0) of the logical block
    const double* src;
    int64_t rows = bsp, cols = bsp, lds = bsp, c0 = 0;
    if (kind == GMRF_BLOCK_L) {
        if (i < 0 || i >= h->N) return bad_shape("block index");
        if (!h->l_valid) { g_last_error = "the L blocks of this factor are not resident (gmrf_bt_set_keep_l(h, 0), or a factor adopted without them)"; return GMRF_ERR_NO_FACTOR; }
        src = h->d_L + h->sel * stride_pL(h) + i * blk_elems(h);
    } else if (kind == GMRF_BLOCK_LINV) {
        if (i < 0 || i >= h->N) return bad_shape("block index");
        GCHK(ensure_full_inverse(h));
        src = h->d_Linv + h->sel * stride_pX(h) + i * blk_elems(h);
    } else if (kind == GMRF_BLOCK_C) {
        if (i < 0 || i >= h->N - 1) return bad_shape("block index");
        src = h->d_C + h->sel * stride_pC(h) + i * c_blk(h);
        rows = h->rmax; cols = c_ld(h); lds = cols; c0 = h->cmin;
    } else return bad_shape("bad block kind");
    std::vector<double> tmp((size_t)rows * cols);
    HIPCHK(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const bool dev_out = is_device_ptr(out);
    std::vector<double> cm;
    double* dst = out;
    int64_t ldd = ld;
    if (dev_out) { cm.resize((size_t)bs * bs); dst = cm.data(); ldd = bs; }
    for (int64_t c = 0; c < bs; ++c)
        for (int64_t r = 0; r < bs; ++r) {                 // row-major window -> column-major block
            const bool in = r < rows && c >= c0 && c - c0 < cols;
            dst[c * ldd + r] = in ? tmp[(size_t)r * lds + (c - c0)] : 0.0;
        }
    if (dev_out) {
        HIPCHK(hipMemcpy2DAsync(out, ld * sizeof(double), cm.data(), bs * sizeof(double), bs * sizeof(double), bs, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GMRF_OK;
}

// Flat image of one problem's factor: int64 header[8] = {magic, version, n, N, bs, 1, 0, 0}, then
// column-major bs x bs blocks: L_1..L_N, C_1..C_{N-1}, Linv_1..Linv_N  (SURVEY 8b export/import).
static const int64_t EXPORT_MAGIC = 0x46524d47;   // "GMRF"

gmrf_status gmrf_bt_export_size(gmrf_handle* h, int64_t* bytes) {
    if (!h || !bytes) return bad_shape("null pointer");
    if (h->N <= 0) { g_last_error = "no shape"; return GMRF_ERR_NO_FACTOR; }
    *bytes = 64 + (int64_t)sizeof(double) * h->bs * h->bs * (3 * h->N - 1);
    return GMRF_OK;
}

gmrf_status gmrf_bt_export_factor(gmrf_handle* h, void* buf, int64_t bytes) {
    if (!h || !buf) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "no factor"; return GMRF_ERR_NO_FACTOR; }
    if (is_device_ptr(buf)) return bad_shape("export buffer must be host memory");
    int64_t need = 0;
    GCHK(gmrf_bt_export_size(h, &need));
    if (bytes < need) return bad_shape("export buffer too small");
    int64_t* hdr = (int64_t*)buf;
    hdr[0] = EXPORT_MAGIC; hdr[1] = 1; hdr[2] = h->n; hdr[3] = h->N; hdr[4] = h->bs; hdr[5] = 1; hdr[6] = hdr[7] = 0;
    double* out = (double*)((char*)buf + 64);
    const int64_t be = h->bs * h->bs;
    for (int64_t i = 0; i < h->N; ++i) GCHK(gmrf_bt_get_block(h, GMRF_BLOCK_L, i, out + i * be, h->bs));
    out += h->N * be;
    for (int64_t i = 0; i + 1 < h->N; ++i) GCHK(gmrf_bt_get_block(h, GMRF_BLOCK_C, i, out + i * be, h->bs));
    out += (h->N - 1) * be;
    for (int64_t i = 0; i < h->N; ++i) GCHK(gmrf_bt_get_block(h, GMRF_BLOCK_LINV, i, out + i * be, h->bs));
    return GMRF_OK;
}

gmrf_status gmrf_bt_import_factor(gmrf_handle* h, const void* buf, int64_t bytes) {
    if (!h || !buf) return bad_shape("null pointer");
    if (is_device_ptr(buf)) return bad_shape("import buffer must be host memory");
    if (bytes < 64) return bad_shape("import buffer too small");
    const int64_t* hdr = (const int64_t*)buf;
    if (hdr[0] != EXPORT_MAGIC || hdr[1] != 1 || hdr[5] != 1) return bad_shape("not a factor image");
    const int64_t n = hdr[2], N = hdr[3], bs = hdr[4];
    if (N <= 0 || bs <= 0 || n != N * bs) return bad_shape("bad shape in factor image");
    if (bytes < 64 + (int64_t)sizeof(double) * bs * bs * (3 * N - 1)) return bad_shape("factor image truncated");
    HIPCHK(hipSetDevice(h->device));
    if (!h->keep_l) return bad_shape("import needs the L blocks kept (gmrf_bt_set_keep_l(h, 1))");
    if ((h->n != n || h->N != N) && h->B != 1) return bad_shape("a batched handle imports into an existing shape only");
    // The image holds dense blocks and says nothing about the coupling blocks' zero structure: the handle
    // takes the dense layout (a pattern analysed earlier no longer describes the factor) and every
    // problem's stored C window is re-laid out accordingly.
    const bool relayout = (h->n != n || h->N != N || h->cmin != 0 || h->rmax != 64 * next_pow2((bs + 63) / 64) || !h->d_Linv);
    if (relayout && h->B != 1 && h->factored) return bad_shape("a batched handle imports only into a dense-layout factor");
    GCHK(set_shape(h, n, N));
    h->analyzed = false;
    if (relayout) {
        GCHK(set_layout_dense(h));
        GCHK(alloc_factor(h));
        if (h->c_dirty) {
            HIPCHK(hipMemsetAsync(h->d_C, 0, sizeof(double) * stride_pC(h) * h->B, h->stream));
            h->c_dirty = false;
        }
    } else {
        std::vector<int64_t> z((size_t)(h->bsp / 64), 0);
        bool dense_ks = true;
        for (int v : h->kst) if (v != 0) dense_ks = false;
        if (!dense_ks) GCHK(set_layout(h, 0, h->bsp, z));      // same window, staircase dropped
        h->c_dirty = false;
    }
    const int64_t bsp = h->bsp, be = bs * bs;
    std::vector<double> tmp((size_t)bsp * bsp);
    auto put = [&](const double* src, double* dst, bool unit_pad) -> gmrf_status {
        std::fill(tmp.begin(), tmp.end(), 0.0);
        for (int64_t r = 0; r < bs; ++r)
            for (int64_t c = 0; c < bs; ++c) tmp[(size_t)r * bsp + c] = src[c * bs + r];   // column-major -> row-major
        if (unit_pad)
            for (int64_t r = bs; r < bsp; ++r) tmp[(size_t)r * bsp + r] = 1.0;
        HIPCHK(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return GMRF_OK;
    };
    const double* in = (const double*)((const char*)buf + 64);
    for (int64_t i = 0; i < N; ++i) GCHK(put(in + i * be, h->d_L + h->sel * stride_pL(h) + i * blk_elems(h), true));
    in += N * be;
    for (int64_t i = 0; i + 1 < N; ++i) GCHK(put(in + i * be, h->d_C + h->sel * stride_pC(h) + i * c_blk(h), false));
    in += (N - 1) * be;
    for (int64_t i = 0; i < N; ++i) GCHK(put(in + i * be, h->d_Linv + h->sel * stride_pX(h) + i * blk_elems(h), true));
    h->factored = true; h->l_valid = true;
    persist_plan(h);
    h->xsplit = 0;                                     // the image holds the full inverses
    return GMRF_OK;
}

// ------------------------------------------------------------------------------------ RCCL (xGMI)
// One communicator per process and GPU.  librccl is opened at run time (dlopen) the first time a
// communicator is asked for: a host that never shares a factor never loads it, and inside a PyTorch
// process the copy PyTorch already mapped is the one that gets used (two RCCL images in one process
// would both export the nccl* symbols).
struct NcclUid { char b[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static gmrf_status rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return GMRF_OK;
    const char* env = getenv("GMRF_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (int pass = 0; pass < 2 && !lib; ++pass)          // pass 0: an image that is already mapped (PyTorch's)
        for (const char* nm : names) {
            if (!nm) continue;
            lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (lib) break;
        }
    if (!lib) { g_last_error = std::string("librccl not found (set GMRF_RCCL_PATH): ") + (dlerror() ? dlerror() : ""); return GMRF_ERR_RCCL; }
#define GMRF_RCCL_SYM(field, name)                                                           \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));               \
    if (!g_rccl.field) { g_last_error = std::string("librccl lacks ") + name; return GMRF_ERR_RCCL; }
    GMRF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") GMRF_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    GMRF_RCCL_SYM(CommDestroy, "ncclCommDestroy") GMRF_RCCL_SYM(Broadcast, "ncclBroadcast")
    GMRF_RCCL_SYM(AllReduce, "ncclAllReduce") GMRF_RCCL_SYM(AllGather, "ncclAllGather") GMRF_RCCL_SYM(GroupStart, "ncclGroupStart")
    GMRF_RCCL_SYM(GroupEnd, "ncclGroupEnd") GMRF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GMRF_RCCL_SYM
    g_rccl.lib = lib;
    return GMRF_OK;
}

#define NCCLCHK(expr)                                                                       \
    do {                                                                                    \
        int _r = (expr);                                                                    \
        if (_r != 0) {                                                                      \
            g_last_error = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);           \
            return GMRF_ERR_RCCL;                                                           \
        }                                                                                   \
    } while (0)

struct gmrf_comm {
    int device = 0, rank = 0, world = 1;
    void* comm = nullptr;
    hipStream_t stream = nullptr;          // collectives run here, beside the handle's compute stream
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    void* d_small = nullptr;               // staging of small host records (layout, scalars)
    size_t small_cap = 0;
    double* d_pack = nullptr;              // staging of the packed transport image of a block range
    size_t pack_cap = 0;
    double bytes_moved = 0.0;              // factor bytes broadcast so far (gmrf_comm_bytes)
};

gmrf_status gmrf_comm_unique_id(void* id128) {
    if (!id128) return bad_shape("null pointer");
    GCHK(rccl_load());
    NcclUid u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id128, u.b, 128);
    return GMRF_OK;
}

gmrf_status gmrf_comm_create(int32_t device, int32_t rank, int32_t world, const void* id128, gmrf_comm** out) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return bad_shape("bad communicator arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        (void)hipGetLastError();
        g_last_error = "no HIP device visible";
        return GMRF_ERR_NO_DEVICE;
    }
    GCHK(rccl_load());
    HIPCHK(hipSetDevice(device));
    gmrf_comm* c = new gmrf_comm();
    c->device = device; c->rank = rank; c->world = world;
    NcclUid u;
    memcpy(u.b, id128, 128);
    int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) { g_last_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return GMRF_ERR_RCCL; }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    *out = c;
    return GMRF_OK;
}

gmrf_status gmrf_comm_destroy(gmrf_comm* c) {
    if (!c) return GMRF_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    free_dev(c->d_small); free_dev(c->d_pack);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GMRF_OK;
}

// Broadcast a small HOST record (layout, log-determinants ...) from `root`; synchronous.
gmrf_status gmrf_comm_bcast_host(gmrf_comm* c, void* host_buf, int64_t bytes, int32_t root) {
    if (!c || !host_buf || bytes <= 0 || root < 0 || root >= c->world) return bad_shape("bad broadcast arguments");
    HIPCHK(hipSetDevice(c->device));
    if ((size_t)bytes > c->small_cap) {
        free_dev(c->d_small); c->d_small = nullptr; c->small_cap = 0;
        HIPCHK(hipMalloc(&c->d_small, (size_t)bytes));
        c->small_cap = (size_t)bytes;
    }
    if (c->rank == root) HIPCHK(hipMemcpyAsync(c->d_small, host_buf, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(g_rccl.Broadcast(c->d_small, c->d_small, (size_t)bytes, /*ncclInt8*/ 0, root, c->comm, c->stream));
    if (c->rank != root) HIPCHK(hipMemcpyAsync(host_buf, c->d_small, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GMRF_OK;
}

// R2 of SURVEY 8e: sum of the ranks' variance accumulators (device buffer of `count` doubles), in place.
// `stream_of`: a handle whose stream produced the buffer and consumes the result (may be NULL: synchronous).
gmrf_status gmrf_comm_allreduce_sum(gmrf_comm* c, gmrf_handle* stream_of, double* dev_buf, int64_t count) {
    if (!c || !dev_buf || count <= 0) return bad_shape("bad all-reduce arguments");
    if (!is_device_ptr(dev_buf)) return bad_shape("all-reduce needs a device buffer");
    HIPCHK(hipSetDevice(c->device));
    if (stream_of) {
        HIPCHK(hipEventRecord(c->ev_in, stream_of->stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    }
    NCCLCHK(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, c->stream));
    if (stream_of) {
        HIPCHK(hipEventRecord(c->ev_out, c->stream));
        HIPCHK(hipStreamWaitEvent(stream_of->stream, c->ev_out, 0));
    } else {
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return GMRF_OK;
}

// R1 of SURVEY 8e: broadcast of the factor blocks [i0, i1) of every problem of the batch from `root` to all ranks
// as ONE packed transport image (pack_blocks_on: lower-triangular tiles of Linv_i, the stored windows of the coupling
// blocks C_{i0-1} .. C_{i1-2}, the blocks' log-determinant parts): the root packs into the communicator's staging
// buffer, one ncclBroadcast moves it, the receivers unpack into their factor storage -- all three on the
// communicator's stream, behind whatever the handle's stream holds at the time of the call (root: the factorisation
// of these blocks; receivers: the sweeps of the previous job that still read the buffers).  The root goes on
// factoring the next range meanwhile.  with_l != 0: the raw L_i blocks follow (full squares).
gmrf_status gmrf_bt_bcast_blocks_async(gmrf_handle* h, gmrf_comm* c, int32_t root, int64_t i0, int64_t i1, int32_t with_l) {
    if (!h || !c) return bad_shape("null pointer");
    if (!h->d_Linv || h->N <= 0) { g_last_error = "no factor storage"; return GMRF_ERR_NO_FACTOR; }
    if (i0 < 0 || i1 > h->N || i0 >= i1 || root < 0 || root >= c->world) return bad_shape("bad block range / root");
    if (with_l && !h->keep_l) return bad_shape("the L blocks are not kept on this handle");
    if (c->device != h->device) return bad_shape("communicator and handle live on different devices");
    HIPCHK(hipSetDevice(h->device));
    int64_t seg = 0;
    GCHK(gmrf_bt_packed_size(h, i0, i1, &seg));
    const size_t need = (size_t)seg * (size_t)h->B * sizeof(double);
    if (need > c->pack_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));          // earlier transfers still use the old staging buffer
        free_dev(c->d_pack); c->d_pack = nullptr; c->pack_cap = 0;
        HIPCHK(hipMalloc(&c->d_pack, need));
        c->pack_cap = need;
    }
    HIPCHK(hipEventRecord(c->ev_in, h->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    if (c->rank == root) GCHK(pack_blocks_on(h, c->stream, i0, i1, c->d_pack, true));
    NCCLCHK(g_rccl.Broadcast(c->d_pack, c->d_pack, (size_t)seg * (size_t)h->B, /*ncclFloat64*/ 8, root, c->comm, c->stream));
    c->bytes_moved += (double)need;
    if (c->rank != root) GCHK(pack_blocks_on(h, c->stream, i0, i1, c->d_pack, false));
    if (with_l) {
        int64_t first = 0, cnt = 0, pstride = 0;
        GCHK(gmrf_bt_block_range(h, GMRF_BLOCK_L, i0, i1, &first, &cnt, &pstride));
        NCCLCHK(g_rccl.GroupStart());
        int rc = 0;
        for (int64_t p = 0; p < h->B && rc == 0; ++p) {
            double* ptr = h->d_L + p * pstride + first;
            rc = g_rccl.Broadcast(ptr, ptr, (size_t)cnt, /*ncclFloat64*/ 8, root, c->comm, c->stream);
        }
        const int rg = g_rccl.GroupEnd();                  // the group is closed on the error path too
        if (rc != 0 || rg != 0) { g_last_error = std::string("ncclBroadcast: ") + g_rccl.GetErrorString(rc ? rc : rg); return GMRF_ERR_RCCL; }
        c->bytes_moved += (double)cnt * (double)h->B * sizeof(double);
    }
    return GMRF_OK;
}

// The other way to share a batch of factors (round 4): EVERY rank factors its own share of the batch -- `src`, src->B
// posteriors -- and the packed images of the blocks [i0, i1) are all-gathered into `dst`, a handle of the same shape and layout
// with a batch of world * src->B: problem r * src->B + p of dst is problem p of rank r.  Against the root broadcast every link
// of the fabric carries a share of the traffic (per rank (world - 1) / world of the batch comes in over all its links, instead of
// the whole batch leaving the root over each of ITS links) and the factorisation itself is spread over the ranks.
// Enqueued on the communicator's stream behind src's stream; gmrf_comm_wait(dst, c) orders dst's stream behind the unpack.
gmrf_status gmrf_bt_allgather_blocks_async(gmrf_handle* src, gmrf_handle* dst, gmrf_comm* c, int64_t i0, int64_t i1) {
    if (!src || !dst || !c) return bad_shape("null pointer");
    if (!src->d_Linv || !dst->d_Linv || src->N <= 0) { g_last_error = "no factor storage"; return GMRF_ERR_NO_FACTOR; }
    if (src->n != dst->n || src->N != dst->N || src->bsp != dst->bsp || src->cmin != dst->cmin || src->rmax != dst->rmax || src->kst != dst->kst)
        return bad_shape("the two handles differ in shape or coupling-block layout (gmrf_bt_adopt_layout the gathering handle with the factoring handle's record)");
    if (dst->B != src->B * (int64_t)c->world) return bad_shape("the gathering handle's batch must be world x the factoring handle's");
    if (i0 < 0 || i1 > src->N || i0 >= i1) return bad_shape("bad block range");
    if (c->device != src->device || c->device != dst->device) return bad_shape("communicator and handles live on different devices");
    HIPCHK(hipSetDevice(src->device));
    int64_t seg = 0;
    GCHK(gmrf_bt_packed_size(src, i0, i1, &seg));
    const size_t own = (size_t)seg * (size_t)src->B, all = own * (size_t)c->world;
    const size_t need = (own + all) * sizeof(double);          // [own image | gathered images]
    if (need > c->pack_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(c->d_pack); c->d_pack = nullptr; c->pack_cap = 0;
        HIPCHK(hipMalloc(&c->d_pack, need));
        c->pack_cap = need;
    }
    HIPCHK(hipEventRecord(c->ev_in, src->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    HIPCHK(hipEventRecord(c->ev_in, dst->stream));            // dst's earlier sweeps still read the blocks the unpack overwrites
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    GCHK(pack_blocks_on(src, c->stream, i0, i1, c->d_pack, true));
    NCCLCHK(g_rccl.AllGather(c->d_pack, c->d_pack + own, own, /*ncclFloat64*/ 8, c->comm, c->stream));
    c->bytes_moved += (double)(all - own) * sizeof(double);    // what came in over this rank's links
    GCHK(pack_blocks_on(dst, c->stream, i0, i1, c->d_pack + own, false));
    return GMRF_OK;
}

// Bytes this communicator has broadcast so far (factor transport; reset = 1 clears the counter).
gmrf_status gmrf_comm_bytes(gmrf_comm* c, int32_t reset, double* bytes) {
    if (!c || !bytes) return bad_shape("null pointer");
    *bytes = c->bytes_moved;
    if (reset) c->bytes_moved = 0.0;
    return GMRF_OK;
}

// The handle's stream waits for every transfer enqueued so far on the communicator's stream.
gmrf_status gmrf_comm_wait(gmrf_handle* h, gmrf_comm* c) {
    if (!h || !c) return bad_shape("null pointer");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(c->ev_out, c->stream));
    HIPCHK(hipStreamWaitEvent(h->stream, c->ev_out, 0));
    return GMRF_OK;
}

gmrf_status gmrf_bt_solve(gmrf_handle* h, const double* b, double* y, int64_t k, int64_t ldb, int64_t ldy, int32_t mode) {
    if (!h || !b || !y) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "solve before factor"; return GMRF_ERR_NO_FACTOR; }
    if (k <= 0 || ldb < h->n || ldy < h->n || mode < 0 || mode > 2) return bad_shape("bad k / ldb / ldy / mode");
    if (b == y && ldb != ldy) return bad_shape("in place (y == b) needs ldb == ldy");
    HIPCHK(hipSetDevice(h->device));
    if (h->B > 1 && k > KP_CHUNK) return bad_shape("with a batch of problems k is limited to 128 per call");
    const bool b_dev = is_device_ptr(b), y_dev = is_device_ptr(y);
    h->stats.solve_ms = 0.0;
    const int64_t nb = h->B;            // b / y hold nb consecutive groups of k columns (problem-major)
    h->sweep_persist_hold = (b == y);   // (in place: a persistent sweep that gave up could not be repeated)
    h->stats.sweep_persist = 0;
    for (int64_t c0 = 0; c0 < k; c0 += KP_CHUNK) {
        const int kc = (int)std::min<int64_t>(KP_CHUNK, k - c0);
        const int kp = pad_k(kc);
        GCHK(ensure_panels(h, kp));
        const double* bsrc = b + c0 * ldb;
        double* ydst = y + c0 * ldy;
        const double* d_b = bsrc;
        if (!b_dev || !y_dev) GCHK(ensure_stage(h, (int64_t)kc * h->n * nb));
        if (!b_dev) {
            HIPCHK(hipMemcpy2DAsync(h->d_stage, h->n * sizeof(double), bsrc, ldb * sizeof(double),
                                    h->n * sizeof(double), kc * nb, hipMemcpyHostToDevice, h->stream));
            d_b = h->d_stage;
        }
        GCHK(launch_pack(h, d_b, b_dev ? ldb : h->n, kc, kp));
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        GCHK(run_sweeps(h, mode, kp));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        const double* result = (mode == GMRF_SOLVE_FULL) ? h->d_P : h->d_Y;
        if (y_dev) {
            GCHK(launch_unpack(h, result, ydst, ldy, kc, nullptr));
            HIPCHK(hipStreamSynchronize(h->stream));
        } else {
            GCHK(launch_unpack(h, result, h->d_stage, h->n, kc, nullptr));
            HIPCHK(hipMemcpy2DAsync(ydst, ldy * sizeof(double), h->d_stage, h->n * sizeof(double),
                                    h->n * sizeof(double), kc * nb, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        bool repeat = false;
        GCHK(sweep_persist_check(h, &repeat));
        if (repeat) { c0 -= KP_CHUNK; continue; }           // (this chunk again, with a launch per product: b is untouched)
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
        h->stats.solve_ms += ms;
        const int nsweeps = (mode == GMRF_SOLVE_FULL) ? 2 : 1;
        h->stats.sweep_ms = ms / nsweeps;
        h->stats.sweep_bytes = sweep_bytes(h, kc) * (double)nb;
        h->stats.sweep_bytes_streamed = (8.0 * ((double)h->N * 0.5 * (double)h->bsp * (double)(h->bsp + 1) +
                                                (double)(h->N - 1) * h->c_streamed) + 16.0 * (double)h->n_pad * kc) * (double)nb;
        if (h->profiling) prof_collect(h);
    }
    return GMRF_OK;
}

static gmrf_status stage_vector(gmrf_handle* h, const double* v, double** d_buf, const double** d_out) {
    if (!v) { *d_out = nullptr; return GMRF_OK; }
    if (is_device_ptr(v)) { *d_out = v; return GMRF_OK; }
    if (!*d_buf) HIPCHK(hipMalloc(d_buf, sizeof(double) * h->n * h->B));
    HIPCHK(hipMemcpyAsync(*d_buf, v, sizeof(double) * h->n * h->B, hipMemcpyHostToDevice, h->stream));
    *d_out = *d_buf;
    return GMRF_OK;
}

// draws (or loads) z for samples [first_id + c0, +kc) into panel P and runs the backward sweep -> panel Y
static gmrf_status sample_chunk(gmrf_handle* h, uint64_t seed, int64_t first_id, int kc, const double* z,
                                int64_t ldz, int64_t id_stride) {
    const int kp = pad_k(kc);
    GCHK(ensure_panels(h, kp));
    if (z) {
        const double* d_z = z;
        int64_t ldd = ldz;
        if (!is_device_ptr(z)) {
            GCHK(ensure_stage(h, (int64_t)kc * h->n * h->B));
            HIPCHK(hipMemcpy2DAsync(h->d_stage, h->n * sizeof(double), z, ldz * sizeof(double),
                                    h->n * sizeof(double), kc * h->B, hipMemcpyHostToDevice, h->stream));
            d_z = h->d_stage; ldd = h->n;
        }
        GCHK(launch_pack(h, d_z, ldd, kc, kp));
    } else {
        const int64_t total = (int64_t)kp * h->n_pad;
        hipLaunchKernelGGL(fill_normals_panel, dim3((unsigned)((total + 255) / 256), (unsigned)h->B), dim3(256), 0,
                           h->stream, h->d_P, h->n_pad, (int)h->bs, (int)h->bsp, kc, kp, seed, first_id, id_stride);
        HIPCHK(hipGetLastError());
    }
    return run_sweeps(h, GMRF_SOLVE_BACKWARD, kp);
}

gmrf_status gmrf_bt_sample(gmrf_handle* h, uint64_t seed, int64_t first_id, int64_t k, const double* mean,
                           const double* z, double* out, int64_t ld) {
    if (!h || !out) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "sample before factor"; return GMRF_ERR_NO_FACTOR; }
    if (k <= 0 || ld < h->n) return bad_shape("bad k / ld");
    if (h->B > 1 && k > KP_CHUNK) return bad_shape("with a batch of problems k is limited to 128 per call");
    HIPCHK(hipSetDevice(h->device));
    const double* d_mean = nullptr;
    GCHK(stage_vector(h, mean, &h->d_mean, &d_mean));
    const bool out_dev = is_device_ptr(out);
    h->sweep_persist_hold = (z == out);
    h->stats.sweep_persist = 0;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int64_t c0 = 0; c0 < k; c0 += KP_CHUNK) {
        const int kc = (int)std::min<int64_t>(KP_CHUNK, k - c0);
        GCHK(sample_chunk(h, seed, first_id + c0, kc, z ? z + c0 * ld : nullptr, ld, k));
        if (out_dev) {
            GCHK(launch_unpack(h, h->d_Y, out + c0 * ld, ld, kc, d_mean));
        } else {
            GCHK(ensure_stage(h, (int64_t)kc * h->n * h->B));
            GCHK(launch_unpack(h, h->d_Y, h->d_stage, h->n, kc, d_mean));
            HIPCHK(hipMemcpy2DAsync(out + c0 * ld, ld * sizeof(double), h->d_stage, h->n * sizeof(double),
                                    h->n * sizeof(double), kc * h->B, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    bool repeat = false;
    GCHK(sweep_persist_check(h, &repeat));
    if (repeat) return gmrf_bt_sample(h, seed, first_id, k, mean, z, out, ld);     // (once: the handle has left the persistent form)
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    h->stats.sample_ms = ms;
    if (h->profiling) prof_collect(h);
    return GMRF_OK;
}

// mean = A^-1 b and k samples mean + L^-T z in ONE call (scripts/darcy/solve_darcy_gmrf-fem.jl:190-191 calls `mean` and `rand` on one
// factor, one after the other).  The samples' backward sweep needs nothing of the mean -- only their last step, adding it, does --
// so where the sweeps are persistent launches (one problem, blocks of 512 .. 1024) it runs BESIDE the mean's two sweeps, on a second
// stream with panels of its own: both are bound by their chains of hand-offs, not by the chip, and two resident workgroups per
// CU (97 + 189 VGPRs, 13 + 33 KB of LDS) take turns in the same time one takes alone.  The same kernels and the same sums as
// gmrf_bt_solve + gmrf_bt_sample: bitwise their results.  Anywhere else the call IS those two calls.
gmrf_status gmrf_bt_posterior(gmrf_handle* h, const double* b, uint64_t seed, int64_t first_id, int64_t k, double* mean,
                              double* samples, int64_t ld) {
    if (!h || !b || !mean || !samples) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "posterior before factor"; return GMRF_ERR_NO_FACTOR; }
    if (k <= 0 || ld < h->n) return bad_shape("bad k / ld");
    HIPCHK(hipSetDevice(h->device));
    const int kp = pad_k(k);
    const bool dev_all = is_device_ptr(b) && is_device_ptr(mean) && is_device_ptr(samples);
    const bool beside = h->B == 1 && k <= KP_CHUNK && k >= 2 && dev_all && b != mean && !h->profiling && sweep_persist_ok(h, 1) && sweep_persist_ok(h, kp);
    if (!beside) {
        GCHK(gmrf_bt_solve(h, b, mean, 1, h->n, h->n, GMRF_SOLVE_FULL));
        return gmrf_bt_sample(h, seed, first_id, k, mean, nullptr, samples, ld);
    }
    GCHK(ensure_panels(h, 1));
    GCHK(sweep_persist_prepare(h, 1));
    const int64_t elems = (int64_t)kp * h->n_pad;
    if (!h->d_P2 || h->p2_elems < elems) {
        HIPCHK(hipStreamSynchronize(h->stream));
        free_dev(h->d_P2); free_dev(h->d_Y2); free_dev(h->d_T2);
        h->d_P2 = h->d_Y2 = h->d_T2 = nullptr; h->p2_elems = 0;
        HIPCHK(hipMalloc(&h->d_P2, sizeof(double) * (size_t)elems));
        HIPCHK(hipMalloc(&h->d_Y2, sizeof(double) * (size_t)elems));
        HIPCHK(hipMalloc(&h->d_T2, sizeof(double) * (size_t)elems));
        h->p2_elems = elems;
    }
    if (!h->aux_distinct) {
        // the second stream must sit on a hardware queue of its own (two streams on one queue serialise: gmrf_streams_create):
        // candidates are timed against this handle's stream with the 1 ms spin kernel, the first that overlaps is kept
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); h->aux = nullptr; }
        const unsigned long long ticks = 100000ull;
        double one = 1e30;
        for (int r = 0; r < 2; ++r) one = std::min(one, spin_pair_ms(h->stream, nullptr, ticks));
        std::vector<hipStream_t> cand;
        for (int c = 0; c < 12 && !h->aux; ++c) {
            hipStream_t st = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            hipLaunchKernelGGL(gmrf_spin_kernel, dim3(1), dim3(64), 0, st, 1ull);      // (first use binds the stream to its queue)
            (void)hipStreamSynchronize(st);
            const double t = std::min(spin_pair_ms(h->stream, st, ticks), spin_pair_ms(h->stream, st, ticks));
            if (t < 1.5 * one) h->aux = st; else cand.push_back(st);
        }
        if (!h->aux && !cand.empty()) { h->aux = cand.back(); cand.pop_back(); }       // (none overlaps: any will do, serialised)
        for (hipStream_t st : cand) (void)hipStreamDestroy(st);
        if (!h->aux) return bad_shape("internal: no second stream");
        h->aux_distinct = true;
    }
    h->fork_next = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    GCHK(fork_event(h, &ev_fork)); GCHK(fork_event(h, &ev_join));
    h->sweep_persist_launched = true;
    h->stats.sweep_persist = 1;
    h->stats.sweep_persist_launches += 3;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(hipEventRecord(ev_fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->aux, ev_fork, 0));
    // second stream: z -> P2, backward sweep P2 -> Y2
    {
        const int64_t total = (int64_t)kp * h->n_pad;
        hipLaunchKernelGGL(fill_normals_panel, dim3((unsigned)((total + 255) / 256), 1), dim3(256), 0, h->aux, h->d_P2, h->n_pad,
                           (int)h->bs, (int)h->bsp, (int)k, kp, seed, first_id, k);
        HIPCHK(hipGetLastError());
        GCHK(launch_sweep_persist(h, true, kp, h->d_P2, h->d_Y2, h->aux, h->d_T2));
        HIPCHK(hipEventRecord(ev_join, h->aux));
    }
    // this stream: b -> P, forward P -> Y, backward Y -> P, mean out
    GCHK(launch_pack(h, b, h->n, 1, 1));
    GCHK(launch_sweep_persist(h, false, 1, h->d_P, h->d_Y));
    GCHK(launch_sweep_persist(h, true, 1, h->d_Y, h->d_P));
    GCHK(launch_unpack(h, h->d_P, mean, h->n, 1, nullptr));
    HIPCHK(hipStreamWaitEvent(h->stream, ev_join, 0));
    GCHK(launch_unpack(h, h->d_Y2, samples, ld, (int)k, mean));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    bool repeat = false;
    GCHK(sweep_persist_check(h, &repeat));
    if (repeat) {                                      // (a persistent sweep gave up: the two calls, with a launch per product)
        HIPCHK(hipStreamSynchronize(h->aux));
        GCHK(gmrf_bt_solve(h, b, mean, 1, h->n, h->n, GMRF_SOLVE_FULL));
        return gmrf_bt_sample(h, seed, first_id, k, mean, nullptr, samples, ld);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
    h->stats.solve_ms = ms;                            // (mean and samples together)
    h->stats.sample_ms = 0.0;
    return GMRF_OK;
}

gmrf_status gmrf_bt_normals(gmrf_handle* h, uint64_t seed, int64_t first_id, int64_t k, double* z, int64_t ld) {
    if (!h || !z) return bad_shape("null pointer");
    if (h->n <= 0) return bad_shape("no shape set");
    if (k <= 0 || ld < h->n) return bad_shape("bad k / ld");
    if (h->B > 1 && k > KP_CHUNK) return bad_shape("with a batch of problems k is limited to 128 per call");
    HIPCHK(hipSetDevice(h->device));
    const bool z_dev = is_device_ptr(z);
    for (int64_t c0 = 0; c0 < k; c0 += KP_CHUNK) {
        const int kc = (int)std::min<int64_t>(KP_CHUNK, k - c0);
        const int kp = pad_k(kc);
        GCHK(ensure_panels(h, kp));
        const int64_t total = (int64_t)kp * h->n_pad;
        hipLaunchKernelGGL(fill_normals_panel, dim3((unsigned)((total + 255) / 256), (unsigned)h->B), dim3(256), 0,
                           h->stream, h->d_P, h->n_pad, (int)h->bs, (int)h->bsp, kc, kp, seed, first_id + c0, k);
        HIPCHK(hipGetLastError());
        if (z_dev) {
            GCHK(launch_unpack(h, h->d_P, z + c0 * ld, ld, kc, nullptr));
        } else {
            GCHK(ensure_stage(h, (int64_t)kc * h->n * h->B));
            GCHK(launch_unpack(h, h->d_P, h->d_stage, h->n, kc, nullptr));
            HIPCHK(hipMemcpy2DAsync(z + c0 * ld, ld * sizeof(double), h->d_stage, h->n * sizeof(double),
                                    h->n * sizeof(double), kc * h->B, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GMRF_OK;
}

gmrf_status gmrf_bt_logdet(gmrf_handle* h, double* out) {
    if (!h || !out) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "logdet before factor"; return GMRF_ERR_NO_FACTOR; }
    HIPCHK(hipSetDevice(h->device));
    if (h->keep_l && h->l_valid) {
        hipLaunchKernelGGL(logdet_blocks, dim3((unsigned)h->N, 1), dim3(256), 0, h->stream,
                           h->d_L + h->sel * stride_pL(h), blk_elems(h), h->bsp, (int)h->bs, h->d_logdet + h->sel * h->N,
                           (int64_t)0, (int64_t)0);
        HIPCHK(hipGetLastError());
    } else if (!h->logdet_valid) {
        // (a factor adopted as raw buffers without its L blocks: nothing ever filled d_logdet)
        g_last_error = "neither the L blocks nor the log-determinant parts of this factor are resident (adopted as raw buffers without L)";
        return GMRF_ERR_NO_FACTOR;
    }   // else: the factorisation (or the transport image) left every block's part in d_logdet
    std::vector<double> part((size_t)h->N);
    HIPCHK(hipMemcpyAsync(part.data(), h->d_logdet + h->sel * h->N, sizeof(double) * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (double v : part) s += v;
    *out = 2.0 * s;
    return GMRF_OK;
}

// --------------------------------------------------------------------------------- CSR / SpMM
gmrf_status gmrf_csr_create(int32_t device, void* stream, int64_t n_rows, int64_t n_cols, const int64_t* rowptr,
                            const int64_t* colidx, const double* vals, int32_t index_base, int32_t values_f32,
                            gmrf_csr** out) {
    if (!out || !rowptr || !colidx || !vals || n_rows <= 0 || n_cols <= 0) return bad_shape("bad CSR arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        (void)hipGetLastError();
        g_last_error = "no HIP device visible";
        return GMRF_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    const int64_t nnz = rowptr[n_rows] - index_base;
    if (nnz < 0 || rowptr[0] != index_base) return bad_shape("bad CSR row pointers");
    std::vector<int64_t> rp((size_t)n_rows + 1);
    for (int64_t i = 0; i <= n_rows; ++i) {
        rp[i] = rowptr[i] - index_base;
        if (i > 0 && rp[i] < rp[i - 1]) return bad_shape("CSR row pointers must be non-decreasing");
    }
    std::vector<int32_t> ci((size_t)nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t c = colidx[p] - index_base;
        if (c < 0 || c >= n_cols) return bad_shape("column index out of range");
        ci[p] = (int32_t)c;
    }
    gmrf_csr* m = new gmrf_csr();
    m->device = device;
    m->tiles_ok = true;
    for (int64_t r = 0; r < n_rows; r += SPMV_ROWS)
        if (rp[std::min(n_rows, r + SPMV_ROWS)] - rp[r] > SPMV_CAP) { m->tiles_ok = false; break; }
    m->tiles128_ok = true;
    for (int64_t r = 0; r < n_rows; r += 128)
        if (rp[std::min<int64_t>(n_rows, r + 128)] - rp[r] > 2 * SPMV_CAP) { m->tiles128_ok = false; break; }
    // from here on a failing HIP call releases what has been allocated so far
#define CSRCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            (void)gmrf_csr_destroy(m);                                                        \
            return GMRF_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)
    if (stream) { m->stream = (hipStream_t)stream; } else { CSRCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)); m->own_stream = true; }
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    CSRCHK(hipMalloc(&m->d_rowptr, sizeof(int64_t) * (n_rows + 1)));
    CSRCHK(hipMalloc(&m->d_colidx, sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
    CSRCHK(hipMemcpyAsync(m->d_rowptr, rp.data(), sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice, m->stream));
    CSRCHK(hipMemcpyAsync(m->d_colidx, ci.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice, m->stream));
    CSRCHK(hipStreamSynchronize(m->stream));
    if (values_f32) {
        std::vector<float> vf((size_t)nnz);
        for (int64_t p = 0; p < nnz; ++p) vf[p] = (float)vals[p];
        CSRCHK(hipMalloc(&m->d_vals32, sizeof(float) * std::max<int64_t>(nnz, 1)));
        CSRCHK(hipMemcpyAsync(m->d_vals32, vf.data(), sizeof(float) * nnz, hipMemcpyHostToDevice, m->stream));
        CSRCHK(hipStreamSynchronize(m->stream));
    } else {
        CSRCHK(hipMalloc(&m->d_vals, sizeof(double) * std::max<int64_t>(nnz, 1)));
        CSRCHK(hipMemcpyAsync(m->d_vals, vals, sizeof(double) * nnz, hipMemcpyHostToDevice, m->stream));
        CSRCHK(hipStreamSynchronize(m->stream));
    }
    if (n_rows == n_cols) {
        CSRCHK(hipMalloc(&m->d_diag, sizeof(double) * n_rows));
        hipLaunchKernelGGL(csr_extract_diag, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, m->stream,
                           m->d_rowptr, m->d_colidx, m->d_vals, m->d_vals32, n_rows, m->d_diag);
        CSRCHK(hipGetLastError());
        CSRCHK(hipStreamSynchronize(m->stream));
    }
#undef CSRCHK
    *out = m;
    return GMRF_OK;
}

gmrf_status gmrf_csr_destroy(gmrf_csr* m) {
    if (!m) return GMRF_OK;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    free_dev(m->d_rowptr); free_dev(m->d_colidx); free_dev(m->d_vals); free_dev(m->d_vals32);
    free_dev(m->d_diag); free_dev(m->d_stage_x); free_dev(m->d_stage_y);
    free_dev(m->d_tile_uptr); free_dev(m->d_ucols); free_dev(m->d_lidx);
    if (m->own_stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return GMRF_OK;
}

static gmrf_status spmm_device(const gmrf_csr* S, hipStream_t st, const double* d_X, int64_t ldx, double* d_Y,
                               int64_t ldy, int k, const double* vals_override = nullptr) {
    const double avg = (double)S->nnz / (double)S->n_rows;
    const int bl = 256;
    if (k == 1 && S->tiles_ok) {
        // SpMV: LDS-staged row tiles (with 3+ right-hand sides the lane-group kernel, which reuses the
        // entries for four of them, is faster than one pass per right-hand side)
        static const int rows_env = [] { const char* e = getenv("GMRF_SPMV_ROWS"); return e ? atoi(e) : SPMV_ROWS; }();   // tuning aid: 64 / 128
        const bool tall = rows_env == 128 && S->tiles128_ok;
        const int rows = tall ? 128 : SPMV_ROWS;
        static const int wg_cap = [] { const char* e = getenv("GMRF_SPMV_WGS"); return e ? atoi(e) : 256 * 8; }();   // tuning aid: resident workgroups
        const dim3 grid((unsigned)std::min<int64_t>((S->n_rows + rows - 1) / rows, wg_cap > 0 ? wg_cap : (1 << 30)));
#define GMRF_SPMV(VT, VP)                                                                                                   \
        do {                                                                                                                \
            if (tall) hipLaunchKernelGGL((csr_spmv_tiles<VT, 128>), grid, dim3(bl), 0, st, S->d_rowptr, S->d_colidx, VP, S->n_rows, d_X + r * ldx, d_Y + r * ldy); \
            else hipLaunchKernelGGL((csr_spmv_tiles<VT, SPMV_ROWS>), grid, dim3(bl), 0, st, S->d_rowptr, S->d_colidx, VP, S->n_rows, d_X + r * ldx, d_Y + r * ldy); \
        } while (0)
        for (int r = 0; r < k; ++r) {
            if (vals_override) GMRF_SPMV(double, vals_override);
            else if (S->d_vals32) GMRF_SPMV(float, S->d_vals32);
            else GMRF_SPMV(double, S->d_vals);
        }
#undef GMRF_SPMV
        HIPCHK(hipGetLastError());
        return GMRF_OK;
    }
    auto grid_for = [&](int G) { return dim3((unsigned)((S->n_rows * G + bl - 1) / bl)); };
#define SPMM_LAUNCH(VT, G, VP)                                                                               \
    hipLaunchKernelGGL((csr_spmm<VT, G>), grid_for(G), dim3(bl), 0, st, S->d_rowptr, S->d_colidx, VP,        \
                       S->n_rows, d_X, ldx, d_Y, ldy, k)
    if (vals_override) {                      // same pattern, another problem's values (fp64)
        if (avg > 40) SPMM_LAUNCH(double, 64, vals_override);
        else if (avg > 20) SPMM_LAUNCH(double, 32, vals_override);
        else if (avg > 10) SPMM_LAUNCH(double, 16, vals_override);
        else SPMM_LAUNCH(double, 8, vals_override);
    } else if (S->d_vals32) {
        if (avg > 40) SPMM_LAUNCH(float, 64, S->d_vals32);
        else if (avg > 20) SPMM_LAUNCH(float, 32, S->d_vals32);
        else if (avg > 10) SPMM_LAUNCH(float, 16, S->d_vals32);
        else SPMM_LAUNCH(float, 8, S->d_vals32);
    } else {
        if (avg > 40) SPMM_LAUNCH(double, 64, S->d_vals);
        else if (avg > 20) SPMM_LAUNCH(double, 32, S->d_vals);
        else if (avg > 10) SPMM_LAUNCH(double, 16, S->d_vals);
        else SPMM_LAUNCH(double, 8, S->d_vals);
    }
#undef SPMM_LAUNCH
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

// Tile plan of csr_spmm_tiles: per SPMM_ROWS-row tile the ascending list of distinct columns and, per
// entry, the 16-bit index of its column in that list.  Built once per matrix (host; the device
// arrays of the matrix are read back), like the symbolic phase of the factor.
static gmrf_status spmm_plan(gmrf_csr* m) {
    if (m->plan_state != 0) return GMRF_OK;
    std::vector<int64_t> rp((size_t)m->n_rows + 1);
    std::vector<int32_t> ci((size_t)std::max<int64_t>(m->nnz, 1));
    HIPCHK(hipMemcpyAsync(rp.data(), m->d_rowptr, sizeof(int64_t) * (m->n_rows + 1), hipMemcpyDeviceToHost, m->stream));
    if (m->nnz > 0) HIPCHK(hipMemcpyAsync(ci.data(), m->d_colidx, sizeof(int32_t) * m->nnz, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    std::vector<int64_t> uptr;
    std::vector<int32_t> ucols;
    std::vector<uint16_t> lidx((size_t)std::max<int64_t>(m->nnz, 1));
    std::vector<int32_t> tmp;
    // the tallest tile whose LDS image (distinct rows of X for 16 right-hand sides + the entries) lets three
    // workgroups share a CU; a matrix whose 16-row tiles do not fit keeps the plain kernel
    static const size_t lds_budget = [] { const char* e = getenv("GMRF_SPMM_LDS_KB"); return (size_t)(e ? atoi(e) : 52) * 1024; }();   // tuning aid
    m->plan_state = -1;
    for (int R : {64, 32, 16}) {
        const int64_t T = (m->n_rows + R - 1) / R;
        uptr.assign((size_t)T + 1, 0);
        ucols.clear();
        ucols.reserve((size_t)(m->nnz / 3 + 16));
        int64_t umax = 0, emax = 0, pmax = 0;
        bool ok = true;
        for (int64_t tix = 0; tix < T && ok; ++tix) {
            const int64_t r0 = tix * R, r1 = std::min(m->n_rows, r0 + R);
            const int64_t a = rp[r0], b = rp[r1];
            tmp.assign(ci.begin() + a, ci.begin() + b);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            umax = std::max<int64_t>(umax, (int64_t)tmp.size()); emax = std::max(emax, b - a);
            int64_t padded = 0;                                  // rows padded to multiples of 8 entries (csr_spmm_tiles_pad)
            for (int64_t r = r0; r < r1; ++r) padded += (rp[r + 1] - rp[r] + 7) / 8 * 8;
            pmax = std::max(pmax, padded);
            if ((int64_t)tmp.size() > 32 * SPMM_NG || b - a > 8192) { ok = false; break; }
            for (int64_t e = a; e < b; ++e)
                lidx[(size_t)e] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), ci[(size_t)e]) - tmp.begin());
            ucols.insert(ucols.end(), tmp.begin(), tmp.end());
            uptr[(size_t)tix + 1] = (int64_t)ucols.size();
        }
        if (!ok) continue;
        const int ucap = (int)std::max<int64_t>(32, (umax + 31) / 32 * 32), ecap = (int)std::max<int64_t>(256, (emax + 255) / 256 * 256);
        const int ecap_pad = (int)std::max<int64_t>(256, (pmax + 255) / 256 * 256);
        if (spmm_tile_lds_bytes(R, ucap, ecap) > lds_budget) continue;
        m->plan_rows = R; m->plan_ucap = ucap; m->plan_ecap = ecap; m->plan_umax = (int)umax;
        // (the padded kernel sizes its LDS image by the distinct columns rounded to 8, the entries to 64: 39.3 KB on the
        //  burgers4096x512 matrix -- four workgroups per CU)
        const int ecap_pad64 = (int)std::max<int64_t>(64, (pmax + 63) / 64 * 64);
        m->plan_ecap_pad = (spmm_tile_pad_lds_bytes(R, (int)((umax + 7) / 8 * 8), ecap_pad64) <= lds_budget + 1024) ? ecap_pad64 : 0;
        (void)ecap_pad;
        m->n_ucols = (int64_t)ucols.size();
        HIPCHK(hipMalloc(&m->d_tile_uptr, sizeof(int64_t) * (T + 1)));
        HIPCHK(hipMalloc(&m->d_ucols, sizeof(int32_t) * std::max<size_t>(ucols.size(), 1)));
        HIPCHK(hipMalloc(&m->d_lidx, sizeof(uint16_t) * std::max<int64_t>(m->nnz, 1)));
        HIPCHK(hipMemcpyAsync(m->d_tile_uptr, uptr.data(), sizeof(int64_t) * (T + 1), hipMemcpyHostToDevice, m->stream));
        if (!ucols.empty()) HIPCHK(hipMemcpyAsync(m->d_ucols, ucols.data(), sizeof(int32_t) * ucols.size(), hipMemcpyHostToDevice, m->stream));
        if (m->nnz > 0) HIPCHK(hipMemcpyAsync(m->d_lidx, lidx.data(), sizeof(uint16_t) * m->nnz, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        m->plan_state = 1;
        break;
    }
    return GMRF_OK;
}

// Y = S X with node-major device operands (X[col][rhs], Y[row][rhs]); vals_override: same pattern,
// another problem's fp64 values.
static gmrf_status spmm_rows_device(const gmrf_csr* S, hipStream_t st, const double* d_X, int64_t ldx, double* d_Y,
                                    int64_t ldy, int k, const double* vals_override = nullptr) {
    gmrf_csr* m = const_cast<gmrf_csr*>(S);
    GCHK(spmm_plan(m));
    const bool aligned = (k % 2 == 0) && (ldx % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)d_X) % 16 == 0) && (((uintptr_t)d_Y) % 16 == 0) &&
                         (S->n_cols * ldx < ((int64_t)1 << 32));
    if (m->plan_state == 1 && aligned) {
        const int R = m->plan_rows, uc = m->plan_ucap, ec = m->plan_ecap;
        const dim3 grid((unsigned)((S->n_rows + R - 1) / R));
        static const bool no_pad = [] { const char* e = getenv("GMRF_SPMM_PAD"); return e && atoi(e) == 0; }();   // tuning aid
        const bool pad = m->plan_ecap_pad > 0 && !no_pad;
        const int ecl = pad ? m->plan_ecap_pad : ec;
        const int ucl = pad ? (m->plan_umax + 7) / 8 * 8 : uc;          // LDS rows of the staged X image
        const size_t lds = pad ? spmm_tile_pad_lds_bytes(R, ucl, ecl) : spmm_tile_lds_bytes(R, uc, ec);
        // 7 gather loads per thread and chunk serve up to 224 distinct columns per tile, 10 up to 320
#define GMRF_SPMM_TILES(VT, VP)                                                                                          \
        do {                                                                                                             \
            if (pad && uc <= 224) hipLaunchKernelGGL((csr_spmm_tiles_pad<VT, 7>), grid, dim3(SPMM_THREADS), lds, st, S->d_rowptr, S->d_lidx, VP, \
                                              S->d_tile_uptr, S->d_ucols, S->n_rows, d_X, ldx, d_Y, ldy, k, R, ucl, ecl);  \
            else if (pad) hipLaunchKernelGGL((csr_spmm_tiles_pad<VT, SPMM_NG>), grid, dim3(SPMM_THREADS), lds, st, S->d_rowptr, S->d_lidx, VP,    \
                                    S->d_tile_uptr, S->d_ucols, S->n_rows, d_X, ldx, d_Y, ldy, k, R, ucl, ecl);            \
            else if (uc <= 224) hipLaunchKernelGGL((csr_spmm_tiles<VT, 7>), grid, dim3(SPMM_THREADS), lds, st, S->d_rowptr, S->d_lidx, VP, \
                                              S->d_tile_uptr, S->d_ucols, S->n_rows, d_X, ldx, d_Y, ldy, k, R, uc, ec);  \
            else hipLaunchKernelGGL((csr_spmm_tiles<VT, SPMM_NG>), grid, dim3(SPMM_THREADS), lds, st, S->d_rowptr, S->d_lidx, VP,    \
                                    S->d_tile_uptr, S->d_ucols, S->n_rows, d_X, ldx, d_Y, ldy, k, R, uc, ec);            \
        } while (0)
        if (vals_override) GMRF_SPMM_TILES(double, vals_override);
        else if (S->d_vals32) GMRF_SPMM_TILES(float, S->d_vals32);
        else GMRF_SPMM_TILES(double, S->d_vals);
#undef GMRF_SPMM_TILES
    } else {
        const dim3 grid((unsigned)((S->n_rows + 15) / 16));
        if (vals_override) hipLaunchKernelGGL(csr_spmm_rows<double>, grid, dim3(256), 0, st, S->d_rowptr, S->d_colidx, vals_override, S->n_rows, d_X, ldx, d_Y, ldy, k);
        else if (S->d_vals32) hipLaunchKernelGGL(csr_spmm_rows<float>, grid, dim3(256), 0, st, S->d_rowptr, S->d_colidx, S->d_vals32, S->n_rows, d_X, ldx, d_Y, ldy, k);
        else hipLaunchKernelGGL(csr_spmm_rows<double>, grid, dim3(256), 0, st, S->d_rowptr, S->d_colidx, S->d_vals, S->n_rows, d_X, ldx, d_Y, ldy, k);
    }
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

// Y = S X with NODE-MAJOR operands: X is n_cols x k with the k values of a column index contiguous
// (row stride ldx >= k), Y likewise (ldy >= k) -- in Julia terms the k x n matrices permutedims(X),
// permutedims(Y).  This is the layout the LDS-tiled kernel wants (one gathered column index fetches
// k contiguous doubles); host pointers are staged.
gmrf_status gmrf_spmm_rows(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy) {
    if (!S || !X || !Y) return bad_shape("null pointer");
    if (k <= 0 || k > (1 << 20) || ldx < k || ldy < k) return bad_shape("bad k / ld");
    HIPCHK(hipSetDevice(S->device));
    gmrf_csr* m = const_cast<gmrf_csr*>(S);
    const bool x_dev = is_device_ptr(X), y_dev = is_device_ptr(Y);
    const double* d_X = X;
    double* d_Y = Y;
    int64_t lx = ldx, ly = ldy;
    const int64_t kk = k + (k & 1);                       // staging keeps rows 16-byte aligned
    const int64_t need = kk * std::max(S->n_rows, S->n_cols);
    if ((!x_dev || !y_dev) && need > m->stage_cap) {
        free_dev(m->d_stage_x); free_dev(m->d_stage_y);
        m->d_stage_x = m->d_stage_y = nullptr;
        HIPCHK(hipMalloc(&m->d_stage_x, sizeof(double) * need));
        HIPCHK(hipMalloc(&m->d_stage_y, sizeof(double) * need));
        m->stage_cap = need;
    }
    if (!x_dev) {
        HIPCHK(hipMemcpy2DAsync(m->d_stage_x, kk * sizeof(double), X, ldx * sizeof(double), k * sizeof(double), S->n_cols,
                                hipMemcpyHostToDevice, S->stream));
        d_X = m->d_stage_x; lx = kk;
    }
    if (!y_dev) { d_Y = m->d_stage_y; ly = kk; }
    GCHK(spmm_rows_device(S, S->stream, d_X, lx, d_Y, ly, (int)k));
    if (!y_dev)
        HIPCHK(hipMemcpy2DAsync(Y, ldy * sizeof(double), d_Y, kk * sizeof(double), k * sizeof(double), S->n_rows,
                                hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    return GMRF_OK;
}

gmrf_status gmrf_spmm(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy) {
    if (!S || !X || !Y) return bad_shape("null pointer");
    if (k <= 0 || ldx < S->n_cols || ldy < S->n_rows) return bad_shape("bad k / ld");
    HIPCHK(hipSetDevice(S->device));
    gmrf_csr* m = const_cast<gmrf_csr*>(S);
    const bool x_dev = is_device_ptr(X), y_dev = is_device_ptr(Y);
    const double* d_X = X;
    double* d_Y = Y;
    int64_t lx = ldx, ly = ldy;
    const int64_t need = k * std::max(S->n_rows, S->n_cols);
    if ((!x_dev || !y_dev) && need > m->stage_cap) {
        free_dev(m->d_stage_x); free_dev(m->d_stage_y);
        m->d_stage_x = m->d_stage_y = nullptr;
        HIPCHK(hipMalloc(&m->d_stage_x, sizeof(double) * need));
        HIPCHK(hipMalloc(&m->d_stage_y, sizeof(double) * need));
        m->stage_cap = need;
    }
    if (!x_dev) {
        HIPCHK(hipMemcpy2DAsync(m->d_stage_x, S->n_cols * sizeof(double), X, ldx * sizeof(double),
                                S->n_cols * sizeof(double), k, hipMemcpyHostToDevice, S->stream));
        d_X = m->d_stage_x; lx = S->n_cols;
    }
    if (!y_dev) { d_Y = m->d_stage_y; ly = S->n_rows; }
    GCHK(spmm_device(S, S->stream, d_X, lx, d_Y, ly, (int)k));
    if (!y_dev)
        HIPCHK(hipMemcpy2DAsync(Y, ldy * sizeof(double), d_Y, S->n_rows * sizeof(double),
                                S->n_rows * sizeof(double), k, hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    return GMRF_OK;
}

// Stream-ordered variants for device-resident operands: no staging, no synchronisation -- the product is enqueued on
// the matrix's stream and the call returns (pipelines of `Q * x` on the device: Gauss-Newton loops, estimators).
gmrf_status gmrf_spmm_async(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy) {
    if (!S || !X || !Y) return bad_shape("null pointer");
    if (k <= 0 || ldx < S->n_cols || ldy < S->n_rows) return bad_shape("bad k / ld");
    HIPCHK(hipSetDevice(S->device));
    if (!is_device_ptr(X) || !is_device_ptr(Y)) return bad_shape("gmrf_spmm_async takes device pointers");
    return spmm_device(S, S->stream, X, ldx, Y, ldy, (int)k);
}

gmrf_status gmrf_spmm_rows_async(const gmrf_csr* S, const double* X, double* Y, int64_t k, int64_t ldx, int64_t ldy) {
    if (!S || !X || !Y) return bad_shape("null pointer");
    if (k <= 0 || k > (1 << 20) || ldx < k || ldy < k) return bad_shape("bad k / ld");
    HIPCHK(hipSetDevice(S->device));
    if (!is_device_ptr(X) || !is_device_ptr(Y)) return bad_shape("gmrf_spmm_rows_async takes device pointers");
    return spmm_rows_device(S, S->stream, X, ldx, Y, ldy, (int)k);
}

// --------------------------------------------------------------------------------- posterior assembly
struct gmrf_assembler {
    int device = -1;                    // -1: symbolic only (pattern queries; no numeric phase)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0, m = 0, nnz_q = 0, nnz_j = 0, nnz_out = 0, n_pairs = 0;
    std::vector<int64_t> colptr, rowval;            // result pattern (CSC, 0-based, rows ascending)
    int64_t *d_pptr = nullptr, *d_qmap = nullptr, *d_jt_ptr = nullptr, *d_j_rptr = nullptr;
    int32_t *d_pa = nullptr, *d_pb = nullptr, *d_jt_row = nullptr, *d_jt_src = nullptr, *d_j_col = nullptr;
    double *d_q = nullptr, *d_jv = nullptr, *d_vn = nullptr, *d_vm = nullptr, *d_vn2 = nullptr, *d_out = nullptr;   // staging for host callers
};

gmrf_status gmrf_assemble_create(int32_t device, void* stream, int64_t n, const int64_t* q_colptr, const int64_t* q_rowval,
                                 int64_t m, const int64_t* j_rowptr, const int64_t* j_colidx, int32_t index_base,
                                 gmrf_assembler** out) {
    if (!out || !q_colptr || !q_rowval || !j_rowptr || !j_colidx || n <= 0 || m <= 0) return bad_shape("bad assembler arguments");
    if (index_base != 0 && index_base != 1) return bad_shape("index_base must be 0 or 1");
    const int64_t b = index_base;
    const int64_t nnz_q = q_colptr[n] - b, nnz_j = j_rowptr[m] - b;
    if (nnz_q < 0 || nnz_j < 0 || nnz_j >= ((int64_t)1 << 31)) return bad_shape("bad pattern sizes");
    // J by columns: for column j the (row k, position in the CSR value array)
    std::vector<int64_t> jt_ptr((size_t)n + 1, 0);
    for (int64_t p = 0; p < nnz_j; ++p) {
        const int64_t c = j_colidx[p] - b;
        if (c < 0 || c >= n) return bad_shape("J column index out of range");
        jt_ptr[c + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) jt_ptr[i + 1] += jt_ptr[i];
    std::vector<int32_t> jt_row((size_t)nnz_j), jt_src((size_t)nnz_j), j_col((size_t)nnz_j);
    {
        std::vector<int64_t> fill(jt_ptr.begin(), jt_ptr.end() - 1);
        for (int64_t k = 0; k < m; ++k)
            for (int64_t p = j_rowptr[k] - b; p < j_rowptr[k + 1] - b; ++p) {
                const int64_t c = j_colidx[p] - b;
                jt_row[fill[c]] = (int32_t)k; jt_src[fill[c]] = (int32_t)p; fill[c]++;
                j_col[p] = (int32_t)c;
            }
    }
    // result column j = rows of Q's column j  U  { i : J[k,i] != 0 and J[k,j] != 0 for some k }
    struct Prod { int64_t i; int32_t k, a, b; };
    auto* as = new gmrf_assembler();
    as->n = n; as->m = m; as->nnz_q = nnz_q; as->nnz_j = nnz_j;
    as->colptr.assign((size_t)n + 1, 0);
    std::vector<int64_t> pptr(1, 0), qmap;
    std::vector<int32_t> pa, pb;
    std::vector<Prod> prods;
    for (int64_t j = 0; j < n; ++j) {
        prods.clear();
        for (int64_t t = jt_ptr[j]; t < jt_ptr[j + 1]; ++t) {
            const int64_t k = jt_row[t];
            for (int64_t p = j_rowptr[k] - b; p < j_rowptr[k + 1] - b; ++p)
                prods.push_back({j_colidx[p] - b, (int32_t)k, (int32_t)p, jt_src[t]});
        }
        std::sort(prods.begin(), prods.end(), [](const Prod& x, const Prod& y) { return x.i != y.i ? x.i < y.i : x.k < y.k; });
        size_t pi = 0;
        int64_t qp = q_colptr[j] - b;
        const int64_t qe = q_colptr[j + 1] - b;
        int64_t last_row = -1;
        while (pi < prods.size() || qp < qe) {
            const int64_t rq = qp < qe ? q_rowval[qp] - b : INT64_MAX, rp = pi < prods.size() ? prods[pi].i : INT64_MAX;
            const int64_t r = std::min(rq, rp);
            if (r < 0 || r >= n || r <= last_row) { delete as; return bad_shape("Q rows must be ascending within a column and in range"); }
            last_row = r;
            as->rowval.push_back(r);
            qmap.push_back(rq == r ? qp++ : -1);
            while (pi < prods.size() && prods[pi].i == r) { pa.push_back(prods[pi].a); pb.push_back(prods[pi].b); ++pi; }
            pptr.push_back((int64_t)pa.size());
        }
        as->colptr[j + 1] = (int64_t)as->rowval.size();
    }
    as->nnz_out = (int64_t)as->rowval.size();
    as->n_pairs = (int64_t)pa.size();
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
            delete as;
            g_last_error = "no HIP device visible (libgmrf_hip needs an MI355X / gfx950 GPU)";
            return GMRF_ERR_NO_DEVICE;
        }
        as->device = device;
        HIPCHK(hipSetDevice(device));
        if (stream) as->stream = (hipStream_t)stream;
        else { HIPCHK(hipStreamCreate(&as->stream)); as->own_stream = true; }
        auto up = [&](auto** d, const auto& v) -> hipError_t {
            using T = typename std::remove_reference<decltype(v[0])>::type;
            hipError_t e = hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
            if (e == hipSuccess && !v.empty()) e = hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, as->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(as->stream);
            return e;
        };
        std::vector<int64_t> j_rptr((size_t)m + 1);
        for (int64_t k = 0; k <= m; ++k) j_rptr[k] = j_rowptr[k] - b;
        HIPCHK(up(&as->d_pptr, pptr)); HIPCHK(up(&as->d_qmap, qmap)); HIPCHK(up(&as->d_pa, pa)); HIPCHK(up(&as->d_pb, pb));
        HIPCHK(up(&as->d_jt_ptr, jt_ptr)); HIPCHK(up(&as->d_jt_row, jt_row)); HIPCHK(up(&as->d_jt_src, jt_src));
        HIPCHK(up(&as->d_j_rptr, j_rptr)); HIPCHK(up(&as->d_j_col, j_col));
    }
    *out = as;
    return GMRF_OK;
}

gmrf_status gmrf_assemble_destroy(gmrf_assembler* as) {
    if (!as) return GMRF_OK;
    if (as->device >= 0) {
        (void)hipSetDevice(as->device);
        (void)hipStreamSynchronize(as->stream);
        free_dev(as->d_pptr); free_dev(as->d_qmap); free_dev(as->d_pa); free_dev(as->d_pb);
        free_dev(as->d_jt_ptr); free_dev(as->d_jt_row); free_dev(as->d_jt_src); free_dev(as->d_j_rptr); free_dev(as->d_j_col);
        free_dev(as->d_q); free_dev(as->d_jv); free_dev(as->d_vn); free_dev(as->d_vm); free_dev(as->d_vn2); free_dev(as->d_out);
        if (as->own_stream) (void)hipStreamDestroy(as->stream);
    }
    delete as;
    return GMRF_OK;
}

gmrf_status gmrf_assemble_pattern(const gmrf_assembler* as, int64_t* nnz_out, int64_t* n_products, int64_t* colptr,
                                  int64_t* rowval, int32_t index_base) {
    if (!as) return bad_shape("null assembler");
    if (nnz_out) *nnz_out = as->nnz_out;
    if (n_products) *n_products = as->n_pairs;
    if (colptr) for (int64_t j = 0; j <= as->n; ++j) colptr[j] = as->colptr[j] + index_base;
    if (rowval) for (int64_t e = 0; e < as->nnz_out; ++e) rowval[e] = as->rowval[e] + index_base;
    return GMRF_OK;
}

// host pointer -> staged device copy (lazily allocated buffer of `cap` doubles)
static gmrf_status as_stage_in(gmrf_assembler* as, const double* p, int64_t count, double** buf, const double** d) {
    if (!p) { *d = nullptr; return GMRF_OK; }
    if (is_device_ptr(p)) { *d = p; return GMRF_OK; }
    if (!*buf) HIPCHK(hipMalloc(buf, sizeof(double) * std::max<int64_t>(count, 1)));
    HIPCHK(hipMemcpyAsync(*buf, p, sizeof(double) * count, hipMemcpyHostToDevice, as->stream));
    *d = *buf;
    return GMRF_OK;
}

static gmrf_status as_numeric_ready(gmrf_assembler* as) {
    if (!as) return bad_shape("null assembler");
    if (as->device < 0) { g_last_error = "symbolic-only assembler (created with device -1)"; return GMRF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(as->device));
    return GMRF_OK;
}

gmrf_status gmrf_assemble_precision(gmrf_assembler* as, const double* q_nzval, const double* j_vals, double noise,
                                    double* out_nzval) {
    GCHK(as_numeric_ready(as));
    if (!q_nzval || !j_vals || !out_nzval) return bad_shape("null pointer");
    const double *dq, *dj;
    GCHK(as_stage_in(as, q_nzval, as->nnz_q, &as->d_q, &dq));
    GCHK(as_stage_in(as, j_vals, as->nnz_j, &as->d_jv, &dj));
    const bool dev = is_device_ptr(out_nzval);
    double* d_out = out_nzval;
    if (!dev) {
        if (!as->d_out) HIPCHK(hipMalloc(&as->d_out, sizeof(double) * std::max<int64_t>(as->nnz_out, 1)));
        d_out = as->d_out;
    }
    hipLaunchKernelGGL(assemble_precision, dim3((unsigned)((as->nnz_out + 255) / 256)), dim3(256), 0, as->stream, as->d_pptr,
                       as->d_pa, as->d_pb, as->d_qmap, dq, dj, noise, as->nnz_out, d_out);
    HIPCHK(hipGetLastError());
    if (!dev) HIPCHK(hipMemcpyAsync(out_nzval, d_out, sizeof(double) * as->nnz_out, hipMemcpyDeviceToHost, as->stream));
    HIPCHK(hipStreamSynchronize(as->stream));
    return GMRF_OK;
}

gmrf_status gmrf_assemble_rhs(gmrf_assembler* as, const double* base, const double* j_vals, const double* x,
                              const double* obs_diff, double noise, double* out) {
    GCHK(as_numeric_ready(as));
    if (!j_vals || !x || !out) return bad_shape("null pointer");
    const double *dj, *dx, *dadd, *dbase;
    GCHK(as_stage_in(as, j_vals, as->nnz_j, &as->d_jv, &dj));
    GCHK(as_stage_in(as, x, as->n, &as->d_vn, &dx));
    if (!as->d_vm) HIPCHK(hipMalloc(&as->d_vm, sizeof(double) * 2 * as->m));
    dadd = obs_diff;
    if (obs_diff && !is_device_ptr(obs_diff)) {
        HIPCHK(hipMemcpyAsync(as->d_vm + as->m, obs_diff, sizeof(double) * as->m, hipMemcpyHostToDevice, as->stream));
        dadd = as->d_vm + as->m;
    }
    GCHK(as_stage_in(as, base, as->n, &as->d_vn2, &dbase));
    // v = J x + obs_diff ;  out = base + noise * J' v
    hipLaunchKernelGGL(assemble_j_apply, dim3((unsigned)((as->m + 255) / 256)), dim3(256), 0, as->stream, as->d_j_rptr,
                       as->d_j_col, dj, dx, dadd, as->m, as->d_vm);
    const bool dev = is_device_ptr(out);
    double* d_out = out;
    if (!dev) {
        if (!as->d_out) HIPCHK(hipMalloc(&as->d_out, sizeof(double) * std::max<int64_t>(std::max(as->nnz_out, as->n), 1)));
        d_out = as->d_out;
    }
    hipLaunchKernelGGL(assemble_jt_apply, dim3((unsigned)((as->n + 255) / 256)), dim3(256), 0, as->stream, as->d_jt_ptr,
                       as->d_jt_row, as->d_jt_src, dj, as->d_vm, dbase, noise, as->n, d_out);
    HIPCHK(hipGetLastError());
    if (!dev) HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * as->n, hipMemcpyDeviceToHost, as->stream));
    HIPCHK(hipStreamSynchronize(as->stream));
    return GMRF_OK;
}

// --------------------------------------------------------------------------------- FEM block assembly (Darcy, P1)
struct gmrf_darcy_p1 {
    int device = -1;                    // -1: pattern only
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t nx = 0, ny = 0, n = 0, nnz = 0;
    int order = 1;                      // 1: P1 triangles on the nx x ny nodes; 2: quadratic triangles, dofs = the (2 nx - 1) x (2 ny - 1) lattice
    std::vector<int64_t> rowptr, colidx;            // 0-based
    int64_t* d_rowptr = nullptr;
    int32_t* d_colidx = nullptr;        // order 2: the pattern's columns and the Dirichlet mask for the generic apply! kernels
    uint8_t* d_pres = nullptr;
    double *d_diag = nullptr, *d_mean = nullptr, *d_table = nullptr, *d_vals = nullptr, *d_f = nullptr;   // work + staging
    int64_t table_cap = 0;
};

// Pattern of the quadratic-triangle lattice: row (I, J) couples with every node of every cell it belongs to (the same
// enumeration as darcy_p2_rows: 5 x 5 window of lattice offsets, touched entries in ascending column order).
static void darcy_p2_pattern(gmrf_darcy_p1* d) {
    const int64_t nx = d->nx, ny = d->ny, W = 2 * nx - 1, H = 2 * ny - 1;
    d->n = W * H;
    d->rowptr.assign((size_t)d->n + 1, 0);
    d->colidx.reserve((size_t)d->n * 12);
    for (int64_t J = 0; J < H; ++J)
        for (int64_t I = 0; I < W; ++I) {
            unsigned present = 0u;
            int cand[6][3], nc;
            if (!(I & 1) && !(J & 1)) { const int t[6][3] = {{-1, -1, 0}, {-1, 0, 0}, {0, 0, 0}, {-1, -1, 1}, {0, -1, 1}, {0, 0, 1}}; nc = 6; memcpy(cand, t, sizeof(t)); }
            else if ((I & 1) && !(J & 1)) { const int t[2][3] = {{0, 0, 0}, {0, -1, 1}}; nc = 2; memcpy(cand, t, sizeof(t)); }
            else if (!(I & 1) && (J & 1)) { const int t[2][3] = {{-1, 0, 0}, {0, 0, 1}}; nc = 2; memcpy(cand, t, sizeof(t)); }
            else { const int t[2][3] = {{0, 0, 0}, {0, 0, 1}}; nc = 2; memcpy(cand, t, sizeof(t)); }
            for (int e = 0; e < nc; ++e) {
                const int64_t qx = I / 2 + cand[e][0], qy = J / 2 + cand[e][1];
                if (qx < 0 || qy < 0 || qx >= nx - 1 || qy >= ny - 1) continue;
                const bool upper = cand[e][2] != 0;
                const int64_t I0 = 2 * qx, J0 = 2 * qy;
                int64_t nI[6], nJ[6];
                nI[0] = I0; nJ[0] = J0;
                if (!upper) { nI[1] = I0 + 2; nJ[1] = J0; nI[2] = I0 + 2; nJ[2] = J0 + 2; }
                else { nI[1] = I0 + 2; nJ[1] = J0 + 2; nI[2] = I0; nJ[2] = J0 + 2; }
                nI[3] = (nI[0] + nI[1]) / 2; nJ[3] = (nJ[0] + nJ[1]) / 2;
                nI[4] = (nI[1] + nI[2]) / 2; nJ[4] = (nJ[1] + nJ[2]) / 2;
                nI[5] = (nI[2] + nI[0]) / 2; nJ[5] = (nJ[2] + nJ[0]) / 2;
                for (int j = 0; j < 6; ++j) present |= 1u << (unsigned)((nJ[j] - J + 2) * 5 + (nI[j] - I + 2));
            }
            for (int s5 = 0; s5 < 25; ++s5)
                if (present & (1u << s5)) d->colidx.push_back((J + s5 / 5 - 2) * W + (I + s5 % 5 - 2));
            d->rowptr[(size_t)(J * W + I) + 1] = (int64_t)d->colidx.size();
        }
}

static gmrf_status darcy_create(int32_t device, void* stream, int64_t nx, int64_t ny, int order, gmrf_darcy_p1** out) {
    if (!out || nx < 2 || ny < 2 || nx > 32768 || ny > 32768 || (order == 2 && (nx > 16384 || ny > 16384))) return bad_shape("bad Darcy mesh size");
    auto* d = new gmrf_darcy_p1();
    d->nx = nx; d->ny = ny; d->n = nx * ny; d->order = order;
    if (order == 2) darcy_p2_pattern(d);
    else {
    d->rowptr.assign((size_t)d->n + 1, 0);
    d->colidx.reserve((size_t)d->n * 7);
    const int dxs[7] = {-1, 0, -1, 0, 1, 0, 1}, dys[7] = {-1, -1, 0, 0, 0, 1, 1};
    for (int64_t iy = 0; iy < ny; ++iy)
        for (int64_t ix = 0; ix < nx; ++ix) {
            for (int s7 = 0; s7 < 7; ++s7) {
                const int64_t jx = ix + dxs[s7], jy = iy + dys[s7];
                if (jx < 0 || jy < 0 || jx >= nx || jy >= ny) continue;
                d->colidx.push_back(jy * nx + jx);
            }
            d->rowptr[(size_t)(iy * nx + ix) + 1] = (int64_t)d->colidx.size();
        }
    }
    d->nnz = (int64_t)d->colidx.size();
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
            (void)hipGetLastError();
            delete d;
            g_last_error = "no HIP device visible (libgmrf_hip needs an MI355X / gfx950 GPU)";
            return GMRF_ERR_NO_DEVICE;
        }
        d->device = device;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) { if (stream) d->stream = (hipStream_t)stream; else { e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking); d->own_stream = (e == hipSuccess); } }
        if (e == hipSuccess) e = hipMalloc(&d->d_rowptr, sizeof(int64_t) * (d->n + 1));
        if (e == hipSuccess) e = hipMalloc(&d->d_diag, sizeof(double) * d->n);
        if (e == hipSuccess) e = hipMalloc(&d->d_mean, sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(d->d_rowptr, d->rowptr.data(), sizeof(int64_t) * (d->n + 1), hipMemcpyHostToDevice, d->stream);
        std::vector<int32_t> col32;
        std::vector<uint8_t> pres;
        if (order == 2) {
            const int64_t W = 2 * nx - 1, H = 2 * ny - 1;
            col32.assign(d->colidx.begin(), d->colidx.end());
            pres.resize((size_t)d->n);
            for (int64_t r = 0; r < d->n; ++r) { const int64_t I = r % W, J = r / W; pres[(size_t)r] = (I == 0 || J == 0 || I == W - 1 || J == H - 1) ? 1 : 0; }
            if (e == hipSuccess) e = hipMalloc(&d->d_colidx, sizeof(int32_t) * d->nnz);
            if (e == hipSuccess) e = hipMalloc(&d->d_pres, (size_t)d->n);
            if (e == hipSuccess) e = hipMemcpyAsync(d->d_colidx, col32.data(), sizeof(int32_t) * d->nnz, hipMemcpyHostToDevice, d->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d->d_pres, pres.data(), (size_t)d->n, hipMemcpyHostToDevice, d->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        if (e != hipSuccess) {
            g_last_error = std::string("gmrf_darcy_p1_create: ") + hipGetErrorString(e);
            (void)gmrf_darcy_p1_destroy(d);
            return GMRF_ERR_HIP;
        }
    }
    *out = d;
    return GMRF_OK;
}

gmrf_status gmrf_darcy_p1_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_darcy_p1** out) {
    return darcy_create(device, stream, nx, ny, 1, out);
}

gmrf_status gmrf_darcy_p2_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_darcy_p1** out) {
    return darcy_create(device, stream, nx, ny, 2, out);
}

gmrf_status gmrf_darcy_p1_destroy(gmrf_darcy_p1* d) {
    if (!d) return GMRF_OK;
    if (d->device >= 0) {
        (void)hipSetDevice(d->device);
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        free_dev(d->d_rowptr); free_dev(d->d_diag); free_dev(d->d_mean); free_dev(d->d_table); free_dev(d->d_vals); free_dev(d->d_f);
        free_dev(d->d_colidx); free_dev(d->d_pres);
        if (d->own_stream) (void)hipStreamDestroy(d->stream);
    }
    delete d;
    return GMRF_OK;
}

gmrf_status gmrf_darcy_p1_pattern(const gmrf_darcy_p1* d, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx, int32_t index_base) {
    if (!d) return bad_shape("null handle");
    if (nnz_out) *nnz_out = d->nnz;
    if (rowptr) for (int64_t i = 0; i <= d->n; ++i) rowptr[i] = d->rowptr[(size_t)i] + index_base;
    if (colidx) for (int64_t e = 0; e < d->nnz; ++e) colidx[e] = d->colidx[(size_t)e] + index_base;
    return GMRF_OK;
}

gmrf_status gmrf_darcy_p1_assemble(gmrf_darcy_p1* d, const double* coeff_table, int64_t ng, double beta, double* vals_out,
                                   double* f_out) {
    if (!d || !coeff_table || !vals_out || !f_out || ng < 2 || ng > 46340) return bad_shape("bad Darcy assembly arguments");
    if (d->device < 0) { g_last_error = "pattern-only Darcy assembler (created with device -1)"; return GMRF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(d->device));
    const double* d_tab = coeff_table;
    if (!is_device_ptr(coeff_table)) {
        if (d->table_cap < ng * ng) {
            free_dev(d->d_table); d->d_table = nullptr; d->table_cap = 0;
            HIPCHK(hipMalloc(&d->d_table, sizeof(double) * ng * ng));
            d->table_cap = ng * ng;
        }
        HIPCHK(hipMemcpyAsync(d->d_table, coeff_table, sizeof(double) * ng * ng, hipMemcpyHostToDevice, d->stream));
        d_tab = d->d_table;
    }
    const bool v_dev = is_device_ptr(vals_out), f_dev = is_device_ptr(f_out);
    if (!v_dev && !d->d_vals) HIPCHK(hipMalloc(&d->d_vals, sizeof(double) * d->nnz));
    if (!f_dev && !d->d_f) HIPCHK(hipMalloc(&d->d_f, sizeof(double) * d->n));
    const dim3 grid((unsigned)((d->n + 255) / 256));
    if (d->order == 2) {
        DarcyP2Args a;
        a.nx = (int)d->nx; a.ny = (int)d->ny; a.ng = (int)ng; a.table = d_tab; a.rowptr = d->d_rowptr; a.beta = beta;
        a.vals = v_dev ? vals_out : d->d_vals; a.f = f_dev ? f_out : d->d_f; a.diag = d->d_diag;
        hipLaunchKernelGGL(darcy_p2_rows, grid, dim3(256), 0, d->stream, a);
        hipLaunchKernelGGL(darcy_meandiag, dim3(1), dim3(256), 0, d->stream, d->d_diag, d->n, d->d_mean);
        // apply!(G, f, ch): rows and columns of the boundary lattice points, meandiag on their diagonal, f = 0 there
        hipLaunchKernelGGL(csr_apply_constraints, grid, dim3(256), 0, d->stream, d->d_rowptr, d->d_colidx, d->d_pres, d->n, d->d_mean, a.vals);
        hipLaunchKernelGGL(vec_set_prescribed, grid, dim3(256), 0, d->stream, d->d_pres, d->n, (const double*)nullptr, 0.0, a.f);
    } else {
        DarcyP1Args a;
        a.nx = (int)d->nx; a.ny = (int)d->ny; a.ng = (int)ng; a.table = d_tab; a.rowptr = d->d_rowptr; a.beta = beta;
        a.vals = v_dev ? vals_out : d->d_vals; a.f = f_dev ? f_out : d->d_f; a.diag = d->d_diag;
        hipLaunchKernelGGL(darcy_p1_rows, grid, dim3(256), 0, d->stream, a);
        hipLaunchKernelGGL(darcy_meandiag, dim3(1), dim3(256), 0, d->stream, d->d_diag, d->n, d->d_mean);
        hipLaunchKernelGGL(darcy_p1_constrain, grid, dim3(256), 0, d->stream, a, d->d_mean);
    }
    HIPCHK(hipGetLastError());
    if (!v_dev) HIPCHK(hipMemcpyAsync(vals_out, d->d_vals, sizeof(double) * d->nnz, hipMemcpyDeviceToHost, d->stream));
    if (!f_dev) HIPCHK(hipMemcpyAsync(f_out, d->d_f, sizeof(double) * d->n, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return GMRF_OK;
}

// --------------------------------------------------------------------------------- Burgers tangent (8f rank 4)
struct gmrf_burgers_p1 {
    int device = -1;                    // -1: pattern only
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t ns = 0, nt = 0, rows = 0, nnz = 0;
    int order = 1;                      // 1: P1 line (6 entries per row), 2: quadratic line (10 / 6 entries per row)
    double dt = 0.0, nu = 0.0;
    double *d_w = nullptr, *d_vals = nullptr, *d_f = nullptr;       // staging for host callers
};

static gmrf_status burgers_line_create(int32_t device, void* stream, int64_t ns, int64_t nt, double dt, double nu, int order,
                                       gmrf_burgers_p1** out) {
    if (!out || ns < 3 || nt < 2 || ns > (1 << 24) || nt > (1 << 20) || !(dt > 0.0) || !(nu >= 0.0))
        return bad_shape("bad Burgers mesh (ns >= 3 nodes, nt >= 2 slices, dt > 0, nu >= 0)");
    if (order == 2 && (ns % 2 || ns < 6)) return bad_shape("the quadratic line has an even number of dofs (>= 6): two per cell");
    auto* b = new gmrf_burgers_p1();
    b->ns = ns; b->nt = nt; b->rows = (nt - 1) * ns; b->nnz = b->rows * (order == 2 ? 8 : 6); b->dt = dt; b->nu = nu; b->order = order;
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
            (void)hipGetLastError();
            delete b;
            g_last_error = "no HIP device visible (libgmrf_hip needs an MI355X / gfx950 GPU)";
            return GMRF_ERR_NO_DEVICE;
        }
        b->device = device;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) { if (stream) b->stream = (hipStream_t)stream; else { e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking); b->own_stream = (e == hipSuccess); } }
        if (e != hipSuccess) { g_last_error = std::string("gmrf_burgers_p1_create: ") + hipGetErrorString(e); delete b; return GMRF_ERR_HIP; }
    }
    *out = b;
    return GMRF_OK;
}

gmrf_status gmrf_burgers_p1_create(int32_t device, void* stream, int64_t ns, int64_t nt, double dt, double nu, gmrf_burgers_p1** out) {
    return burgers_line_create(device, stream, ns, nt, dt, nu, 1, out);
}
gmrf_status gmrf_burgers_p2_create(int32_t device, void* stream, int64_t ns, int64_t nt, double dt, double nu, gmrf_burgers_p1** out) {
    return burgers_line_create(device, stream, ns, nt, dt, nu, 2, out);
}

gmrf_status gmrf_burgers_p1_destroy(gmrf_burgers_p1* b) {
    if (!b) return GMRF_OK;
    if (b->device >= 0) {
        (void)hipSetDevice(b->device);
        if (b->stream) (void)hipStreamSynchronize(b->stream);
        free_dev(b->d_w); free_dev(b->d_vals); free_dev(b->d_f);
        if (b->own_stream) (void)hipStreamDestroy(b->stream);
    }
    delete b;
    return GMRF_OK;
}

// CSR pattern of J: row (t, i), t = 1 .. nt-1 (0-based slices), holds the columns {i-1, i, i+1} (periodic) of slices
// t-1 and t, ascending
gmrf_status gmrf_burgers_p1_pattern(const gmrf_burgers_p1* b, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx,
                                    int32_t index_base) {
    if (!b) return bad_shape("null handle");
    if (nnz_out) *nnz_out = b->nnz;
    if (b->order == 2) {
        // quadratic line: vertex rows (even i) hold the columns i-2 .. i+2, midpoint rows i-1 .. i+1 (periodic), of slices t-1 and t
        if (rowptr) {
            for (int64_t r = 0; r < b->rows; ++r) rowptr[r] = burgers_p2_row_offset(b->ns, r / b->ns, r % b->ns) + index_base;
            rowptr[b->rows] = b->nnz + index_base;
        }
        if (colidx)
            for (int64_t r = 0; r < b->rows; ++r) {
                const int64_t t = r / b->ns + 1, i = r % b->ns;
                const int cnt = (i & 1) ? 3 : 5;
                int64_t c[5];
                for (int k = 0; k < cnt; ++k) c[k] = (i - cnt / 2 + k + b->ns) % b->ns;
                std::sort(c, c + cnt);
                int64_t* out = colidx + burgers_p2_row_offset(b->ns, t - 1, i);
                for (int k = 0; k < cnt; ++k) { out[k] = (t - 1) * b->ns + c[k] + index_base; out[cnt + k] = t * b->ns + c[k] + index_base; }
            }
        return GMRF_OK;
    }
    if (rowptr) for (int64_t r = 0; r <= b->rows; ++r) rowptr[r] = 6 * r + index_base;
    if (colidx)
        for (int64_t r = 0; r < b->rows; ++r) {
            const int64_t t = r / b->ns + 1, i = r % b->ns;
            int64_t c[3] = {(i + b->ns - 1) % b->ns, i, (i + 1) % b->ns};
            std::sort(c, c + 3);
            for (int k = 0; k < 3; ++k) {
                colidx[6 * r + k] = (t - 1) * b->ns + c[k] + index_base;
                colidx[6 * r + 3 + k] = t * b->ns + c[k] + index_base;
            }
        }
    return GMRF_OK;
}

gmrf_status gmrf_burgers_p1_tangent(gmrf_burgers_p1* b, const double* w, double* vals_out, double* f_out) {
    if (!b || !w || !vals_out || !f_out) return bad_shape("bad Burgers tangent arguments");
    if (b->device < 0) { g_last_error = "pattern-only Burgers assembler (created with device -1)"; return GMRF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(b->device));
    const int64_t n = b->ns * b->nt;
    const double* d_w = w;
    if (!is_device_ptr(w)) {
        if (!b->d_w) HIPCHK(hipMalloc(&b->d_w, sizeof(double) * n));
        HIPCHK(hipMemcpyAsync(b->d_w, w, sizeof(double) * n, hipMemcpyHostToDevice, b->stream));
        d_w = b->d_w;
    }
    const bool v_dev = is_device_ptr(vals_out), f_dev = is_device_ptr(f_out);
    if (!v_dev && !b->d_vals) HIPCHK(hipMalloc(&b->d_vals, sizeof(double) * b->nnz));
    if (!f_dev && !b->d_f) HIPCHK(hipMalloc(&b->d_f, sizeof(double) * b->rows));
    BurgersP1Args a;
    a.ns = (int)b->ns; a.nt = (int)b->nt; a.dt = b->dt; a.nu = b->nu; a.w = d_w;
    a.vals = v_dev ? vals_out : b->d_vals; a.f = f_dev ? f_out : b->d_f;
    if (b->order == 2) hipLaunchKernelGGL(burgers_p2_rows, dim3((unsigned)((b->rows + 255) / 256)), dim3(256), 0, b->stream, a);
    else hipLaunchKernelGGL(burgers_p1_rows, dim3((unsigned)((b->rows + 255) / 256)), dim3(256), 0, b->stream, a);
    HIPCHK(hipGetLastError());
    if (!v_dev) HIPCHK(hipMemcpyAsync(vals_out, b->d_vals, sizeof(double) * b->nnz, hipMemcpyDeviceToHost, b->stream));
    if (!f_dev) HIPCHK(hipMemcpyAsync(f_out, b->d_f, sizeof(double) * b->rows, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return GMRF_OK;
}

// --------------------------------------------------------------------------------- shallow-water element kernels
struct gmrf_swe_p1 {
    int device = -1;                    // -1: patterns / quadrature points only
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t nx = 0, ny = 0, nn = 0, n = 0, cells = 0, nnz_k = 0, nnz_s = 0;
    std::vector<int64_t> rowptr_k, col_k, rowptr_s, col_s;     // 0-based
    int64_t *d_rowptr_k = nullptr, *d_rowptr_s = nullptr;
    int32_t *d_col_k = nullptr, *d_col_s = nullptr;
    double *d_dk = nullptr, *d_ds = nullptr, *d_mean = nullptr;   // |diagonals|, three meandiag scalars
    // staging for host callers
    double *d_hq = nullptr, *d_kv = nullptr, *d_sv = nullptr, *d_ml = nullptr, *d_g = nullptr, *d_j = nullptr, *d_mt = nullptr, *d_beta = nullptr;
    uint8_t* d_pres = nullptr;
};

gmrf_status gmrf_shallow_water_p1_destroy(gmrf_swe_p1* w) {
    if (!w) return GMRF_OK;
    if (w->device >= 0) {
        (void)hipSetDevice(w->device);
        if (w->stream) (void)hipStreamSynchronize(w->stream);
        free_dev(w->d_rowptr_k); free_dev(w->d_rowptr_s); free_dev(w->d_col_k); free_dev(w->d_col_s);
        free_dev(w->d_dk); free_dev(w->d_ds); free_dev(w->d_mean);
        free_dev(w->d_hq); free_dev(w->d_kv); free_dev(w->d_sv); free_dev(w->d_ml); free_dev(w->d_g); free_dev(w->d_j);
        free_dev(w->d_mt); free_dev(w->d_beta); free_dev(w->d_pres);
        if (w->own_stream) (void)hipStreamDestroy(w->stream);
    }
    delete w;
    return GMRF_OK;
}

gmrf_status gmrf_shallow_water_p1_create(int32_t device, void* stream, int64_t nx, int64_t ny, gmrf_swe_p1** out) {
    if (!out || nx < 2 || ny < 2 || nx > 16384 || ny > 16384) return bad_shape("bad shallow-water mesh size");
    auto* w = new gmrf_swe_p1();
    w->nx = nx; w->ny = ny; w->nn = nx * ny; w->n = 3 * w->nn; w->cells = 2 * (nx - 1) * (ny - 1);
    w->rowptr_k.assign((size_t)w->n + 1, 0); w->rowptr_s.assign((size_t)w->n + 1, 0);
    w->col_k.reserve((size_t)w->n * 21); w->col_s.reserve((size_t)w->n * 7);
    const int dxs[7] = {-1, 0, -1, 0, 1, 0, 1}, dys[7] = {-1, -1, 0, 0, 0, 1, 1};
    for (int64_t node = 0; node < w->nn; ++node) {
        const int64_t ix = node % nx, iy = node / nx;
        for (int fa = 0; fa < 3; ++fa) {
            const int64_t row = 3 * node + fa;
            for (int s7 = 0; s7 < 7; ++s7) {
                const int64_t jx = ix + dxs[s7], jy = iy + dys[s7];
                if (jx < 0 || jy < 0 || jx >= nx || jy >= ny) continue;
                const int64_t nj = jy * nx + jx;
                for (int fb = 0; fb < 3; ++fb) w->col_k.push_back(3 * nj + fb);      // full field coupling (:140)
                w->col_s.push_back(3 * nj + fa);                                     // block-diagonal coupling (:141-150)
            }
            w->rowptr_k[(size_t)row + 1] = (int64_t)w->col_k.size();
            w->rowptr_s[(size_t)row + 1] = (int64_t)w->col_s.size();
        }
    }
    w->nnz_k = (int64_t)w->col_k.size(); w->nnz_s = (int64_t)w->col_s.size();
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
            (void)hipGetLastError();
            delete w;
            g_last_error = "no HIP device visible (libgmrf_hip needs an MI355X / gfx950 GPU)";
            return GMRF_ERR_NO_DEVICE;
        }
        w->device = device;
        std::vector<int32_t> ck(w->col_k.begin(), w->col_k.end()), cs(w->col_s.begin(), w->col_s.end());
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) { if (stream) w->stream = (hipStream_t)stream; else { e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking); w->own_stream = (e == hipSuccess); } }
        if (e == hipSuccess) e = hipMalloc(&w->d_rowptr_k, sizeof(int64_t) * (w->n + 1));
        if (e == hipSuccess) e = hipMalloc(&w->d_rowptr_s, sizeof(int64_t) * (w->n + 1));
        if (e == hipSuccess) e = hipMalloc(&w->d_col_k, sizeof(int32_t) * w->nnz_k);
        if (e == hipSuccess) e = hipMalloc(&w->d_col_s, sizeof(int32_t) * w->nnz_s);
        if (e == hipSuccess) e = hipMalloc(&w->d_dk, sizeof(double) * w->n);
        if (e == hipSuccess) e = hipMalloc(&w->d_ds, sizeof(double) * w->n);
        if (e == hipSuccess) e = hipMalloc(&w->d_mean, sizeof(double) * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(w->d_rowptr_k, w->rowptr_k.data(), sizeof(int64_t) * (w->n + 1), hipMemcpyHostToDevice, w->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(w->d_rowptr_s, w->rowptr_s.data(), sizeof(int64_t) * (w->n + 1), hipMemcpyHostToDevice, w->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(w->d_col_k, ck.data(), sizeof(int32_t) * w->nnz_k, hipMemcpyHostToDevice, w->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(w->d_col_s, cs.data(), sizeof(int32_t) * w->nnz_s, hipMemcpyHostToDevice, w->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
        if (e != hipSuccess) {
            g_last_error = std::string("gmrf_shallow_water_p1_create: ") + hipGetErrorString(e);
            (void)gmrf_shallow_water_p1_destroy(w);
            return GMRF_ERR_HIP;
        }
    }
    *out = w;
    return GMRF_OK;
}

gmrf_status gmrf_shallow_water_p1_pattern(const gmrf_swe_p1* w, int32_t which, int64_t* nnz_out, int64_t* rowptr, int64_t* colidx,
                                          int32_t index_base) {
    if (!w || (which != 0 && which != 1)) return bad_shape("pattern: 0 = K (coupling), 1 = S / M (block diagonal)");
    const auto& rp = which == 0 ? w->rowptr_k : w->rowptr_s;
    const auto& ci = which == 0 ? w->col_k : w->col_s;
    if (nnz_out) *nnz_out = (int64_t)ci.size();
    if (rowptr) for (int64_t i = 0; i <= w->n; ++i) rowptr[i] = rp[(size_t)i] + index_base;
    if (colidx) for (size_t p = 0; p < ci.size(); ++p) colidx[p] = ci[p] + index_base;
    return GMRF_OK;
}

// spatial_coordinate(cvh, qp, cell_coords) (:52) of every cell: xy[cell][q][2]
gmrf_status gmrf_shallow_water_p1_qpoints(const gmrf_swe_p1* w, double* xy) {
    if (!w || !xy) return bad_shape("null pointer");
    if (is_device_ptr(xy)) return bad_shape("quadrature points are written to host memory");
    const int64_t nx = w->nx, ny = w->ny, nlow = (nx - 1) * (ny - 1);
    auto lin = [](int64_t i, int64_t n) { return (i == n - 1) ? 1.0 : (double)i * (1.0 / (double)(n - 1)); };
    const double bary[3][3] = {{1.0 / 6, 1.0 / 6, 2.0 / 3}, {1.0 / 6, 2.0 / 3, 1.0 / 6}, {2.0 / 3, 1.0 / 6, 1.0 / 6}};
    for (int up = 0; up < 2; ++up)
        for (int64_t qy = 0; qy < ny - 1; ++qy)
            for (int64_t qx = 0; qx < nx - 1; ++qx) {
                const int64_t cell = up * nlow + qy * (nx - 1) + qx;
                const int64_t nxs[3] = {qx, qx + 1, up ? qx : qx + 1}, nys[3] = {qy, up ? qy + 1 : qy, qy + 1};
                for (int q = 0; q < 3; ++q) {
                    xy[(cell * 3 + q) * 2 + 0] = (bary[q][0] * lin(nxs[0], nx) + bary[q][1] * lin(nxs[1], nx)) + bary[q][2] * lin(nxs[2], nx);
                    xy[(cell * 3 + q) * 2 + 1] = (bary[q][0] * lin(nys[0], ny) + bary[q][1] * lin(nys[1], ny)) + bary[q][2] * lin(nys[2], ny);
                }
            }
    return GMRF_OK;
}

static gmrf_status swe_ready(gmrf_swe_p1* w) {
    if (!w) return bad_shape("null handle");
    if (w->device < 0) { g_last_error = "pattern-only shallow-water handle (created with device -1)"; return GMRF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(w->device));
    return GMRF_OK;
}

static gmrf_status swe_in_bytes(gmrf_swe_p1* w, const void* p, size_t bytes, void** buf, const void** d) {
    if (!p) { *d = nullptr; return GMRF_OK; }
    if (is_device_ptr(p)) { *d = p; return GMRF_OK; }
    if (!*buf) HIPCHK(hipMalloc(buf, std::max<size_t>(bytes, 8)));
    HIPCHK(hipMemcpyAsync(*buf, p, bytes, hipMemcpyHostToDevice, w->stream));
    *d = *buf;
    return GMRF_OK;
}
static gmrf_status swe_in(gmrf_swe_p1* w, const double* p, int64_t count, double** buf, const double** d) {
    return swe_in_bytes(w, p, sizeof(double) * (size_t)count, (void**)buf, (const void**)d);
}
static gmrf_status swe_in(gmrf_swe_p1* w, const uint8_t* p, int64_t count, uint8_t** buf, const uint8_t** d) {
    return swe_in_bytes(w, p, (size_t)count, (void**)buf, (const void**)d);
}

static gmrf_status swe_out(gmrf_swe_p1* w, double* user, int64_t count, double** buf, double** d) {
    if (is_device_ptr(user)) { *d = user; return GMRF_OK; }
    if (!*buf) HIPCHK(hipMalloc(buf, sizeof(double) * std::max<int64_t>(count, 1)));
    *d = *buf;
    return GMRF_OK;
}

// meandiag of a vector of |diagonal| values + apply! on CSR values
static gmrf_status swe_constrain(gmrf_swe_p1* w, const int64_t* d_rowptr, const int32_t* d_col, const uint8_t* d_pres, const double* d_absdiag,
                                 double* d_mean, double* d_vals) {
    hipLaunchKernelGGL(darcy_meandiag, dim3(1), dim3(256), 0, w->stream, d_absdiag, w->n, d_mean);
    hipLaunchKernelGGL(csr_apply_constraints, dim3((unsigned)((w->n + 255) / 256)), dim3(256), 0, w->stream, d_rowptr, d_col, d_pres, w->n,
                       d_mean, d_vals);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

gmrf_status gmrf_shallow_water_p1_assemble(gmrf_swe_p1* w, const double* H_q, double k, double f, double g, const uint8_t* prescribed,
                                           double* K_vals, double* M_lumped, double* S_vals) {
    GCHK(swe_ready(w));
    if (!H_q || !K_vals || !M_lumped || !S_vals) return bad_shape("null pointer");
    const double* d_hq; const uint8_t* d_pres;
    GCHK(swe_in(w, H_q, w->cells * 3, &w->d_hq, &d_hq));
    GCHK(swe_in(w, prescribed, w->n, &w->d_pres, &d_pres));
    double *d_kv, *d_sv, *d_ml;
    GCHK(swe_out(w, K_vals, w->nnz_k, &w->d_kv, &d_kv));
    GCHK(swe_out(w, S_vals, w->nnz_s, &w->d_sv, &d_sv));
    GCHK(swe_out(w, M_lumped, w->n, &w->d_ml, &d_ml));
    SweP1Args a;
    a.nx = (int)w->nx; a.ny = (int)w->ny; a.Hq = d_hq; a.k = k; a.f = f; a.g = g;
    a.rowptr_k = w->d_rowptr_k; a.rowptr_s = w->d_rowptr_s; a.kv = d_kv; a.sv = d_sv; a.ml = d_ml; a.dk = w->d_dk; a.ds = w->d_ds;
    hipLaunchKernelGGL(swe_p1_rows, dim3((unsigned)((w->n + 255) / 256)), dim3(256), 0, w->stream, a);
    HIPCHK(hipGetLastError());
    if (d_pres) {
        // apply!(K, ..), apply!(M, ..), apply!(S, ..)  (:119-121)
        GCHK(swe_constrain(w, w->d_rowptr_k, w->d_col_k, d_pres, w->d_dk, w->d_mean, d_kv));
        GCHK(swe_constrain(w, w->d_rowptr_s, w->d_col_s, d_pres, w->d_ds, w->d_mean + 1, d_sv));
        hipLaunchKernelGGL(darcy_meandiag, dim3(1), dim3(256), 0, w->stream, d_ml, w->n, w->d_mean + 2);      // M >= 0: |M_ii| = M_ii
        hipLaunchKernelGGL(vec_set_prescribed, dim3((unsigned)((w->n + 255) / 256)), dim3(256), 0, w->stream, d_pres, w->n,
                           (const double*)(w->d_mean + 2), 0.0, d_ml);
        HIPCHK(hipGetLastError());
    }
    if (d_kv != K_vals) HIPCHK(hipMemcpyAsync(K_vals, d_kv, sizeof(double) * w->nnz_k, hipMemcpyDeviceToHost, w->stream));
    if (d_sv != S_vals) HIPCHK(hipMemcpyAsync(S_vals, d_sv, sizeof(double) * w->nnz_s, hipMemcpyDeviceToHost, w->stream));
    if (d_ml != M_lumped) HIPCHK(hipMemcpyAsync(M_lumped, d_ml, sizeof(double) * w->n, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    return GMRF_OK;
}

gmrf_status gmrf_shallow_water_p1_operators(gmrf_swe_p1* w, const double* K_vals, const double* M_lumped, const double* S_vals,
                                            const uint8_t* prescribed, double kappa_matern, double tau, double dt, double* G_vals,
                                            double* J_vals, double* M_tilde, double* beta) {
    GCHK(swe_ready(w));
    if (!K_vals || !M_lumped || !S_vals || !G_vals || !J_vals || !M_tilde || !beta) return bad_shape("null pointer");
    if (!(kappa_matern > 0.0) || !(dt > 0.0)) return bad_shape("kappa_matern and dt must be positive");
    SweOpArgs a;
    const uint8_t* d_pres;
    // (inputs that live on the host are staged into the buffers the assemble call uses for its outputs)
    GCHK(swe_in(w, K_vals, w->nnz_k, &w->d_kv, &a.kv));
    GCHK(swe_in(w, S_vals, w->nnz_s, &w->d_sv, &a.sv));
    GCHK(swe_in(w, M_lumped, w->n, &w->d_ml, &a.ml));
    GCHK(swe_in(w, prescribed, w->n, &w->d_pres, &d_pres));
    GCHK(swe_out(w, G_vals, w->nnz_k, &w->d_g, &a.Gd));
    GCHK(swe_out(w, J_vals, w->nnz_s, &w->d_j, &a.J));
    GCHK(swe_out(w, M_tilde, w->n, &w->d_mt, &a.Mt));
    GCHK(swe_out(w, beta, w->n, &w->d_beta, &a.beta));
    a.n = w->n; a.rowptr_k = w->d_rowptr_k; a.col_k = w->d_col_k; a.rowptr_s = w->d_rowptr_s; a.col_s = w->d_col_s;
    a.pres = d_pres; a.kappa2 = kappa_matern * kappa_matern; a.tau = tau; a.dt = dt; a.dg = w->d_dk;
    const double nu = 2.0;                                     // :180
    a.sqrt_ratio = std::sqrt(std::tgamma(nu) / (std::tgamma(nu + 1.0) * (4.0 * M_PI) * std::pow(kappa_matern, 2.0 * nu)));
    hipLaunchKernelGGL(swe_p1_operators, dim3((unsigned)((w->n + 255) / 256)), dim3(256), 0, w->stream, a);
    HIPCHK(hipGetLastError());
    if (d_pres) GCHK(swe_constrain(w, w->d_rowptr_k, w->d_col_k, d_pres, w->d_dk, w->d_mean + 3, a.Gd));      // apply!(S_tmp, f, ch) :213
    if (a.Gd != G_vals) HIPCHK(hipMemcpyAsync(G_vals, a.Gd, sizeof(double) * w->nnz_k, hipMemcpyDeviceToHost, w->stream));
    if (a.J != J_vals) HIPCHK(hipMemcpyAsync(J_vals, a.J, sizeof(double) * w->nnz_s, hipMemcpyDeviceToHost, w->stream));
    if (a.Mt != M_tilde) HIPCHK(hipMemcpyAsync(M_tilde, a.Mt, sizeof(double) * w->n, hipMemcpyDeviceToHost, w->stream));
    if (a.beta != beta) HIPCHK(hipMemcpyAsync(beta, a.beta, sizeof(double) * w->n, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    return GMRF_OK;
}

// --------------------------------------------------------------------------------- variances
static gmrf_status need_single(gmrf_handle* h) {
    if (h->B != 1) return bad_shape("marginal variances are computed per handle with batch 1");
    return GMRF_OK;
}

static gmrf_status var_exact(gmrf_handle* h, double* d_out) {
    // Selected inversion (Takahashi recurrence on the block-tridiagonal factor), all problems of a batch in lock step:
    //   S_NN = X_N^T X_N,   S_ii = X_i^T (I + C_i^T S_{i+1,i+1} C_i) X_i,      X_i = Linv_i, C_i = L_{i+1,i};  d_out[B][n] = diag.
    // Round 4: what the recurrence NEEDS of S_ii is its diagonal and its leading rmax x rmax block (C_i is zero below row rmax
    // and left of column cmin), and both follow from  S_ii = X^T Y,  Y = X with its rows >= cmin replaced by
    // (I + M) X[cmin:, :],  M = C_w^T S' C_w  (C_w: the stored window of C_i, S' = S_{i+1,i+1}[0:rmax, 0:rmax]):
    //   T1 = S' C_w                 rmax x wc,   K = rmax, column tile u of C_w ends at row mend[u]
    //   M  = T1^T C_w               wc x wc, lower tiles, same K bound (T1 as it lies: gemm_f64_dma's A [k][m] form), mirrored
    //   Y[cmin:, :] = M X[cmin:, :] + X[cmin:, :]          K = wc from the first non-zero row of each column tile of X
    //   S[0:rmax, 0:rmax] = X^T[0:rmax, :] Y[:, 0:rmax]    lower tiles, K from each row tile's diagonal (X as it lies, [k][m])
    //   diag(S)[n] = sum_k X[k][n] Y[k][n]                  (coldot_lower)
    // 1.9 GF per 1024-block instead of 5.7 (two full 1024^3 products, two of them on the register-staged kernel that the
    // [k][m] operand forced), every product on gemm_f64_dma.
    const int bsp = (int)h->bsp;
    const int64_t ld = bsp, bstride = (int64_t)bsp * bsp;
    const int64_t pX = stride_pX(h), pCm = stride_pC(h), pW = bstride;
    const int cm = (int)h->cmin, rm = (int)h->rmax, wc = bsp - cm;
    const unsigned nb = (unsigned)h->B;
    GCHK(ensure_full_inverse(h));
    if (!h->d_V || h->v_elems < 2 * bstride * h->B) {
        HIPCHK(hipStreamSynchronize(h->stream));
        free_dev(h->d_V); h->d_V = nullptr; h->v_elems = 0;
        HIPCHK(hipMalloc(&h->d_V, sizeof(double) * 2 * (size_t)bstride * (size_t)h->B));
        h->v_elems = 2 * bstride * h->B;
    }
    double* Sg = h->d_S;     // leading block of Sigma_{i+1,i+1} (full, mirrored)
    double* Sn = h->d_W;
    double* Y = h->d_T;
    double* V1 = h->d_V;                                 // T1
    double* V2 = h->d_V + bstride * h->B;                // M
    const int nt = bsp / 64, trm = rm / 64, twc = wc / 64;
    for (int64_t i = h->N - 1; i >= 0; --i) {
        const double* X = h->d_Linv + i * bstride;
        const bool coupled = i < h->N - 1;
        if (coupled) {
            const double* C = h->d_C + i * c_blk(h);
            // T1 = S' C_w
            GCHK(gemm(h, false, true, rm, wc, rm, 0, 0, 1.0, Sg, ld, C, c_ld(h), 0.0, V1, ld, pW, pCm, pW,
                      1, 0, 0, 0, nullptr, 0, 0, 0, 2.0 * rm * h->c_streamed * (double)h->B, nullptr, nullptr, h->d_mend));
            // M = T1^T C_w (lower tiles; T1 taken as it lies, [k][m]), mirrored
            GCHK(gemm(h, true, true, wc, wc, rm, 0, 1, 1.0, V1, ld, C, c_ld(h), 0.0, V2, ld, pW, pCm, pW,
                      1, 0, 0, 0, nullptr, 0, 0, 0, -1.0, nullptr, nullptr, h->d_mend));
            hipLaunchKernelGGL(transpose_tiles, dim3((unsigned)(twc * (twc + 1) / 2), nb), dim3(256), 0, h->stream, V2, ld, V2, ld, pW, pW, twc, twc, 2);
            HIPCHK(hipGetLastError());
            // Y[cmin:, :] = M X[cmin:, :] + X[cmin:, :]   (column tile u of X[cmin:, :] starts at row max(0, 64 u - cmin))
            GCHK(gemm(h, false, true, wc, bsp, wc, 0, 0, 1.0, V2, ld, X + (int64_t)cm * ld, ld, 1.0, Y + (int64_t)cm * ld, ld, pW, pX, pW,
                      1, 0, 0, 0, X + (int64_t)cm * ld, ld, pX, 0, -1.0, nullptr, h->d_kbx, nullptr));
        }
        // diag(S_ii): of the last block processed (i = 0) and of the columns >= rmax by column dot products; the columns below
        // rmax of the other blocks are the diagonal of the leading block formed below (saves reading X and Y once more)
        const int u0 = (i == 0) ? 0 : trm;
        if (u0 < nt) {
            hipLaunchKernelGGL(coldot_lower, dim3((unsigned)(nt - u0), nb), dim3(256), 0, h->stream, X, coupled ? Y : nullptr, ld, bsp, cm, (int)h->bs,
                               d_out + i * h->bs, pX, pW, h->n, u0);
            HIPCHK(hipGetLastError());
        }
        if (i == 0) break;                               // nobody needs the leading block of S_00
        // leading block of S_ii for the next step: the product X^T Y on its lower tiles (X taken as it lies, [k][m]), mirrored
        if (coupled) {
            if (cm > 0)
                GCHK(gemm(h, true, true, rm, rm, cm, TRI_A_UPPER | TRI_B_LOWER, 1, 1.0, X, ld, X, ld, 0.0, Sn, ld, pX, pX, pW));
            GCHK(gemm(h, true, true, rm, rm, wc, 0, 1, 1.0, X + (int64_t)cm * ld, ld, Y + (int64_t)cm * ld, ld, cm > 0 ? 1.0 : 0.0, Sn, ld, pX, pW, pW,
                      1, 0, 0, 0, nullptr, 0, 0, 0, -1.0, h->d_kbx, nullptr, nullptr));
        } else {
            GCHK(gemm(h, true, true, rm, rm, bsp, TRI_A_UPPER | TRI_B_LOWER, 1, 1.0, X, ld, X, ld, 0.0, Sn, ld, pX, pX, pW));
        }
        hipLaunchKernelGGL(transpose_tiles, dim3((unsigned)(trm * (trm + 1) / 2), nb), dim3(256), 0, h->stream, Sn, ld, Sn, ld, pW, pW, trm, trm, 2);
        HIPCHK(hipGetLastError());
        {
            const int cnt = (int)std::min<int64_t>(rm, h->bs);
            hipLaunchKernelGGL(extract_diag_dense, dim3((unsigned)((cnt + 255) / 256), nb), dim3(256), 0, h->stream, Sn, ld, cnt,
                               d_out + i * h->bs, bstride, h->n);
            HIPCHK(hipGetLastError());
        }
        std::swap(Sg, Sn);
    }
    return GMRF_OK;
}

// One chunk of samples (panel p of d_Y, kc right-hand sides) -> node-major X[n][kcp] -> the RBMC / MC
// accumulator: the samples leave the sweep as a panel (each right-hand side contiguous); the
// transposing unpack puts the kc values of a node side by side, which is the layout the LDS-tiled
// SpMM and the accumulator read with full 128-byte lines.
static gmrf_status var_chunk(gmrf_handle* h, int method, int64_t p, int kc, const gmrf_csr* Q, const double* q_vals,
                             const double* d_diag, double* d_acc) {
    const int64_t n = h->n;
    const int kp = pad_k(kc), kcp = kc + (kc & 1);
    GCHK(ensure_stage(h, 2 * (int64_t)kcp * n));
    double* Xr = h->d_stage;
    double* QX = h->d_stage + (int64_t)kcp * n;
    hipLaunchKernelGGL(unpack_panel_rows, dim3((unsigned)((n + 63) / 64), (unsigned)((kc + 63) / 64)), dim3(256), 0, h->stream,
                       h->d_Y + p * kp * h->n_pad, h->n_pad, Xr, (int64_t)kcp, (int)h->bs, (int)h->bsp, n, kc);
    HIPCHK(hipGetLastError());
    if (method == GMRF_VAR_RBMC) GCHK(spmm_rows_device(Q, h->stream, Xr, kcp, QX, kcp, kc, q_vals));
    hipLaunchKernelGGL(rbmc_accumulate_rows, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, h->stream, QX, Xr, (int64_t)kcp,
                       d_diag, n, kc, d_acc, method == GMRF_VAR_RBMC ? 0 : 1);
    HIPCHK(hipGetLastError());
    return GMRF_OK;
}

static gmrf_status var_accumulate_dev(gmrf_handle* h, int method, int64_t first_id, int64_t k, uint64_t seed,
                                      const gmrf_csr* Q, double* d_acc) {
    for (int64_t c0 = 0; c0 < k; c0 += 64) {
        const int kc = (int)std::min<int64_t>(64, k - c0);
        GCHK(sample_chunk(h, seed, first_id + c0, kc, nullptr, 0, 0));
        GCHK(var_chunk(h, method, 0, kc, Q, nullptr, Q ? Q->d_diag : nullptr, d_acc));
    }
    return GMRF_OK;
}

gmrf_status gmrf_bt_var_accumulate(gmrf_handle* h, int32_t method, int64_t first_id, int64_t k, uint64_t seed,
                                   const gmrf_csr* Q, double* acc) {
    if (!h || !acc) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "variance before factor"; return GMRF_ERR_NO_FACTOR; }
    if (method != GMRF_VAR_RBMC && method != GMRF_VAR_MC) return bad_shape("accumulate needs RBMC or MC");
    if (method == GMRF_VAR_RBMC && (!Q || Q->n_rows != h->n || !Q->d_diag)) return bad_shape("RBMC needs the square matrix Q");
    if (k <= 0) return bad_shape("k <= 0");
    GCHK(need_single(h));
    HIPCHK(hipSetDevice(h->device));
    const bool dev = is_device_ptr(acc);
    double* d_acc = acc;
    if (!dev) {
        if (!h->d_acc) { HIPCHK(hipMalloc(&h->d_acc, sizeof(double) * h->n)); h->acc_B = 1; }
        HIPCHK(hipMemcpyAsync(h->d_acc, acc, sizeof(double) * h->n, hipMemcpyHostToDevice, h->stream));
        d_acc = h->d_acc;
    }
    GCHK(var_accumulate_dev(h, method, first_id, k, seed, Q, d_acc));
    if (!dev) HIPCHK(hipMemcpyAsync(acc, d_acc, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMRF_OK;
}

gmrf_status gmrf_bt_marginal_var(gmrf_handle* h, int32_t method, int64_t k, uint64_t seed, const gmrf_csr* Q,
                                 double* var_out) {
    if (!h || !var_out) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "variance before factor"; return GMRF_ERR_NO_FACTOR; }
    if (method != GMRF_VAR_EXACT) GCHK(need_single(h));    // sampled estimators: one problem per handle
    HIPCHK(hipSetDevice(h->device));
    if (!h->d_acc || h->acc_B < h->B) {
        free_dev(h->d_acc); h->d_acc = nullptr;
        HIPCHK(hipMalloc(&h->d_acc, sizeof(double) * h->n * h->B));
        h->acc_B = h->B;
    }
    const bool dev = is_device_ptr(var_out);
    double* d_out = var_out;
    if (!dev) { GCHK(ensure_stage(h, std::max<int64_t>(h->n, 1))); }
    if (method == GMRF_VAR_EXACT) {
        // selected inversion; var_out is [batch][n]
        GCHK(var_exact(h, h->d_acc));
        const size_t bytes = sizeof(double) * h->n * h->B;
        if (dev) HIPCHK(hipMemcpyAsync(d_out, h->d_acc, bytes, hipMemcpyDeviceToDevice, h->stream));
        else HIPCHK(hipMemcpyAsync(var_out, h->d_acc, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->profiling) prof_collect(h);
        return GMRF_OK;
    }
    if (method != GMRF_VAR_RBMC && method != GMRF_VAR_MC) return bad_shape("bad variance method");
    if (method == GMRF_VAR_RBMC && (!Q || Q->n_rows != h->n || !Q->d_diag)) return bad_shape("RBMC needs the square matrix Q");
    if (k <= 0) return bad_shape("k <= 0");
    HIPCHK(hipMemsetAsync(h->d_acc, 0, sizeof(double) * h->n, h->stream));
    GCHK(var_accumulate_dev(h, method, 0, k, seed, Q, h->d_acc));
    // finish in place: var = base + acc / k
    double* d_fin = h->d_acc;
    hipLaunchKernelGGL(var_finish, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->d_acc,
                       method == GMRF_VAR_RBMC ? Q->d_diag : (const double*)nullptr, 1.0 / (double)k, h->n, d_fin);
    HIPCHK(hipGetLastError());
    if (dev) HIPCHK(hipMemcpyAsync(d_out, d_fin, sizeof(double) * h->n, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(hipMemcpyAsync(var_out, d_fin, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->profiling) prof_collect(h);
    return GMRF_OK;
}

// Sampled marginal variances of every problem of a batch (the reference's per-problem
// `std(x_cond)` with RBMCStrategy(k), scripts/darcy/solve_darcy_gmrf-fem.jl:174,192).  Q gives the
// sparsity pattern (its values are not used); q_vals[p] are problem p's values in Q's CSR order --
// for a symmetric matrix the nzval array the factor was given.  Problem p draws the sample ids
// p * k .. p * k + k - 1, so problem 0 equals what a one-problem handle computes.
gmrf_status gmrf_bt_marginal_var_batch(gmrf_handle* h, int32_t method, int64_t k, uint64_t seed, const gmrf_csr* Q,
                                       const double* q_vals, double* var_out) {
    if (!h || !var_out) return bad_shape("null pointer");
    if (!h->factored) { g_last_error = "variance before factor"; return GMRF_ERR_NO_FACTOR; }
    if (method != GMRF_VAR_RBMC && method != GMRF_VAR_MC) return bad_shape("batch variances: RBMC or MC (exact: gmrf_bt_marginal_var)");
    if (method == GMRF_VAR_RBMC && (!Q || Q->n_rows != h->n || Q->n_cols != h->n || !q_vals)) return bad_shape("RBMC needs the pattern of Q and its values per problem");
    if (k <= 0) return bad_shape("k <= 0");
    HIPCHK(hipSetDevice(h->device));
    const int64_t n = h->n, B = h->B;
    if (!h->d_acc || h->acc_B < B) {
        free_dev(h->d_acc); h->d_acc = nullptr;
        HIPCHK(hipMalloc(&h->d_acc, sizeof(double) * n * B));
        h->acc_B = B;
    }
    HIPCHK(hipMemsetAsync(h->d_acc, 0, sizeof(double) * n * B, h->stream));
    double *d_qv = nullptr, *d_diag = nullptr;
    const double* qv = q_vals;
    if (method == GMRF_VAR_RBMC) {
        if (!is_device_ptr(q_vals)) {
            HIPCHK(hipMalloc(&d_qv, sizeof(double) * Q->nnz * B));
            if (hipMemcpyAsync(d_qv, q_vals, sizeof(double) * Q->nnz * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
                free_dev(d_qv); g_last_error = "hipMemcpyAsync(q_vals) failed"; return GMRF_ERR_HIP;
            }
            qv = d_qv;
        }
        if (hipMalloc(&d_diag, sizeof(double) * n * B) != hipSuccess) { free_dev(d_qv); g_last_error = "hipMalloc(diag) failed"; return GMRF_ERR_HIP; }
        for (int64_t p = 0; p < B; ++p)
            hipLaunchKernelGGL(csr_extract_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, Q->d_rowptr,
                               Q->d_colidx, qv + p * Q->nnz, (const float*)nullptr, n, d_diag + p * n);
        if (hipGetLastError() != hipSuccess) { free_dev(d_qv); free_dev(d_diag); g_last_error = "csr_extract_diag launch failed"; return GMRF_ERR_HIP; }
    }
    gmrf_status st = GMRF_OK;
    for (int64_t c0 = 0; c0 < k && st == GMRF_OK; c0 += 64) {
        const int kc = (int)std::min<int64_t>(64, k - c0);
        st = sample_chunk(h, seed, c0, kc, nullptr, 0, k);          // every problem's chunk in one sweep
        for (int64_t p = 0; p < B && st == GMRF_OK; ++p)
            st = var_chunk(h, method, p, kc, Q, method == GMRF_VAR_RBMC ? qv + p * Q->nnz : nullptr,
                           method == GMRF_VAR_RBMC ? d_diag + p * n : nullptr, h->d_acc + p * n);
    }
    if (st == GMRF_OK) {
        for (int64_t p = 0; p < B; ++p)
            hipLaunchKernelGGL(var_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_acc + p * n,
                               method == GMRF_VAR_RBMC ? d_diag + p * n : (const double*)nullptr, 1.0 / (double)k, n,
                               h->d_acc + p * n);
        if (hipGetLastError() != hipSuccess) st = GMRF_ERR_HIP;
        const hipMemcpyKind kind = is_device_ptr(var_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (st == GMRF_OK && hipMemcpyAsync(var_out, h->d_acc, sizeof(double) * n * B, kind, h->stream) != hipSuccess) st = GMRF_ERR_HIP;
    }
    (void)hipStreamSynchronize(h->stream);
    free_dev(d_qv); free_dev(d_diag);
    return st;
}

// --------------------------------------------------------------------------------- test hooks
gmrf_status gmrf_test_gemm(int32_t device, int64_t M, int64_t N, int64_t K, int32_t transA, int32_t transB,
                           int32_t tri_flags, int32_t lower_only, double alpha, const double* A, int64_t lda,
                           const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
    if (M % 64 || N % 64 || K % 16 || (lda & 1) || (ldb & 1)) return bad_shape("gemm test sizes");
    HIPCHK(hipSetDevice(device));
    const int64_t a_rows = transA ? K : M, b_rows = transB ? N : K;
    double *dA, *dB, *dC;
    HIPCHK(hipMalloc(&dA, sizeof(double) * a_rows * lda));
    HIPCHK(hipMalloc(&dB, sizeof(double) * b_rows * ldb));
    HIPCHK(hipMalloc(&dC, sizeof(double) * M * ldc));
    HIPCHK(hipMemcpy(dA, A, sizeof(double) * a_rows * lda, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, B, sizeof(double) * b_rows * ldb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dC, C, sizeof(double) * M * ldc, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A = dA; g.B = dB; g.C = dC; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.strideA = g.strideB = g.strideC = 0;
    g.pA = g.pB = g.pC = 0; g.nb1 = 1;
    g.D = nullptr; g.ldd = 0; g.pD = 0;
    // tri_flags bit 2048: take the 128 x 128 kernel whatever the tile count
    // bit 4096: the 32 x 32 low-latency kernel (taken by launches of <= 128 tiles of 64 x 64)
    // bits 8192 / 16384 / 32768: the LDS-DMA kernel with 64 x 64 / 128 x 64 / 64 x 128 tiles (gemm_f64_dma.hpp); none of
    // them: the register-staged kernels only
    const bool force_big = (tri_flags & 2048) != 0, use_ll = (tri_flags & 4096) != 0;
    const int dma_shape = (tri_flags & 8192) ? 1 : ((tri_flags & 16384) ? 2 : ((tri_flags & 32768) ? 3 : 0));
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.tri = tri_flags & ~(2048 | 4096 | 8192 | 16384 | 32768); g.lower_only = lower_only;
    g.alpha = alpha; g.beta = beta;
    unsigned long long* dst = nullptr;
    HIPCHK(hipMalloc(&dst, 16));
    HIPCHK(hipMemset(dst, 0, 16));
    g.stamps = (force_big || use_ll || dma_shape) ? nullptr : dst;
    HIPCHK(gemm_init());
    HIPCHK(gemm_dma_init());
    const int saved = gemm_big_policy(), saved_ll = gemm_ll_policy(), saved_dma = gemm_dma_policy(), saved_force = gemm_dma_force();
    gemm_big_policy() = force_big ? 1 : 2;
    gemm_ll_policy() = use_ll ? 0 : 2;
    gemm_dma_policy() = dma_shape ? 2 : 0;
    gemm_dma_force() = dma_shape;
    // BLAS-style flags: op(A) is M x K, op(B) is K x N; B "not transposed" is stored K x N
    hipError_t le = launch_gemm(nullptr, transA != 0, transB == 0, g, 1);
    gemm_big_policy() = saved; gemm_ll_policy() = saved_ll; gemm_dma_policy() = saved_dma; gemm_dma_force() = saved_force;
    HIPCHK(le);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(C, dC, sizeof(double) * M * ldc, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(g_tile_stamps, dst, 16, hipMemcpyDeviceToHost));
    hipFree(dst);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return GMRF_OK;
}

// rows of 11 doubles per distinct GEMM launch shape seen while profiling was on: class, M, N, K, tri, lower_only, problems,
// per-tile K bounds?, launches, ms, work (flops as booked in kernel_work).  *n_rows = rows available (may exceed cap_rows).
gmrf_status gmrf_test_gemm_shapes(gmrf_handle* h, double* rows, int64_t cap_rows, int64_t* n_rows) {
    if (!h || !n_rows) return bad_shape("null argument");
    HIPCHK(hipStreamSynchronize(h->stream));
    prof_collect(h);
    *n_rows = (int64_t)h->gemm_shapes.size();
    for (int64_t i = 0; rows && i < std::min<int64_t>(cap_rows, *n_rows); ++i) {
        const auto& g = h->gemm_shapes[(size_t)i];
        for (int c = 0; c < 8; ++c) rows[i * 11 + c] = (double)g.key[c];
        rows[i * 11 + 8] = g.launches; rows[i * 11 + 9] = g.ms; rows[i * 11 + 10] = g.work;
    }
    return GMRF_OK;
}

// Shader clock under load: _start launches clock_probe_kernel on a stream of its own and returns; the caller
// runs the load under test on its streams; _finish waits for the probe and returns, per sample interval, the clock in GHz
// (median over the probe's 8 waves) and the interval's start in ms since the first sample.  ghz / t_ms: n - 1 values.
struct ClockProbe { int device; hipStream_t st; unsigned long long* d; int n; };
gmrf_status gmrf_test_clock_probe_start(int32_t device, int32_t n, int32_t sleeps, void** probe) {
    if (!probe || n < 2 || n > 100000 || sleeps < 1 || sleeps > 64) return bad_shape("clock probe: 2 <= n <= 100000, 1 <= sleeps <= 64");
    HIPCHK(hipSetDevice(device));
    ClockProbe* p = new ClockProbe{device, nullptr, nullptr, n};
    auto drop = [&](hipError_t e) {                 // (no probe, stream or buffer is left behind on an error path: ADVICE r4)
        if (p->d) (void)hipFree(p->d);
        if (p->st) (void)hipStreamDestroy(p->st);
        delete p;
        g_last_error = std::string("clock probe: ") + hipGetErrorString(e);
        return GMRF_ERR_HIP;
    };
    hipError_t e = hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&p->d, sizeof(unsigned long long) * 2 * 8 * (size_t)n);
    if (e == hipSuccess) e = hipMemsetAsync(p->d, 0, sizeof(unsigned long long) * 2 * 8 * (size_t)n, p->st);
    if (e != hipSuccess) return drop(e);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(8), dim3(64), 0, p->st, p->d, n, sleeps);
    if ((e = hipGetLastError()) != hipSuccess) return drop(e);
    *probe = p;
    return GMRF_OK;
}
gmrf_status gmrf_test_clock_probe_finish(void* probe, double* ghz, double* t_ms) {
    ClockProbe* p = static_cast<ClockProbe*>(probe);
    if (!p || !ghz) return bad_shape("null pointer");
    std::vector<unsigned long long> h((size_t)2 * 8 * p->n);
    hipError_t e = hipSetDevice(p->device);
    if (e == hipSuccess) e = hipStreamSynchronize(p->st);
    if (e == hipSuccess) e = hipMemcpy(h.data(), p->d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipFree(p->d); (void)hipStreamDestroy(p->st);
        delete p;
        g_last_error = std::string("clock probe: ") + hipGetErrorString(e);
        return GMRF_ERR_HIP;
    }
    for (int i = 0; i + 1 < p->n; ++i) {
        double v[8];
        for (int b = 0; b < 8; ++b) {
            const unsigned long long* a = &h[((size_t)b * p->n + i) * 2];
            const double dc = (double)(a[2] - a[0]), dr = (double)(a[3] - a[1]);
            v[b] = dr > 0 ? dc / dr * 0.1 : 0.0;
        }
        std::sort(v, v + 8);
        ghz[i] = 0.5 * (v[3] + v[4]);
        if (t_ms) t_ms[i] = (double)(h[(size_t)i * 2 + 1] - h[1]) * 1e-5;
    }
    (void)hipFree(p->d); (void)hipStreamDestroy(p->st);
    delete p;
    return GMRF_OK;
}

gmrf_status gmrf_test_gemm_rate(int32_t device, int64_t M, int64_t N, int64_t K, int32_t transB, int32_t tri_flags,
                                int32_t lower_only, int32_t batch, int32_t big, int32_t reps, double* ms_per_launch) {
    if (M % 64 || N % 64 || K % 16 || batch < 1 || reps < 1) return bad_shape("gemm rate sizes");
    HIPCHK(hipSetDevice(device));
    const int64_t ld = std::max(std::max(M, N), K);
    const int64_t pm = ld * ld;
    double *dA, *dB, *dC;
    HIPCHK(hipMalloc(&dA, sizeof(double) * pm * batch));
    HIPCHK(hipMalloc(&dB, sizeof(double) * pm * batch));
    HIPCHK(hipMalloc(&dC, sizeof(double) * pm * batch));
    const int64_t tot = pm * batch;
    hipLaunchKernelGGL(fill_normals_panel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, nullptr, dA, tot, 1 << 30,
                       1 << 30, 1, 1, 1ull, 0, 0);
    hipLaunchKernelGGL(fill_normals_panel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, nullptr, dB, tot, 1 << 30,
                       1 << 30, 1, 1, 2ull, 0, 0);
    HIPCHK(hipMemset(dC, 0, sizeof(double) * pm * batch));
    GemmArgs g;
    g.A = dA; g.B = dB; g.C = dC; g.lda = g.ldb = g.ldc = ld;
    g.strideA = g.strideB = g.strideC = 0;
    g.pA = g.pB = g.pC = pm; g.nb1 = 1;
    g.D = nullptr; g.ldd = 0; g.pD = 0;
    g.M = (int)M; g.N = (int)N; g.K = (int)K; g.tri = tri_flags; g.lower_only = lower_only;
    g.alpha = 1.0; g.beta = 0.0; g.stamps = nullptr;
    HIPCHK(gemm_init());
    HIPCHK(gemm_dma_init());
    const int saved = gemm_big_policy(), saved_dma = gemm_dma_policy(), saved_force = gemm_dma_force();
    gemm_big_policy() = big == 1 ? 1 : (big == 2 ? 0 : 2);       // big = 2: the model's choice
    // big = 3 / 4 / 5: the LDS-DMA kernel with 64 x 64 / 128 x 64 / 64 x 128 tiles; 6: the production choice (all policies at their defaults)
    gemm_dma_policy() = big >= 3 ? 2 : 0;
    gemm_dma_force() = (big >= 3 && big <= 5) ? big - 2 : 0;
    if (big == 6) gemm_big_policy() = 0;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    hipError_t le = hipSuccess;
    for (int r = 0; r < 3 && le == hipSuccess; ++r) le = launch_gemm(nullptr, false, transB == 0, g, batch);
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps && le == hipSuccess; ++r) le = launch_gemm(nullptr, false, transB == 0, g, batch);
    HIPCHK(hipEventRecord(e1, nullptr));
    gemm_big_policy() = saved; gemm_dma_policy() = saved_dma; gemm_dma_force() = saved_force;
    HIPCHK(le);
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return GMRF_OK;
}

gmrf_status gmrf_test_potrf_tile(int32_t device, double* tile64, double* inv64, int32_t* info) {
    HIPCHK(hipSetDevice(device));
    double *dS, *dL, *dX;
    int* dinfo;
    HIPCHK(hipMalloc(&dS, sizeof(double) * 4096));
    HIPCHK(hipMalloc(&dL, sizeof(double) * 4096));
    HIPCHK(hipMalloc(&dX, sizeof(double) * 4096));
    HIPCHK(hipMalloc(&dinfo, sizeof(int)));
    HIPCHK(hipMemset(dinfo, 0, sizeof(int)));
    HIPCHK(hipMemcpy(dS, tile64, sizeof(double) * 4096, hipMemcpyHostToDevice));
    auto kern = potrf_tile_kernel;
    const dim3 tpb(256);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_TILE_LDS);
    unsigned long long* dstamps;
    HIPCHK(hipMalloc(&dstamps, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(dstamps, 0, 64 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(kern, dim3(1), tpb, POTRF_TILE_LDS, nullptr, dS, dL, dX, dinfo, (unsigned long long*)nullptr);
    HIPCHK(hipDeviceSynchronize());
    {   // timing: 200 back-to-back launches without stamps, then one stamped launch
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 200; ++i)
            hipLaunchKernelGGL(kern, dim3(1), tpb, POTRF_TILE_LDS, nullptr, dS, dL, dX, dinfo, (unsigned long long*)nullptr);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        g_tile_us = ms * 1e3 / 200.0;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipLaunchKernelGGL(kern, dim3(1), tpb, POTRF_TILE_LDS, nullptr, dS, dL, dX, dinfo, dstamps);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(g_tile_stamps, dstamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    hipFree(dstamps);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(tile64, dL, sizeof(double) * 4096, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(inv64, dX, sizeof(double) * 4096, hipMemcpyDeviceToHost));
    int hi = 0;
    HIPCHK(hipMemcpy(&hi, dinfo, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = hi;
    hipFree(dS); hipFree(dL); hipFree(dX); hipFree(dinfo);
    return GMRF_OK;
}

// diagnostics of the last gmrf_test_potrf_tile: out[0] = us per launch, out[1..17] = s_memtime stamps (cycles, relative)
gmrf_status gmrf_test_tile_timing(double* out, int32_t n) {
    if (!out || n < 18) return bad_shape("need 18 outputs");
    out[0] = g_tile_us;
    for (int i = 0; i < 17; ++i) out[1 + i] = (double)(g_tile_stamps[i] - g_tile_stamps[15]);
    if (n >= 24) for (int i = 0; i < 3; ++i) out[18 + i] = (double)(g_tile_stamps[20 + i] - g_tile_stamps[15]);
    if (n >= 30) for (int i = 0; i < 6; ++i) out[24 + i] = (double)(g_tile_stamps[24 + i] - g_tile_stamps[24]);   // fused step phases (gmrf_test_potrf_block)
    // round 5: per panel p (up to 8) three stamps of its owner wave: its column chain begins / is done / its columns are stored
    if (n >= 62) for (int i = 0; i < 8; ++i) out[54 + i] = g_tile_stamps[56 + i] ? (double)(g_tile_stamps[56 + i] - g_tile_stamps[15]) : -1.0;   // panel i: its owner has applied every earlier column
    if (n >= 54) for (int i = 0; i < 24; ++i) out[30 + i] = g_tile_stamps[32 + i] ? (double)(g_tile_stamps[32 + i] - g_tile_stamps[15]) : -1.0;
    return GMRF_OK;
}

// s_memtime stamps of the chain workgroup of the last gmrf_test_potrf_block that ran the persistent form: out[0] = after tile 0,
// then per step j four stamps (flags seen, operands in LDS, tile updated, tile j + 1 published), relative to the first, in cycles
gmrf_status gmrf_test_persist_stamps(double* out, int32_t n) {
    if (!out || n < 1 || n > 128) return bad_shape("1 .. 128 outputs");
    // (slot 8 s + 5 holds a small number, not a time: the panel in which the next step's operands were seen ready)
    for (int i = 0; i < n; ++i)
        out[i] = g_persist_stamps[i] ? (g_persist_stamps[i] < 64 ? (double)g_persist_stamps[i] : (double)(g_persist_stamps[i] - g_persist_stamps[0])) : -1.0;
    return GMRF_OK;
}

// How often a factorisation of this handle saw the abort word of its persistent launches (a bounded wait gave up) and was
// repeated with the launch-per-step form (tests force it with GMRF_PERSIST_SPIN_MS=0).
gmrf_status gmrf_test_persist_aborts(gmrf_handle* h, int32_t* n) {
    if (!h || !n) return bad_shape("null argument");
    *n = h->persist_aborts;
    return GMRF_OK;
}

// The budget of CUs for persistent launches on a device with `cus` CUs, host only: `n` handles claim demands[i] CUs one after
// the other; granted[i] = 1 if the claim fitted beside the earlier ones (gmrf_handle::persist_cus), 0 if it was refused.
gmrf_status gmrf_test_persist_budget(int32_t cus, int32_t n, const int32_t* demands, int32_t* granted) {
    if (n < 0 || (n > 0 && (!demands || !granted))) return bad_shape("null argument");
    std::map<const void*, int> claims;
    for (int i = 0; i < n; ++i) granted[i] = persist_budget_claim(claims, (const void*)(demands + i), demands[i], cus, 0) ? 1 : 0;
    return GMRF_OK;
}

gmrf_status gmrf_test_potrf_block(int32_t device, int64_t bs, double* S, double* Linv, int32_t* info) {
    if (bs % 64 || next_pow2(bs / 64) != bs / 64) return bad_shape("bs must be 64 * 2^p");
    gmrf_handle* h = nullptr;
    GCHK(gmrf_bt_create(device, nullptr, &h));
    h->N = 1; h->n = bs; h->bs = bs; h->bsp = bs; h->n_pad = bs; h->B = 1;
    gmrf_status s = set_layout_dense(h);
    if (s == GMRF_OK) s = alloc_factor(h);
    unsigned long long* dst = nullptr;
    if (s == GMRF_OK && hipMalloc(&dst, 128 * sizeof(unsigned long long)) == hipSuccess) {
        (void)hipMemset(dst, 0, 128 * sizeof(unsigned long long));
        h->dbg_stamps = dst;
    }
    if (s == GMRF_OK) {
        hipError_t e = hipMemcpyAsync(h->d_S, S, sizeof(double) * bs * bs, hipMemcpyHostToDevice, h->stream);
        persist_plan(h);
        if (e == hipSuccess) s = potrf_block(h, h->d_S, h->d_L, h->d_Linv, h->d_T, 1);
        if (s == GMRF_OK) {
            int hi = 0;
            (void)hipMemcpyAsync(&hi, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream);
            (void)hipMemcpyAsync(S, h->d_L, sizeof(double) * bs * bs, hipMemcpyDeviceToHost, h->stream);
            (void)hipMemcpyAsync(Linv, h->d_Linv, sizeof(double) * bs * bs, hipMemcpyDeviceToHost, h->stream);
            if (hipStreamSynchronize(h->stream) != hipSuccess) s = GMRF_ERR_HIP;
            if (info) *info = hi;
            if (dst) (void)hipMemcpy(g_tile_stamps + 24, dst, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (dst) (void)hipMemcpy(g_persist_stamps, dst, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        }
    }
    if (dst) (void)hipFree(dst);
    gmrf_bt_destroy(h);
    return s;
}

gmrf_status gmrf_test_mfma_f64_rate(int32_t device, double* tflops) {
    HIPCHK(hipSetDevice(device));
    double* d;
    HIPCHK(hipMalloc(&d, 64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const int iters = 20000, blocks = 256 * 2;
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, 100);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters);
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    hipFree(d); hipEventDestroy(a); hipEventDestroy(b);
    return GMRF_OK;
}


// Micro-benchmark battery; fills out[0..n) (see tools/microbench.py for the meaning).
// The host-only part of the symbolic phase of gmrf_bt_factor_csc (block split + band check, zero structure of the coupling
// blocks, row pointers, tile plan of the sparse C = B X^T) without a device: what the sanitizer build of the host side
// (make libgmrf_hip_asan.so, tests/test_host_logic.py) drives.  out[8] = {cmin, rmax, entries, most entries in a row of a
// lower block, sparse route?, tile plan?, LDS entry capacity of the plan, checksum of the plan's arrays}.
gmrf_status gmrf_test_symbolic_csc(int64_t n, int64_t n_blocks, const int64_t* colptr, const int64_t* rowval, int32_t index_base,
                                   int64_t* out8) {
    if (!colptr || !rowval || !out8) return bad_shape("null pointer");
    if (n <= 0 || n_blocks <= 0 || n % n_blocks != 0) return bad_shape("n must be a positive multiple of N_blocks");
    const int64_t bs = n / n_blocks, bsp = 64 * next_pow2((bs + 63) / 64);
    std::vector<std::vector<HostEntry>> dg, lo;
    GCHK(split_csc(n, n_blocks, bs, colptr, rowval, index_base, dg, lo));
    SymbolicPlan sp;
    build_symbolic(n_blocks, bsp, dg, lo, false, sp);
    uint64_t sum = 1469598103934665603ull;
    auto mix = [&sum](uint64_t v) { sum = (sum ^ v) * 1099511628211ull; };
    for (uint64_t k : sp.keys) mix(k);
    for (int64_t v : sp.src) mix((uint64_t)v);
    for (int v : sp.rowptr) mix((uint64_t)(uint32_t)v);
    for (int v : sp.uptr) mix((uint64_t)(uint32_t)v);
    for (int v : sp.gtiles) mix((uint64_t)(uint32_t)v);
    for (int v : sp.ucols) mix((uint64_t)(uint32_t)v);
    for (uint16_t v : sp.lidx) mix(v);
    for (int64_t v : sp.first) mix((uint64_t)v);
    out8[0] = sp.cmin; out8[1] = sp.rmax; out8[2] = (int64_t)sp.keys.size(); out8[3] = sp.max_row; out8[4] = sp.sparse_b;
    out8[5] = sp.bxt_ok; out8[6] = sp.ecap; out8[7] = (int64_t)(sum >> 1);
    return GMRF_OK;
}

gmrf_status gmrf_test_microbench(int32_t device, double* out, int32_t n) {
    if (!out || n < 16) return bad_shape("need 16 outputs");
    HIPCHK(hipSetDevice(device));
    unsigned long long* d_t; double* d_s; int* d_i;
    HIPCHK(hipMalloc(&d_t, 64)); HIPCHK(hipMalloc(&d_s, 64)); HIPCHK(hipMalloc(&d_i, 64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    unsigned long long ht[2];
    float ms = 0.f;
    auto timed = [&](auto launch, double flops, int idx) -> gmrf_status {
        launch(); HIPCHK(hipDeviceSynchronize());              // warm
        HIPCHK(hipEventRecord(a, nullptr)); launch(); HIPCHK(hipEventRecord(b, nullptr));
        HIPCHK(hipEventSynchronize(b)); HIPCHK(hipEventElapsedTime(&ms, a, b));
        HIPCHK(hipMemcpy(ht, d_t, 16, hipMemcpyDeviceToHost));
        out[idx] = flops / (ms * 1e-3) / 1e12;                 // TFLOP/s
        out[idx + 1] = (double)ht[0] / (double)ht[1] * 0.1;    // shader clock GHz inside the loop
        return GMRF_OK;
    };
    const int it = 20000;
    // 0,1: MFMA f64, 1 wave/SIMD (256 blocks), 4 accumulators
    GCHK(timed([&] { hipLaunchKernelGGL((mb_mfma_f64<4>), dim3(256), dim3(256), 0, nullptr, d_t, d_s, it); },
               256.0 * 4 * it * 4 * 2048.0, 0));
    // 2,3: MFMA f64, 2 waves/SIMD
    GCHK(timed([&] { hipLaunchKernelGGL((mb_mfma_f64<4>), dim3(512), dim3(256), 0, nullptr, d_t, d_s, it); },
               512.0 * 4 * it * 4 * 2048.0, 2));
    // 4,5: MFMA f64, 1 wave/SIMD, single dependent accumulator
    GCHK(timed([&] { hipLaunchKernelGGL((mb_mfma_f64<1>), dim3(256), dim3(256), 0, nullptr, d_t, d_s, it); },
               256.0 * 4 * it * 1 * 2048.0, 4));
    // 6,7: VALU f64 FMA, 2 waves/SIMD, 8 chains
    GCHK(timed([&] { hipLaunchKernelGGL(mb_valu_f64, dim3(512), dim3(256), 0, nullptr, d_t, d_s, it * 4); },
               512.0 * 256 * (it * 4.0) * 8 * 2.0, 6));
    // 8,9: one workgroup only (light load): MFMA loop -> cycles per MFMA and clock
    GCHK(timed([&] { hipLaunchKernelGGL((mb_mfma_f64<4>), dim3(1), dim3(64), 0, nullptr, d_t, d_s, it); },
               1.0 * it * 4 * 2048.0, 8));
    out[10] = (double)ht[0] / (it * 4.0);                       // shader cycles per MFMA, one wave
    // 11: empty-kernel launch cadence (us per launch, back to back on one stream)
    hipLaunchKernelGGL(mb_null, dim3(1), dim3(64), 0, nullptr, d_i);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(a, nullptr));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(mb_null, dim3(1), dim3(64), 0, nullptr, d_i);
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b)); HIPCHK(hipEventElapsedTime(&ms, a, b));
    out[11] = ms * 1e3 / 2000.0;
    // 12: same through a captured graph of 2000 nodes
    {
        hipStream_t st; HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipGraph_t g; hipGraphExec_t ge;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(mb_null, dim3(1), dim3(64), 0, st, d_i);
        HIPCHK(hipStreamEndCapture(st, &g));
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIPCHK(hipGraphLaunch(ge, st)); HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipEventRecord(a, st)); HIPCHK(hipGraphLaunch(ge, st)); HIPCHK(hipEventRecord(b, st));
        HIPCHK(hipEventSynchronize(b)); HIPCHK(hipEventElapsedTime(&ms, a, b));
        out[12] = ms * 1e3 / 2000.0;
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
    }
    // 13,14: light kernel right after 200 ms of idling (clock ramp check)
    {
        struct timespec ts = {0, 200000000};
        nanosleep(&ts, nullptr);
        HIPCHK(hipEventRecord(a, nullptr));
        hipLaunchKernelGGL((mb_mfma_f64<4>), dim3(1), dim3(64), 0, nullptr, d_t, d_s, 2000);
        HIPCHK(hipEventRecord(b, nullptr));
        HIPCHK(hipEventSynchronize(b)); HIPCHK(hipEventElapsedTime(&ms, a, b));
        HIPCHK(hipMemcpy(ht, d_t, 16, hipMemcpyDeviceToHost));
        out[13] = (double)ht[0] / (double)ht[1] * 0.1;
        out[14] = ms * 1e3;
    }
    out[15] = 0;
    hipFree(d_t); hipFree(d_s); hipFree(d_i); hipEventDestroy(a); hipEventDestroy(b);
    return GMRF_OK;
}

gmrf_status gmrf_test_hbm_rate(int32_t device, int64_t bytes, double* gbps) {
    HIPCHK(hipSetDevice(device));
    double2* src;
    double* d;
    HIPCHK(hipMalloc(&src, bytes));
    HIPCHK(hipMalloc(&d, 64));
    HIPCHK(hipMemset(src, 0, bytes));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const int64_t n16 = bytes / 16;
    hipLaunchKernelGGL(hbm_read_kernel, dim3(2048), dim3(256), 0, nullptr, src, n16, d);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(a, nullptr));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(hbm_read_kernel, dim3(2048), dim3(256), 0, nullptr, src, n16, d);
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *gbps = 5.0 * (double)bytes / (ms * 1e-3) / 1e9;
    hipFree(src); hipFree(d); hipEventDestroy(a); hipEventDestroy(b);
    return GMRF_OK;
}

}  // extern "C"
