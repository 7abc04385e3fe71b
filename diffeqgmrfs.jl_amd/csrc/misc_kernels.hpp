// Memory-bound helper kernels of the hot path: sparse-block scatter (K0), right-hand-side
// panel pack/unpack, Philox normals, CSR SpMV/SpMM (K6) and the variance accumulators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmrf {

// ------------------------------------------------------------------------------- K0
// Scatter the stored entries of one sparse block into a zeroed dense row-major block.
// Replaces `Array(A[rows, cols])` of /root/reference/src/tridiagonal_cholesky.jl:67,73,76.
// Entries are (row << 32 | col) keys local to the block.
// blockIdx.y = problem: values at vals + y * pvals, destination block at dst + y * pdst.
// add != 0: dst += value (the diagonal block lands on top of -C C^T; every key occurs once).
__global__ void scatter_block(const uint64_t* __restrict__ keys, const double* __restrict__ vals,
                              int64_t first, int64_t count, double* __restrict__ dst, int64_t ld,
                              int64_t pvals, int64_t pdst, int add) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t key = keys[first + i];
    const int64_t r = (int64_t)(key >> 32), c = (int64_t)(key & 0xffffffffu);
    double* d = dst + (int64_t)blockIdx.y * pdst + r * ld + c;
    const double v = vals[(int64_t)blockIdx.y * pvals + first + i];
    *d = add ? *d + v : v;
}

// Zero `count` doubles (a multiple of 2, 16-byte aligned) at dst + y * pdst for every problem y.
// (hipMemset2DAsync does the same at a quarter of the rate.)
__global__ void zero_rows(double* __restrict__ dst, int64_t count, int64_t pdst) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i < count) *reinterpret_cast<v2d*>(dst + (int64_t)blockIdx.y * pdst + i) = (v2d){0.0, 0.0};
}

// Identity on the padding rows [bs, bsp) of a padded diagonal block.
__global__ void pad_identity(double* __restrict__ dst, int64_t ld, int bs, int bsp, int64_t pdst) {
    const int i = bs + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bsp) dst[(int64_t)blockIdx.y * pdst + (int64_t)i * ld + i] = 1.0;
}

// ------------------------------------------------------------------------------- factor transport
// Lower-triangular 64 x 64 TILES of the block inverses Linv_i <-> a contiguous buffer (what travels over xGMI
// when a factor is shared: the tiles above the block diagonal are zero and never read, so a bsp x bsp block
// of nt x nt tiles moves as nt (nt + 1) / 2 of them: 136 of 256 for bsp = 1024).  Tile (r, c), c <= r, of block i
// lands at ((i - i0) * ntri + r (r + 1) / 2 + c) * 4096 of the problem's segment, row-major inside the tile.
// grid (ntri * blocks, problems), 256 threads: a thread moves 16 doubles of its tile as 16-byte pieces.
template <bool PACK>
__global__ __launch_bounds__(256) void linv_tiles_copy(double* __restrict__ X, int64_t ld, int64_t blk_stride,
                                                       int64_t pX, double* __restrict__ buf, int64_t pbuf, int ntri) {
    const int b = (int)blockIdx.x / ntri, tile = (int)blockIdx.x % ntri;
    int r = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
    while (r * (r + 1) / 2 > tile) --r;
    while ((r + 1) * (r + 2) / 2 <= tile) ++r;
    const int c = tile - r * (r + 1) / 2;
    double* x = X + (int64_t)blockIdx.y * pX + (int64_t)b * blk_stride + (int64_t)r * 64 * ld + (int64_t)c * 64;
    double* q = buf + (int64_t)blockIdx.y * pbuf + ((int64_t)b * ntri + tile) * 4096;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = (i * 256 + t) * 2, row = e >> 6, col = e & 63;       // a wave covers two 512-byte tile rows
        if (PACK) *reinterpret_cast<v2d*>(q + e) = *reinterpret_cast<const v2d*>(x + (int64_t)row * ld + col);
        else *reinterpret_cast<v2d*>(x + (int64_t)row * ld + col) = *reinterpret_cast<const v2d*>(q + e);
    }
}

// Representation tag of a packed transport image (gmrf_handle::xsplit of the SENDER when it packed): two doubles per problem
// segment, {xsplit, 0x474d5246}.  The receiver's unpack leaves xsplit + 1 (or -1 if the ranges of one factor disagree / the tag is
// not one) in a device word that gmrf_bt_adopt_commit reads: the layout record can be stale, the image cannot.
constexpr double PACK_TAG_MAGIC = 1196249670.0;     // "GMRF"
__global__ void pack_tag_write(double* __restrict__ buf, int64_t pbuf, double xsplit) {
    double* q = buf + (int64_t)threadIdx.x * pbuf;
    q[0] = xsplit; q[1] = PACK_TAG_MAGIC;
}
// (one thread walks the tags of all `nprob` segments: in the all-gather form the problems of one image come from different ranks,
//  and a rank whose inverses are in the other representation, or a corrupt segment, must not pass because problem 0 is fine:
//  ADVICE r4)
__global__ void pack_tag_read(const double* __restrict__ buf, int64_t pbuf, int nprob, int* __restrict__ seen) {
    int code = *seen;
    for (int p = 0; p < nprob; ++p) {
        const double* q = buf + (int64_t)p * pbuf;
        const double v = q[0];
        const int c = (q[1] == PACK_TAG_MAGIC && v >= 0.0 && v < 1.0e6 && v == (double)(int)v) ? (int)v + 1 : -1;
        code = (code == 0 || code == c) ? c : -1;
    }
    *seen = code;
}

// ------------------------------------------------------------------------------- panels
// user matrix (column-major n x k, leading dimension ld) <-> padded panel P[rhs][n_pad]
// blockIdx.y = problem p: columns [p*k, (p+1)*k) of the user matrix <-> panel p (kp * n_pad doubles)
__global__ void pack_panel(const double* __restrict__ src, int64_t ld, double* __restrict__ P,
                           int64_t n_pad, int bs, int bsp, int nblk, int k, int kp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)kp * n_pad;
    if (idx >= total) return;
    const int64_t r = idx / n_pad, j = idx % n_pad;
    const int64_t blk = j / bsp, off = j % bsp;
    double v = 0.0;
    if (r < k && off < bs) v = src[((int64_t)blockIdx.y * k + r) * ld + blk * bs + off];
    P[(int64_t)blockIdx.y * total + idx] = v;
}

__global__ void unpack_panel(const double* __restrict__ P, int64_t n_pad, double* __restrict__ dst,
                             int64_t ld, int bs, int bsp, int64_t n, int k, int kp,
                             const double* __restrict__ mean) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (int64_t)k) return;
    const int64_t r = idx / n, j = idx % n;
    const int64_t blk = j / bs, off = j % bs;
    double v = P[(int64_t)blockIdx.y * kp * n_pad + r * n_pad + blk * bsp + off];
    if (mean) v += mean[(int64_t)blockIdx.y * n + j];
    dst[((int64_t)blockIdx.y * k + r) * ld + j] = v;
}

// ------------------------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al. 2011), key = seed, counter = (dof, sample id).
__device__ __host__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __host__ inline double philox_normal(uint64_t seed, uint64_t dof, uint64_t sample) {
    uint32_t c[4] = {(uint32_t)dof, (uint32_t)(dof >> 32), (uint32_t)sample, (uint32_t)(sample >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t a = ((uint64_t)c[1] << 32) | c[0];
    const uint64_t b = ((uint64_t)c[3] << 32) | c[2];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);   // (0,1)
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

// Fill the padded panel rows [0,k) with normals of samples first_id.. (padding stays zero).
// problem p (blockIdx.y) draws the sample ids first_id + p * id_stride + r
__global__ void fill_normals_panel(double* __restrict__ P, int64_t n_pad, int bs, int bsp,
                                   int k, int kp, uint64_t seed, int64_t first_id, int64_t id_stride) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)kp * n_pad) return;
    const int64_t r = idx / n_pad, j = idx % n_pad;
    const int64_t blk = j / bsp, off = j % bsp;
    double v = 0.0;
    if (r < k && off < bs)
        v = philox_normal(seed, (uint64_t)(blk * bs + off), (uint64_t)(first_id + (int64_t)blockIdx.y * id_stride + r));
    P[(int64_t)blockIdx.y * kp * n_pad + idx] = v;
}

// ------------------------------------------------------------------------------- K7
// C = B * X^T with the lower block B kept SPARSE (src/tridiagonal_cholesky.jl:74 forms
// `A[block_idcs, prev_block_idcs] / L'`; a FEM / finite-difference coupling block has a handful of
// entries per row -- darcy256: 3 574 in 1024 x 1024 -- so the product is 2 nnz bs flop instead of
// 2 bs^3 and the kernel is bound by writing C):
//   C[r][c] = sum_{(r,j) in B} B[r][j] X[c][j]          X lower triangular, zeros stored above
// A thread owns one row r of B: it loads the row's (at most KM) entries once into registers, then
// walks cw columns c; for entry t the 64 lanes of a wave (consecutive r) read X[c][j_t(r)], which
// for a stencil coupling are consecutive addresses of one row of X.  No LDS, every load of the c
// loop is independent.  The lower blocks' entry lists are stored row by row (rowptr).
// 1-D grid of (bsp - cm) / cw * ceil(rm / 256) * problems workgroups.
struct BxtArgs {
    const int* rowptr;        // [bsp + 1], local row -> range in keys / vals
    const uint64_t* keys;     // row << 32 | col
    const double* vals;       // [problems][n_entries]
    int64_t n_entries;
    const double* X;          // previous block's inverse
    double* C;                // compact coupling block [rm][bsp - cm] (columns cm .. of the logical block)
    int64_t ld, ldc, pX, pC;  // row strides of X and C, problem strides
    const int* kst;           // staircase: row tile t (64 rows) of B is zero left of column cm + kst[t] (monotone)
    int cm, rm, bsp;
    int cw;                   // columns of C per workgroup (16, 32 or 64: enough workgroups for a lone problem)
};

template <int KM>
__global__ __launch_bounds__(256) void spmm_bxt(BxtArgs a) {
    // 1-D grid over (column chunk, row chunk, problem).  Workgroup ids go round-robin over the 8
    // XCDs: with a multiple of 8 problems id % 8 picks the problem group and the row chunks of one
    // column chunk follow each other on the SAME XCD, so the rows of X they share come from its L2
    // (PMC before: X fetched 2.6 times per launch).
    const int ncc = (a.bsp - a.cm) / a.cw, nrc = (a.rm + 255) / 256;
    const int nprob = (int)gridDim.x / (ncc * nrc);
    int cc, rc, prob;
    {
        const int groups = (nprob % 8 == 0) ? 8 : 1;
        const int xg = (int)blockIdx.x % groups, q = (int)blockIdx.x / groups;
        rc = q % nrc;
        cc = (q / nrc) % ncc;
        prob = xg + groups * (q / (nrc * ncc));
    }
    const int r = rc * 256 + (int)threadIdx.x;
    const int c0 = a.cm + cc * a.cw;
    const double* __restrict__ X = a.X + (int64_t)prob * a.pX;
    double* __restrict__ C = a.C + (int64_t)prob * a.pC;
    const double* __restrict__ vals = a.vals + (int64_t)prob * a.n_entries;
    int k0 = 0, len = 0;
    if (r < a.rm) { k0 = a.rowptr[r]; len = a.rowptr[r + 1] - k0; }
    int jt[KM];
    double vt[KM];
#pragma unroll
    for (int t = 0; t < KM; ++t) {
        const bool ok = t < len;
        jt[t] = ok ? (int)(a.keys[k0 + t] & 0xffffffffu) : 0;
        vt[t] = ok ? vals[k0 + t] : 0.0;
    }
    int lm = len;                                  // longest row of the wave bounds the entry loop
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lm = max(lm, __shfl_xor(lm, off));
    lm = __builtin_amdgcn_readfirstlane(lm);
    // 16 columns at a time; the 256 x 16 results turn through LDS so that they leave as full
    // 128-byte lines (stores of 8 or 16 bytes a line apart cost one L2 transaction each and made
    // this kernel slower than the dense GEMM it replaces)
    __shared__ double ls[256 * 17];
    const int tid = (int)threadIdx.x;
    const int wr = tid >> 4, wc = tid & 15;        // write-out: 16 lanes per row, 16 rows per pass
    const int rbase = rc * 256;
    // C[r][c] = sum_j B[r][j] X[c][j] with X lower triangular vanishes for c < the row's first column:
    // the column groups left of the staircase of this row chunk are never written (they hold zeros
    // from the allocation) -- neither C nor the rows of X they would gather are touched
    const int cfirst = a.cm + (a.kst ? a.kst[rbase >> 6] : 0);
    for (int c = c0; c < c0 + a.cw; c += 16) {
        if (c + 16 <= cfirst) continue;
        const double* xr = X + (int64_t)c * a.ld;
        double acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.0;
#pragma unroll
        for (int t = 0; t < KM; ++t)
            if (t < lm) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u] = fma(vt[t], xr[(int64_t)u * a.ld + jt[t]], acc[u]);
            }
#pragma unroll
        for (int u = 0; u < 16; ++u) ls[tid * 17 + u] = acc[u];
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < 16; ++p) {
            const int rr = p * 16 + wr;
            if (rbase + rr < a.rm) C[(int64_t)(rbase + rr) * a.ldc + (c - a.cm) + wc] = ls[rr * 17 + wc];
        }
        __syncthreads();
    }
}

// The same product with the operands of a 64-row tile staged in LDS (the plan comes from the symbolic phase: per
// lower block and 64-row tile the sorted DISTINCT columns of its entries, and per entry its 16-bit index into that
// list -- a stencil tile of 64 rows touches ~200 of the 1024 columns).  spmm_bxt above reads X[c][j_t] once per
// (row, entry, column) through L1 (18 TB/s of L1 traffic per launch: the texture path is what bounds it); here a
// workgroup gathers X[c0 .. c0+15][distinct columns] ONCE per 16-column chunk (thread u = distinct column u: 16
// independent 8-byte loads, coalesced along u, issued while the previous chunk is being multiplied), writes them as
// swizzled 128-byte LDS rows [u][16] and then runs the multiply phase of csr_spmm_tiles_pad: lane (row, kq) owns the
// columns 4 kq .. 4 kq + 3 of the chunk, entries padded to multiples of 4 per row (aligned 8-byte index / 16-byte value
// reads), fixed CSR summation order -- the results are bitwise those of spmm_bxt.  C leaves as 128-byte lines.
// Round 4: a workgroup serves a GROUP of up to three row tiles of the block that meet (mostly) the same columns of X -- in a
// stencil block the tiles of the second and third mesh row of a block meet subsets of what the tile of the first mesh row above
// them meets -- so ONE gathered chunk is multiplied by up to 192 rows instead of 64 (darcy256: 388 gathered columns per chunk
// position become 195: half the L2 -> CU traffic the round-3 review measured, 620 MB per launch).  Every tile keeps its own
// staircase start (a chunk left of it is skipped for that tile) and its rows' entry order: bitwise the results of spmm_bxt.
struct BxtTileArgs {
    const int* rowptr;        // this block's [bsp + 1] row pointers (absolute entry indices)
    const uint16_t* lidx;     // per entry (absolute index): position of its column in its GROUP's list
    const double* vals;       // [problems][n_entries]
    int64_t n_entries;
    const int* gtiles;        // this block's [ng][3] row tiles of every group (-1: none)
    int ng;                   // groups of this block
    const int* uptr;          // this block's [ng + 1] offsets into ucols
    const int* ucols;         // distinct columns (0-based inside the block), ascending per group
    const double* X;
    double* C;
    int64_t ld, ldc, pX, pC;
    const int* kst;
    int cm, rm, bsp;
    int nch;                  // 16-column chunks per workgroup
    int ecap;                 // padded entries per group (LDS capacity, multiple of 8)
    int skip_dead;            // leave a row's multiply loop at the first turn that only meets zero rows of the staged image
    // round 5: the launch also zeroes the rows of the Schur block that the product S = -C C^T will not write (zero_rows was a
    // launch of its own per block, 5 us + a boundary on one problem's chain): `zwgs` more workgroups per problem behind the others
    int nprob;                // problems of the launch (the grid no longer says)
    double* zdst; int64_t zcount, zpdst; int zwgs;      // zdst == nullptr / zwgs == 0: nothing to zero
    // ... and (one problem) scatters the diagonal block D_i into the zeroed block: srowptr != nullptr -- zdst is then the WHOLE
    // block (zcount = bsp^2), a zeroing workgroup owns whole rows and writes the entries of its rows behind its zeros
    const int* srowptr;       // [bsp + 1] positions in skeys / svals of the rows of D_i
    const uint64_t* skeys;    // row << 32 | col
    const double* svals;
};

constexpr int BXT_UCAP = 256;             // distinct columns per group (one per thread)
constexpr int BXT_GT = 3;                 // row tiles per group

inline size_t bxt_tile_lds_bytes(int ecap) {
    return (size_t)BXT_UCAP * 16 * 8 + (size_t)ecap * 10 + BXT_GT * 128 * 8 + (size_t)BXT_UCAP * 4 + (size_t)BXT_GT * (2 * 64 + 2) * 4 + 16;
}

__global__ __launch_bounds__(256, 3) void spmm_bxt_tiles(BxtTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xs = smem;                                       // [BXT_UCAP][16], swizzled
    double* vs = xs + BXT_UCAP * 16;                         // [ecap + 3 * 128]  (row r of tile slot s skewed by 2 r + 128 s doubles, see csr_spmm_tiles_pad)
    uint16_t* ls = reinterpret_cast<uint16_t*>(vs + a.ecap + BXT_GT * 128); // [ecap]
    int* uc = reinterpret_cast<int*>(ls + a.ecap);           // [BXT_UCAP]
    int* rp = uc + BXT_UCAP;                                 // [3][65]: a row's first entry, relative to its tile's first
    int* pp = rp + BXT_GT * 65;                              // [3][65]: a row's first PADDED entry in the group's staged list
    const int t = threadIdx.x;
    const int W = a.bsp - a.cm;
    const int ncg = (W / 16 + a.nch - 1) / a.nch;
    const int nprob = a.nprob;
    if ((int)blockIdx.x >= ncg * a.ng * nprob) {
        // zeroing role: workgroup z of problem p clears its share of zdst[p]
        const int zi = (int)blockIdx.x - ncg * a.ng * nprob;
        const int p = zi / a.zwgs, z = zi % a.zwgs;
        double* d = a.zdst + (int64_t)p * a.zpdst;
        if (a.srowptr) {
            const int rpw = (a.bsp + a.zwgs - 1) / a.zwgs;                        // whole rows per workgroup
            const int ra = min(a.bsp, z * rpw), rb = min(a.bsp, ra + rpw);
            // (a row up to the end of its diagonal tile: the product and the Cholesky read the lower tiles only)
            for (int r = ra; r < rb; ++r) {
                const int cend = min(a.bsp, ((r >> 6) + 1) * 64);
                for (int c = 2 * t; c < cend; c += 512) *reinterpret_cast<v2d*>(d + (int64_t)r * a.ld + c) = (v2d){0.0, 0.0};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // (the zeros have arrived before an entry lands on one)
            __syncthreads();
            const int e0 = a.srowptr[ra], e1 = a.srowptr[rb];
            for (int e = e0 + t; e < e1; e += 256) {
                const uint64_t key = a.skeys[e];
                d[(int64_t)(key >> 32) * a.ld + (int64_t)(key & 0xffffffffu)] = a.svals[e];
            }
            return;
        }
        const int64_t per = ((a.zcount / 2 + a.zwgs - 1) / a.zwgs) * 2;           // doubles per workgroup (even)
        const int64_t lo = (int64_t)z * per, hi = min(a.zcount, lo + per);
        for (int64_t i = lo + 2 * t; i < hi; i += 512) *reinterpret_cast<v2d*>(d + i) = (v2d){0.0, 0.0};
        return;
    }
    int cg, g, prob;
    {
        const int groups = (nprob % 8 == 0) ? 8 : 1;
        const int xg = (int)blockIdx.x % groups, q = (int)blockIdx.x / groups;
        g = q % a.ng;
        cg = (q / a.ng) % ncg;
        prob = xg + groups * (q / (a.ng * ncg));
    }
    int tile[BXT_GT], cst[BXT_GT], e0s[BXT_GT];
    int nT = 0, cmin_g = 1 << 30;
#pragma unroll
    for (int s = 0; s < BXT_GT; ++s) {
        tile[s] = a.gtiles[g * BXT_GT + s];
        if (tile[s] >= 0) nT = s + 1;
        const int ts = tile[s] >= 0 ? tile[s] : 0;
        // first chunk of tile s: the one its staircase start lies in (uniform per workgroup)
        cst[s] = tile[s] >= 0 ? ((a.cm + (a.kst ? a.kst[ts] : 0)) / 16) * 16 : (1 << 30);
        e0s[s] = a.rowptr[ts * 64];
        cmin_g = min(cmin_g, cst[s]);
    }
    int cbeg = a.cm + cg * a.nch * 16;
    const int cend = min(a.cm + W, cbeg + a.nch * 16);
    cbeg = max(cbeg, cmin_g);
    if (cbeg >= cend) return;                                // whole column group left of every tile's staircase (uniform per workgroup)
    const double* __restrict__ X = a.X + (int64_t)prob * a.pX;
    double* __restrict__ C = a.C + (int64_t)prob * a.pC;
    const double* __restrict__ vals = a.vals + (int64_t)prob * a.n_entries;
    if (t < 64 * BXT_GT) {                                   // padded row starts: wave s scans tile slot s
        const int s = t >> 6, r = t & 63;
        if (s < nT && tile[s] >= 0) {
            const int r0 = tile[s] * 64;
            const int ea = a.rowptr[r0 + r] - e0s[s], eb = a.rowptr[r0 + r + 1] - e0s[s];
            int x = (eb - ea + 3) & ~3;                      // rows padded to turns of 4 entries (FEM coupling rows are short: 9 / 4 / 1 entries on the darcy mesh)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d, 64);
                if (r >= d) x += y;
            }
            rp[s * 65 + r] = ea; pp[s * 65 + r + 1] = x;     // (pp: inclusive scan inside the tile; the tiles' bases are added below)
            if (r == 63) rp[s * 65 + 64] = eb;
            if (r == 0) pp[s * 65] = 0;
        }
    }
    const int u0 = a.uptr[g];
    const int U = a.uptr[g + 1] - u0;
    const int ucol = (t < U) ? a.ucols[u0 + t] : -1;
    __syncthreads();
    int base[BXT_GT];                                        // first padded entry of tile slot s in the staged list
    base[0] = 0;
#pragma unroll
    for (int s = 1; s < BXT_GT; ++s) base[s] = base[s - 1] + ((s - 1 < nT && tile[s - 1] >= 0) ? pp[(s - 1) * 65 + 64] : 0);
    const int kq = t & 3, row = t >> 2;
#pragma unroll
    for (int s = 0; s < BXT_GT; ++s) {
        if (s < nT && tile[s] >= 0) {   // entries of this thread's row of tile s, 4 lanes per row, padded with (value 0, the row's first index)
            const int ea = rp[s * 65 + row], len = rp[s * 65 + row + 1] - ea;
            const int pa = base[s] + pp[s * 65 + row], plen = pp[s * 65 + row + 1] - pp[s * 65 + row];
            const uint16_t first = (len > 0) ? a.lidx[e0s[s] + ea] : (uint16_t)0;
            for (int j = kq; j < plen; j += 4) {
                const bool ok = j < len;
                ls[pa + j] = ok ? a.lidx[e0s[s] + ea + j] : first;
                vs[pa + j + 2 * row + 128 * s] = ok ? vals[e0s[s] + ea + j] : 0.0;
            }
        }
    }
    // (Round 4 tried TWO chunks of X in flight -- g0 / g1 alternating, a buffer requested again as soon as its chunk is in LDS --
    //  on the theory that a chunk's ~300-cycle multiply cannot cover the latency of the gather issued just before it: 112.9 ->
    //  118.4 us per launch of 64 problems in the same 1-stream trace, 144 VGPRs.  The gather is not latency-bound; not kept.)
    double gx[16];
    auto gather = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)      // X = L^-1 is lower triangular: entries right of the diagonal are exact zeros, not fetched
            gx[cc] = (ucol >= 0 && ucol <= c0 + cc) ? X[(int64_t)(c0 + cc) * a.ld + ucol] : 0.0;
    };
    gather(cbeg);
    for (int c0 = cbeg; c0 < cend; c0 += 16) {
        if (t < U) {
            // row t of the staged image: slot s (columns 2 s, 2 s + 1 of the chunk = lane kq = s >> 1, half s & 1 ->
            // logical slot (half << 2 | kq)) at position logical ^ swizzle(t); swizzle keeps both the 16 rows a
            // ds_write_b128 serves together and the 4 rows x 4 lanes of a read on distinct banks
            const int swz = (((t >> 1) & 1) << 2) ^ ((t >> 2) & 3);
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int logical = ((s2 & 1) << 2) | (s2 >> 1);
                *reinterpret_cast<v2d*>(xs + t * 16 + 2 * (logical ^ swz)) = (v2d){gx[2 * s2], gx[2 * s2 + 1]};
            }
        }
        // (first chunk: also the staged entries.)  live = distinct columns <= c0 + 15, i.e. staged rows that are not all zero:
        // the list is ascending and so are a row's entries, so a turn whose first index is >= live multiplies zeros only
        int live = 1 << 30;
        if (a.skip_dead) live = __syncthreads_count(t < U && ucol <= c0 + 15);
        else __syncthreads();
        if (c0 + 16 < cend) gather(c0 + 16);                 // in flight while this chunk is multiplied
#pragma unroll
        for (int s = 0; s < BXT_GT; ++s) {
            if (s >= nT || tile[s] < 0 || c0 < cst[s]) continue;      // (uniform per workgroup: this tile is zero left of its staircase)
            v2d a01 = (v2d){0.0, 0.0}, a23 = (v2d){0.0, 0.0};
            const int eb = base[s] + pp[s * 65 + row], ee = base[s] + pp[s * 65 + row + 1];
            for (int e = eb; e < ee; e += 4) {
                const uint2 q = *reinterpret_cast<const uint2*>(ls + e);
                if ((int)(q.x & 0xffffu) >= live) break;
                const double* ve = vs + e + 2 * row + 128 * s;
                const v2d v01 = *reinterpret_cast<const v2d*>(ve), v23 = *reinterpret_cast<const v2d*>(ve + 2);
                const int ii[4] = {(int)(q.x & 0xffffu), (int)(q.x >> 16), (int)(q.y & 0xffffu), (int)(q.y >> 16)};
                const double vv[4] = {v01.x, v01.y, v23.x, v23.y};
                v2d xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int swz = (((ii[u] >> 1) & 1) << 2) ^ ((ii[u] >> 2) & 3);
                    const int o = (ii[u] << 4) + 2 * (kq ^ swz);                 // in doubles: row * 16 + 2 * slot(kq, half 0)
                    xa[u] = *reinterpret_cast<const v2d*>(xs + o);
                    xb[u] = *reinterpret_cast<const v2d*>(xs + (o ^ 8));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a01.x = fma(vv[u], xa[u].x, a01.x); a01.y = fma(vv[u], xa[u].y, a01.y);
                    a23.x = fma(vv[u], xb[u].x, a23.x); a23.y = fma(vv[u], xb[u].y, a23.y);
                }
            }
            double* cp = C + (int64_t)(tile[s] * 64 + row) * a.ldc + (c0 - a.cm) + 4 * kq;
            *reinterpret_cast<v2d*>(cp) = a01;
            *reinterpret_cast<v2d*>(cp + 2) = a23;
        }
        __syncthreads();
    }
}

// rows x cols rectangle of doubles (cols a multiple of 2, 16-byte aligned rows), one wave per row, blockIdx.y = problem
__global__ __launch_bounds__(256) void copy_rect(const double* __restrict__ src, int64_t lds, int64_t ps, double* __restrict__ dst,
                                                 int64_t ldd, int64_t pd, int rows, int cols) {
    const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (row >= rows) return;
    const double* s = src + (int64_t)blockIdx.y * ps + (int64_t)row * lds;
    double* d = dst + (int64_t)blockIdx.y * pd + (int64_t)row * ldd;
    for (int c = 2 * lane; c < cols; c += 128) *reinterpret_cast<v2d*>(d + c) = *reinterpret_cast<const v2d*>(s + c);
}

// ------------------------------------------------------------------------------- K6
// CSR SpMV/SpMM  Y = S X, X and Y stored one right-hand side after the other (strides
// ldx/ldy).  Replaces SparseArrays' `Q * x` (scripts/solve_burger.jl:157-158,166,177 and the
// RBMC variance estimator).  G lanes cooperate on one row: column indices and values are
// read once, coalesced, and reused for every right-hand side; partial sums are combined by
// a butterfly inside the lane group.  VT = float stores fp32 values (config 5), the
// accumulation is fp64 either way.
template <typename VT, int G>
__global__ __launch_bounds__(256) void csr_spmm(const int64_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ colidx,
                                                const VT* __restrict__ vals, int64_t n_rows,
                                                const double* __restrict__ X, int64_t ldx,
                                                double* __restrict__ Y, int64_t ldy, int k) {
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (gid >= n_rows) return;            // whole lane group leaves together
    const int64_t p0 = rowptr[gid], p1 = rowptr[gid + 1];
    for (int r0 = 0; r0 < k; r0 += 4) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const int kr = min(4, k - r0);
        for (int64_t p = p0 + sub; p < p1; p += G) {
            const int64_t c = colidx[p];
            const double v = (double)vals[p];
            const double* x = X + (int64_t)r0 * ldx + c;
            s0 = fma(v, x[0], s0);
            if (kr > 1) s1 = fma(v, x[ldx], s1);
            if (kr > 2) s2 = fma(v, x[2 * ldx], s2);
            if (kr > 3) s3 = fma(v, x[3 * ldx], s3);
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, G);
            s1 += __shfl_xor(s1, off, G);
            s2 += __shfl_xor(s2, off, G);
            s3 += __shfl_xor(s3, off, G);
        }
        if (sub == 0) {
            double* y = Y + (int64_t)r0 * ldy + gid;
            y[0] = s0;
            if (kr > 1) y[ldy] = s1;
            if (kr > 2) y[2 * ldy] = s2;
            if (kr > 3) y[3 * ldy] = s3;
        }
    }
}

// SpMV with LDS-staged row tiles (one right-hand side).  A workgroup owns SPMV_ROWS consecutive rows
// = one contiguous range of entries.  Phase 1: thread t takes entries e0 + t, e0 + t + 256, ...: the
// colidx / vals reads are perfectly coalesced and every load of a thread is independent (a whole
// tile's worth of bytes in flight per workgroup -- the lane-group kernel above keeps one dependent
// chain rowptr -> entries -> x per row in flight and ran at 2 TB/s on a 31 M-entry matrix); the
// products v * x[col] go to LDS.  Phase 2: one thread per row adds its segment of the LDS image in
// entry order (fixed summation order) and writes y.  The host checks at creation that no tile has
// more than SPMV_CAP entries; otherwise the lane-group kernel is used.
constexpr int SPMV_ROWS = 64;
constexpr int SPMV_CAP = 2304;            // doubles of LDS per workgroup (18 KB: eight workgroups per CU)

template <typename VT, int ROWS = SPMV_ROWS>
__global__ __launch_bounds__(256) void csr_spmv_tiles(const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colidx,
                                                      const VT* __restrict__ vals, int64_t n_rows,
                                                      const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double prod[SPMV_CAP * (ROWS / SPMV_ROWS)];
    __shared__ int64_t rp[ROWS + 1];
    const int t = threadIdx.x;
    const int64_t ntiles = (n_rows + ROWS - 1) / ROWS;
    // a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher caps the grid at a few resident
    // workgroups per CU: 32 768 four-wave workgroups of ~1.3 us each were bound by the dispatch rate, not by memory)
    // software pipeline over the tiles of this workgroup: the entries (first pass of 1024) of the NEXT tile are requested
    // while the current tile's x values are gathered and its rows are summed
    int64_t tile = blockIdx.x;
    int64_t e0 = 0;
    int cnt = 0;
    int c[4];
    double v[4];
    auto fetch = [&](int64_t tl, int64_t& e0o, int& cnto, int (&co)[4], double (&vo)[4]) {
        const int64_t r0f = tl * ROWS;
        const int nrf = (int)min((int64_t)ROWS, n_rows - r0f);
        e0o = rowptr[r0f];
        cnto = (int)(rowptr[r0f + nrf] - e0o);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = t + 256 * i;
            const bool ok = e < cnto;
            co[i] = ok ? colidx[e0o + e] : 0;
            vo[i] = ok ? (double)vals[e0o + e] : 0.0;
        }
    };
    if (tile < ntiles) fetch(tile, e0, cnt, c, v);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * ROWS;
        const int nr = (int)min((int64_t)ROWS, n_rows - r0);
        if (t <= nr) rp[t] = rowptr[r0 + t];               // not needed before the row sums (the barrier below)
        double xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = x[c[i]];
        int64_t e0n = 0;
        int cntn = 0;
        int cn[4] = {0, 0, 0, 0};
        double vn[4] = {0.0, 0.0, 0.0, 0.0};
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, e0n, cntn, cn, vn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = t + 256 * i;
            if (e < cnt) prod[e] = v[i] * xv[i];
        }
        for (int base = 1024; base < cnt; base += 1024) {     // (tiles with more than 1024 entries: the rest, unpipelined)
            int c2[4];
            double v2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = base + t + 256 * i;
                const bool ok = e < cnt;
                c2[i] = ok ? colidx[e0 + e] : 0;
                v2[i] = ok ? (double)vals[e0 + e] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = base + t + 256 * i;
                if (e < cnt) prod[e] = v2[i] * x[c2[i]];
            }
        }
        __syncthreads();
        if (t < nr) {
            const int a = (int)(rp[t] - e0), b = (int)(rp[t + 1] - e0);
            double sacc = 0.0;
            for (int e = a; e < b; ++e) sacc += prod[e];
            y[r0 + t] = sacc;
        }
        __syncthreads();                                   // prod / rp are rewritten by the next tile
        e0 = e0n; cnt = cntn;
#pragma unroll
        for (int i = 0; i < 4; ++i) { c[i] = cn[i]; v[i] = vn[i]; }
    }
}

// SpMM with LDS-staged row tiles for right-hand sides stored NODE-MAJOR ("interleaved": X[col][rhs],
// row stride ldx >= k -- the k values one gathered column index needs are contiguous):
//     Y[r][:] = sum_e vals[e] * X[col[e]][:]
// A workgroup owns R consecutive rows (R = 64, 32 or 16, chosen by the host-side plan so that three
// workgroups fit a CU's LDS).  The plan (built once per matrix, like the symbolic phase of the
// factor) lists the DISTINCT columns of every tile and replaces each entry's column by its 16-bit
// index into that list.  Per tile: (1) the entries (local index, value) are staged in LDS with
// coalesced independent loads; (2) per chunk of 16 right-hand sides the distinct rows of X are
// gathered ONCE -- a FEM / finite-difference tile of 64 rows touches ~3.4 rows of X per output row
// instead of ~15, each gathered piece one full 128-byte line -- into registers while the previous
// chunk is being multiplied, then into LDS; (3) lane (row, rhs) walks its row's entries in CSR
// order (fixed summation order, the same as csr_spmv_tiles) reading values, indices and X from
// LDS; (4) the R x 16 results leave as 128-byte pieces.  fp32 values (config 5) are widened when
// staged; accumulation is fp64 either way.
constexpr int SPMM_KC = 16;               // right-hand sides per chunk (one 128-byte line per row of X)
constexpr int SPMM_XLD = 18;              // LDS row stride of the staged X rows (16-byte aligned, spreads banks)
constexpr int SPMM_NG = 10;               // most gather loads per thread and chunk: up to 320 distinct columns per tile

inline size_t spmm_tile_lds_bytes(int R, int ucap, int ecap) {
    return (size_t)ucap * SPMM_XLD * 8 + (size_t)ecap * 8 + (size_t)ecap * 2 + (size_t)ucap * 4 + (size_t)(R + 1) * 4 + 16;
}

constexpr int SPMM_THREADS = 256;

// LDS traffic decides this kernel (rocprofv3: HBM bytes = the algorithmic ones, the re-gathered rows of X
// come from L2): a lane owns FOUR right-hand sides of one row, so the index and value of an entry are
// read once per 4 products and the X values as 16-byte pieces -- 12 LDS cycles per 256 products instead
// of 6 per 64 with one right-hand side per lane.
template <typename VT, int NG>
__global__ __launch_bounds__(SPMM_THREADS, 3) void csr_spmm_tiles(const int64_t* __restrict__ rowptr,
                                                               const uint16_t* __restrict__ lidx,
                                                               const VT* __restrict__ vals,
                                                               const int64_t* __restrict__ tile_uptr,
                                                               const int32_t* __restrict__ ucols, int64_t n_rows,
                                                               const double* __restrict__ X, int64_t ldx,
                                                               double* __restrict__ Y, int64_t ldy, int k,
                                                               int R, int ucap, int ecap) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xs = smem;                                       // [ucap][SPMM_XLD]
    double* vs = xs + (size_t)ucap * SPMM_XLD;               // [ecap]
    uint16_t* ls = reinterpret_cast<uint16_t*>(vs + ecap);   // [ecap]
    int* uc = reinterpret_cast<int*>(ls + ecap + (ecap & 1));  // [ucap]
    int* rp = uc + ucap;                                     // [R + 1]
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)min((int64_t)R, n_rows - r0);
    const int64_t e0 = rowptr[r0];
    if (t <= R) rp[t] = (int)(rowptr[r0 + min(t, nr)] - e0);
    const int64_t u0 = tile_uptr[blockIdx.x];
    const int U = (int)(tile_uptr[blockIdx.x + 1] - u0);
    for (int u = t; u < U; u += SPMM_THREADS) uc[u] = ucols[u0 + u];
    const int cnt = (int)(rowptr[r0 + nr] - e0);
    for (int base = 0; base < cnt; base += 4 * SPMM_THREADS) {     // four independent loads per thread and pass
        uint16_t li[4];
        double v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = base + t + SPMM_THREADS * i;
            const bool ok = e < cnt;
            li[i] = ok ? lidx[e0 + e] : (uint16_t)0;
            v[i] = ok ? (double)vals[e0 + e] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = base + t + SPMM_THREADS * i;
            if (e < cnt) { ls[e] = li[i]; vs[e] = v[i]; }
        }
    }
    __syncthreads();
    const int seg = t & 7, ub = t >> 3;                      // gather: 8 threads x 16 B per row piece, 32 rows per pass
    const int kq = t & 3, rsub = t >> 2;                     // multiply: 4 lanes x 4 right-hand sides per row, 64 rows per pass
    // this thread's pieces of the distinct rows of X: row offsets once (units of 16 bytes, 32 bits: the host checks
    // n_cols * ldx < 2^32 for this path), the loads per chunk
    uint32_t goff[NG];
    const v2d* __restrict__ X2 = reinterpret_cast<const v2d*>(X);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int u = ub + 32 * i;
        goff[i] = (u < U) ? (uint32_t)(((int64_t)uc[u] * ldx) / 2 + seg) : 0xffffffffu;
    }
    v2d g[NG];
    auto gather = [&](int kc0) {
        const bool seg_ok = kc0 + 2 * seg < k;               // k is even on this path (host checks)
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            g[i] = (v2d){0.0, 0.0};
            if (goff[i] != 0xffffffffu && seg_ok) g[i] = X2[(size_t)goff[i] + (size_t)(kc0 / 2)];
        }
    };
    gather(0);
    for (int kc0 = 0; kc0 < k; kc0 += SPMM_KC) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int u = ub + 32 * i;
            if (u < U) *reinterpret_cast<v2d*>(xs + u * SPMM_XLD + 2 * seg) = g[i];
        }
        __syncthreads();
        if (kc0 + SPMM_KC < k) gather(kc0 + SPMM_KC);        // in flight while this chunk is multiplied
        for (int p = 0; p < R; p += 64) {
            const int row = p + rsub;
            int e = rp[min(row, R)];
            const int end = rp[min(row + 1, R)];
            v2d a01 = (v2d){0.0, 0.0}, a23 = (v2d){0.0, 0.0};
            const double* xk = xs + 4 * kq;
#define GMRF_SPMM_FMA(V, XA, XB)                                                                                  \
    a01.x = fma(V, XA.x, a01.x); a01.y = fma(V, XA.y, a01.y); a23.x = fma(V, XB.x, a23.x); a23.y = fma(V, XB.y, a23.y);
            // four entries per turn: 4 index + 4 value + 8 sixteen-byte X reads in flight before the first product (the
            // index -> X read chain is ~130 cycles of LDS latency; with three workgroups per CU nothing else hides it)
            // 8 / 4 / 2 / 1 entries per turn: all of a turn's index -> X read chains (~130 cycles of LDS latency each) are in
            // flight before its first product; with three workgroups per CU nothing else hides that latency
            // (2 per turn: 833 us, 4: 729 us, 8: 697 us on the burgers4096x512 matrix at k = 64; a predicated
            // 8-entry turn for the tail: 740 us)
            for (; e + 8 <= end; e += 8) {
                int ii[8];
                double vv[8];
                v2d xa[8], xb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { ii[u] = ls[e + u]; vv[u] = vs[e + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    xa[u] = *reinterpret_cast<const v2d*>(xk + ii[u] * SPMM_XLD);
                    xb[u] = *reinterpret_cast<const v2d*>(xk + ii[u] * SPMM_XLD + 2);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { GMRF_SPMM_FMA(vv[u], xa[u], xb[u]) }
            }
            for (; e + 4 <= end; e += 4) {
                const int i0 = ls[e], i1 = ls[e + 1], i2 = ls[e + 2], i3 = ls[e + 3];
                const double v0 = vs[e], v1 = vs[e + 1], v2 = vs[e + 2], v3 = vs[e + 3];
                const v2d x0a = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD), x0b = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD + 2);
                const v2d x1a = *reinterpret_cast<const v2d*>(xk + i1 * SPMM_XLD), x1b = *reinterpret_cast<const v2d*>(xk + i1 * SPMM_XLD + 2);
                const v2d x2a = *reinterpret_cast<const v2d*>(xk + i2 * SPMM_XLD), x2b = *reinterpret_cast<const v2d*>(xk + i2 * SPMM_XLD + 2);
                const v2d x3a = *reinterpret_cast<const v2d*>(xk + i3 * SPMM_XLD), x3b = *reinterpret_cast<const v2d*>(xk + i3 * SPMM_XLD + 2);
                GMRF_SPMM_FMA(v0, x0a, x0b) GMRF_SPMM_FMA(v1, x1a, x1b) GMRF_SPMM_FMA(v2, x2a, x2b) GMRF_SPMM_FMA(v3, x3a, x3b)
            }
            for (; e + 2 <= end; e += 2) {
                const int i0 = ls[e], i1 = ls[e + 1];
                const double v0 = vs[e], v1 = vs[e + 1];
                const v2d x0a = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD), x0b = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD + 2);
                const v2d x1a = *reinterpret_cast<const v2d*>(xk + i1 * SPMM_XLD), x1b = *reinterpret_cast<const v2d*>(xk + i1 * SPMM_XLD + 2);
                GMRF_SPMM_FMA(v0, x0a, x0b) GMRF_SPMM_FMA(v1, x1a, x1b)
            }
            if (e < end) {
                const int i0 = ls[e];
                const double v0 = vs[e];
                const v2d x0a = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD), x0b = *reinterpret_cast<const v2d*>(xk + i0 * SPMM_XLD + 2);
                GMRF_SPMM_FMA(v0, x0a, x0b)
            }
#undef GMRF_SPMM_FMA
            if (row < nr) {
                double* yp = Y + (r0 + row) * ldy + kc0 + 4 * kq;
                if (kc0 + 4 * kq + 4 <= k && (ldy & 1) == 0) {
                    *reinterpret_cast<v2d*>(yp) = a01; *reinterpret_cast<v2d*>(yp + 2) = a23;
                } else {
                    if (kc0 + 4 * kq < k) yp[0] = a01.x;
                    if (kc0 + 4 * kq + 1 < k) yp[1] = a01.y;
                    if (kc0 + 4 * kq + 2 < k) yp[2] = a23.x;
                    if (kc0 + 4 * kq + 3 < k) yp[3] = a23.y;
                }
            }
        }
        __syncthreads();
    }
}

// The same kernel with the entries of a row PADDED to a multiple of 8 in LDS (value 0, the row's first local
// index): every row starts at a multiple of 8 entries, so the 8 indices of a turn are one ALIGNED 16-byte
// read and the values four -- the unpadded kernel's merged index read sits at a 2-byte boundary
// (SQ_LDS_UNALIGNED_STALL = a third of its LDS-active cycles, tools/pmc_spmm_lds.sh) -- and the tail
// turns (4 / 2 / 1 entries) disappear.  A product with a padding entry adds 0 * x = 0 (exact for finite x).
// Entries are staged row by row (4 lanes per row), the padded row starts come from a wave scan.
// The staged rows of X are 128 bytes with NO padding, swizzled instead: a row holds eight 16-byte slots, lane
// (row, kq) of the multiply reads slot kq (right-hand sides 4kq, 4kq+1) and slot 4 + kq (4kq+2, 4kq+3), and slot s of
// local row u lives at position s ^ (4 * bit1(u)).  The 16 lanes a ds_read_b128 serves together are 4 rows x 4 kq:
// rows with consecutive local indices (the common case: neighbouring rows of a stencil matrix) then cover all 64
// banks exactly once -- parity of u picks the 128-byte half of the bank array, bit 1 the 64-byte quarter.  With the
// padded stride of 18 doubles the unpadded kernel's reads of rows u and u + 2 overlap in 3 of 4 slots
// (SQ_LDS_BANK_CONFLICT = 21 % of its LDS-active cycles); stride 20: 767 us, 18: 621 us, swizzle: 585 us.
template <typename VT, int NG>
__global__ __launch_bounds__(SPMM_THREADS, 3) void csr_spmm_tiles_pad(const int64_t* __restrict__ rowptr,
                                                                   const uint16_t* __restrict__ lidx,
                                                                   const VT* __restrict__ vals,
                                                                   const int64_t* __restrict__ tile_uptr,
                                                                   const int32_t* __restrict__ ucols, int64_t n_rows,
                                                                   const double* __restrict__ X, int64_t ldx,
                                                                   double* __restrict__ Y, int64_t ldy, int k,
                                                                   int R, int ucap, int ecap) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xs = smem;                                       // [ucap][16], swizzled (below)
    double* vs = xs + (size_t)ucap * SPMM_KC;               // [ecap + 2 R]  (padded rows; row r skewed by 2 r doubles: rows 16 entries
                                                             //               apart would put rows r and r + 2 on the same banks)
    uint16_t* ls = reinterpret_cast<uint16_t*>(vs + ecap + 2 * R);   // [ecap]
    int* uc = reinterpret_cast<int*>(ls + ecap);             // [ucap]   (ecap is a multiple of 8)
    int* rp = uc + ucap;                                     // [R + 1]  CSR offsets relative to the tile
    int* pp = rp + R + 1;                                    // [R + 1]  padded offsets
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)min((int64_t)R, n_rows - r0);
    const int64_t e0 = rowptr[r0];
    if (t < 64) {                                            // R <= 64: one wave scans the padded row lengths
        const int a = (t < R) ? (int)(rowptr[r0 + min(t, nr)] - e0) : 0;
        const int b = (t < R) ? (int)(rowptr[r0 + min(t + 1, nr)] - e0) : 0;
        int x = (b - a + 7) & ~7;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (t >= d) x += y;
        }
        if (t < R) { rp[t] = a; pp[t + 1] = x; if (t == R - 1) rp[R] = b; }
        if (t == 0) pp[0] = 0;
    }
    const int64_t u0 = tile_uptr[blockIdx.x];
    const int U = (int)(tile_uptr[blockIdx.x + 1] - u0);
    for (int u = t; u < U; u += SPMM_THREADS) uc[u] = ucols[u0 + u];
    __syncthreads();
    const int kq = t & 3, rsub = t >> 2;                     // 4 lanes per row: staging of entries and multiply
    for (int p = 0; p < R; p += 64) {
        const int row = p + rsub;
        if (row < R) {
            const int a = rp[row], len = rp[row + 1] - a, pa = pp[row], plen = pp[row + 1] - pa;
            uint16_t li[4];
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                    // the first 16 entries of the row: independent loads
                const int j = kq + 4 * i;
                const bool ok = j < len;
                li[i] = ok ? lidx[e0 + a + j] : (uint16_t)0xffff;
                v[i] = ok ? (double)vals[e0 + a + j] : 0.0;
            }
            const uint16_t first = (len > 0) ? lidx[e0 + a] : (uint16_t)0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = kq + 4 * i;
                if (j < plen) { ls[pa + j] = (j < len) ? li[i] : first; vs[pa + j + 2 * row] = v[i]; }
            }
            for (int j = 16 + kq; j < plen; j += 4) {
                const bool ok = j < len;
                ls[pa + j] = ok ? lidx[e0 + a + j] : first;
                vs[pa + j + 2 * row] = ok ? (double)vals[e0 + a + j] : 0.0;
            }
        }
    }
    const int seg = t & 7, ub = t >> 3;                      // gather: 8 threads x 16 B per row piece, 32 rows per pass
    uint32_t goff[NG];
    const v2d* __restrict__ X2 = reinterpret_cast<const v2d*>(X);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int u = ub + 32 * i;
        goff[i] = (u < U) ? (uint32_t)(((int64_t)uc[u] * ldx) / 2 + seg) : 0xffffffffu;
    }
    v2d g[NG];
    auto gather = [&](int kc0) {
        const bool seg_ok = kc0 + 2 * seg < k;               // k is even on this path (host checks)
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            g[i] = (v2d){0.0, 0.0};
            if (goff[i] != 0xffffffffu && seg_ok) g[i] = X2[(size_t)goff[i] + (size_t)(kc0 / 2)];
        }
    };
    gather(0);
    for (int kc0 = 0; kc0 < k; kc0 += SPMM_KC) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int u = ub + 32 * i;
            if (u < U) {
                const int slot = (((seg & 1) << 2) | (seg >> 1)) ^ ((u & 2) << 1);
                *reinterpret_cast<v2d*>(xs + u * SPMM_KC + 2 * slot) = g[i];
            }
        }
        __syncthreads();                                     // (first chunk: also the staged entries)
        if (kc0 + SPMM_KC < k) gather(kc0 + SPMM_KC);        // in flight while this chunk is multiplied
        for (int p = 0; p < R; p += 64) {
            const int row = p + rsub;
            const int e_begin = pp[min(row, R)], e_end = pp[min(row + 1, R)];
            v2d a01 = (v2d){0.0, 0.0}, a23 = (v2d){0.0, 0.0};
            for (int e = e_begin; e < e_end; e += 8) {
                const uint4 q = *reinterpret_cast<const uint4*>(ls + e);
                const double* ve = vs + e + 2 * min(row, R);
                const v2d v01 = *reinterpret_cast<const v2d*>(ve), v23 = *reinterpret_cast<const v2d*>(ve + 2);
                const v2d v45 = *reinterpret_cast<const v2d*>(ve + 4), v67 = *reinterpret_cast<const v2d*>(ve + 6);
                const int ii[8] = {(int)(q.x & 0xffffu), (int)(q.x >> 16), (int)(q.y & 0xffffu), (int)(q.y >> 16),
                                   (int)(q.z & 0xffffu), (int)(q.z >> 16), (int)(q.w & 0xffffu), (int)(q.w >> 16)};
                const double vv[8] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y, v67.x, v67.y};
                v2d xa[8], xb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int o = ((ii[u] << 4) | (kq << 1)) ^ ((ii[u] & 2) << 2);     // in doubles: row * 16 + 2 * slot(kq, 0)
                    xa[u] = *reinterpret_cast<const v2d*>(xs + o);
                    xb[u] = *reinterpret_cast<const v2d*>(xs + (o ^ 8));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a01.x = fma(vv[u], xa[u].x, a01.x); a01.y = fma(vv[u], xa[u].y, a01.y);
                    a23.x = fma(vv[u], xb[u].x, a23.x); a23.y = fma(vv[u], xb[u].y, a23.y);
                }
            }
            if (row < nr) {
                double* yp = Y + (r0 + row) * ldy + kc0 + 4 * kq;
                if (kc0 + 4 * kq + 4 <= k && (ldy & 1) == 0) {
                    *reinterpret_cast<v2d*>(yp) = a01; *reinterpret_cast<v2d*>(yp + 2) = a23;
                } else {
                    if (kc0 + 4 * kq < k) yp[0] = a01.x;
                    if (kc0 + 4 * kq + 1 < k) yp[1] = a01.y;
                    if (kc0 + 4 * kq + 2 < k) yp[2] = a23.x;
                    if (kc0 + 4 * kq + 3 < k) yp[3] = a23.y;
                }
            }
        }
        __syncthreads();
    }
}

inline size_t spmm_tile_pad_lds_bytes(int R, int ucap, int ecap_pad) {
    return (size_t)ucap * SPMM_KC * 8 + (size_t)(ecap_pad + 2 * R) * 8 + (size_t)ecap_pad * 2 + (size_t)ucap * 4 + (size_t)(2 * R + 2) * 4 + 16;
}

// Node-major right-hand sides without a tile plan (a tile with too many entries or distinct columns,
// odd k / ldx): 16 lanes along the right-hand sides, 4 rows per wave, operands straight from global
// memory; same summation order.
template <typename VT>
__global__ __launch_bounds__(256) void csr_spmm_rows(const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colidx,
                                                     const VT* __restrict__ vals, int64_t n_rows,
                                                     const double* __restrict__ X, int64_t ldx,
                                                     double* __restrict__ Y, int64_t ldy, int k) {
    const int t = threadIdx.x;
    const int kl = t & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (t >> 4);
    if (row >= n_rows) return;
    const int64_t a = rowptr[row], b = rowptr[row + 1];
    for (int kc0 = 0; kc0 < k; kc0 += 16) {
        const bool ok = kc0 + kl < k;
        double acc = 0.0;
        for (int64_t e = a; e < b; ++e) {
            const double xv = ok ? X[(int64_t)colidx[e] * ldx + kc0 + kl] : 0.0;
            acc = fma((double)vals[e], xv, acc);
        }
        if (ok) Y[row * ldy + kc0 + kl] = acc;
    }
}

// panel P[rhs][n_pad] (each right-hand side contiguous, padded blocks) -> node-major X[n][k] through a
// 64 x 64 LDS tile (both sides move 128-byte pieces); blockIdx.x = 64-dof tile, blockIdx.y = 64-rhs tile
__global__ __launch_bounds__(256) void unpack_panel_rows(const double* __restrict__ P, int64_t n_pad,
                                                         double* __restrict__ dst, int64_t ld, int bs, int bsp,
                                                         int64_t n, int k) {
    __shared__ double tile[64][65];
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    const int r0 = blockIdx.y * 64;
    for (int i = t; i < 4096; i += 256) {
        const int r = i >> 6, jj = i & 63;
        const int64_t j = j0 + jj;
        double v = 0.0;
        if (r0 + r < k && j < n) v = P[(int64_t)(r0 + r) * n_pad + (j / bs) * bsp + (j % bs)];
        tile[r][jj] = v;
    }
    __syncthreads();
    for (int i = t; i < 4096; i += 256) {
        const int jj = i >> 6, r = i & 63;
        if (r0 + r < k && j0 + jj < n) dst[(j0 + jj) * ld + r0 + r] = tile[r][jj];
    }
}

// acc[i] += sum_s ((QX[i][s] - d_i X[i][s]) / d_i)^2 with node-major QX, X (row stride ld): 16 lanes per node
__global__ __launch_bounds__(256) void rbmc_accumulate_rows(const double* __restrict__ QX, const double* __restrict__ X,
                                                            int64_t ld, const double* __restrict__ diag, int64_t n, int k,
                                                            double* __restrict__ acc, int plain_mc) {
    const int kl = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    double s = 0.0;
    if (i < n) {
        const double d = plain_mc ? 1.0 : diag[i];
        for (int r = kl; r < k; r += 16) {
            const double o = plain_mc ? X[i * ld + r] : (QX[i * ld + r] - d * X[i * ld + r]) / d;
            s = fma(o, o, s);
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
    if (i < n && kl == 0) acc[i] += s;
}

// ------------------------------------------------------------------------------- variances
// acc[i] += sum_s ((QX[s][i] - d_i X[s][i]) / d_i)^2     (RBMC off-diagonal term)
__global__ void rbmc_accumulate(const double* __restrict__ QX, const double* __restrict__ X,
                                int64_t ld, const double* __restrict__ diag, int64_t n, int k,
                                double* __restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = diag[i];
    double s = 0.0;
    for (int r = 0; r < k; ++r) {
        const double o = (QX[(int64_t)r * ld + i] - d * X[(int64_t)r * ld + i]) / d;
        s = fma(o, o, s);
    }
    acc[i] += s;
}

// acc[i] += sum_s X[s][i]^2
__global__ void mc_accumulate(const double* __restrict__ X, int64_t ld, int64_t n, int k,
                              double* __restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int r = 0; r < k; ++r) {
        const double v = X[(int64_t)r * ld + i];
        s = fma(v, v, s);
    }
    acc[i] += s;
}

__global__ void csr_extract_diag(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                 const double* __restrict__ vals, const float* __restrict__ vals32,
                                 int64_t n, double* __restrict__ diag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p)
        if (colidx[p] == i) d += vals ? vals[p] : (double)vals32[p];
    diag[i] = d;
}

// var = base + acc * scale     (base: 1/Q_ii for RBMC, 0 for MC)
__global__ void var_finish(const double* __restrict__ acc, const double* __restrict__ diag,
                           double scale, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (diag ? 1.0 / diag[i] : 0.0) + acc[i] * scale;
}

// sum over the diagonal of log(L[j][j]) for every block; one workgroup per block (blockIdx.x) and
// problem (blockIdx.y), fixed-order tree so the result is reproducible.
__global__ __launch_bounds__(256) void logdet_blocks(const double* __restrict__ L, int64_t blk_stride,
                                                     int64_t ld, int bs, double* __restrict__ out,
                                                     int64_t pL, int64_t pout) {
    const double* Lb = L + (int64_t)blockIdx.y * pL + (int64_t)blockIdx.x * blk_stride;
    double s = 0.0;
    for (int j = threadIdx.x; j < bs; j += 256) s += log(Lb[(int64_t)j * ld + j]);
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)blockIdx.y * pout + blockIdx.x] = red[0];
}

// diag(S) of a dense block into out (selected inversion output)
__global__ void extract_diag_dense(const double* __restrict__ S, int64_t ld, int bs,
                                   double* __restrict__ out, int64_t pS, int64_t pout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bs) out[(int64_t)blockIdx.y * pout + i] = S[(int64_t)blockIdx.y * pS + (int64_t)i * ld + i];
}

// S = 0.5 (S + S^T) is not needed: selected inversion keeps symmetry to rounding.

// ---- selected inversion, round 4 (gmrf_hip.hip: var_exact) ----------------------------------------------------------
// 64 x 64 tiles of a row-major matrix transposed through LDS: dst[c][r] = src[r][c].
//   mode 0: every tile of a (64 tr) x (64 tc) matrix            -> grid.x = tr * tc
//   mode 1: the tiles r >= c, c < tc, of a lower-triangular (64 tr)^2 matrix (Linv_i -> its transpose, rows 0 .. 64 tc)
//           -> grid.x = sum_{c < tc} (tr - c)
//   mode 2: in place, the lower triangle of a (64 tr)^2 matrix into the upper one (mirror of a product that was formed
//           on its lower tiles only) -> grid.x = tr (tr + 1) / 2.  Diagonal tiles are mirrored element-wise: the 64 x 64
//           kernels write them whole, the 32 x 32-tile kernel of small launches (gemm_f64_ll) leaves their upper-right
//           quarter unwritten.
// blockIdx.y = problem.
__global__ __launch_bounds__(256) void transpose_tiles(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst,
                                                       int64_t ldd, int64_t pS, int64_t pD, int tr, int tc, int mode) {
    __shared__ double t[64][65];
    int r, c;
    const int w = (int)blockIdx.x;
    if (mode == 0) { r = w / tc; c = w % tc; }
    else if (mode == 1) { int q = w; c = 0; while (q >= tr - c) { q -= tr - c; ++c; } r = c + q; }
    else { int q = w; r = 0; while (q > r) { q -= r + 1; ++r; } c = q; }
    const double* s = src + (int64_t)blockIdx.y * pS + (int64_t)r * 64 * lds_ + (int64_t)c * 64;
    double* d = dst + (int64_t)blockIdx.y * pD + (int64_t)c * 64 * ldd + (int64_t)r * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[ty + 4 * i][tx] = s[(int64_t)(ty + 4 * i) * lds_ + tx];
    __syncthreads();
    const bool diag = mode == 2 && r == c;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (!diag || ty + 4 * i < tx) d[(int64_t)(ty + 4 * i) * ldd + tx] = t[tx][ty + 4 * i];
}

// out[n] = sum_{k >= n} X[k][n] * (k < cm || !Y ? X[k][n] : Y[k][n])     diag(Linv^T Y) with Y = Linv above row cm:
// a workgroup owns 64 columns (a lane a column: rows are read as whole 512-byte lines), its four waves split the
// rows from the column block's diagonal down, fixed summation order (rows ascending per wave, waves 0 .. 3).
__global__ __launch_bounds__(256) void coldot_lower(const double* __restrict__ X, const double* __restrict__ Y, int64_t ld, int bsp,
                                                    int cm, int bs, double* __restrict__ out, int64_t pX, int64_t pY, int64_t pout, int u0) {
    __shared__ double part[4][64];
    const int u = u0 + (int)blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;      // column tiles u0 .. only
    const int n = 64 * u + lane;
    const double* x = X + (int64_t)blockIdx.y * pX;
    const double* y = Y ? Y + (int64_t)blockIdx.y * pY : nullptr;
    const int k0 = 64 * u, span = bsp - k0, per = (span / 4 + 63) / 64 * 64;
    const int kb = k0 + wv * per, ke = min(bsp, kb + per);
    double acc = 0.0;
    for (int k = kb; k < ke; ++k) {
        const double xv = x[(int64_t)k * ld + n];
        const double yv = (y && k >= cm) ? y[(int64_t)k * ld + n] : xv;
        acc = fma(xv, yv, acc);
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && n < bs) out[(int64_t)blockIdx.y * pout + n] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// dense identity add: M[i][i] += 1
__global__ void add_identity(double* __restrict__ M, int64_t ld, int bs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bs) M[(int64_t)i * ld + i] += 1.0;
}

// micro-benchmarks -------------------------------------------------------------------
typedef double mb_v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(double* out, int iters) {
    mb_v4d a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    const mb_v4d s = a0 + a1 + a2 + a3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];
}

__global__ __launch_bounds__(256) void hbm_read_kernel(const double2* __restrict__ src, int64_t n16,
                                                       double* out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}

}  // namespace gmrf
