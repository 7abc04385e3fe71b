// One problem (or a few): the dense Cholesky of a bs x bs Schur block (dpotrf of D_i - C C^T,
// /root/reference/src/tridiagonal_cholesky.jl:67,77) as ONE persistent launch per block (blocks of up to 16 tiles) or per
// 256-column panel (larger blocks) instead of one launch per 64-column step.  Round 4.
//
// The arithmetic is the look-ahead chain of potrf_step.hpp (`lookahead`), operation for operation -- the factor and the
// inverse come out bitwise equal to the launch-per-step form -- but the launch boundary between two steps is gone:
//
//   workgroup 0 ("chain")   for j = j0 .. j1-1:  tile (j, j) -> L_jj, X_jj = L_jj^-1 (tile_potrf_inv), published; then
//                           L[j+1, j] = S'[j+1, j] X_jj^T, tile (j+1, j+1) -= L[j+1, j] L[j+1, j]^T in LDS, and on to j + 1.
//                           X_jj stays in LDS between the two (the launch-per-step form stored and re-loaded it).
//   workgroup of tile (r,c) keeps its tile of the trailing block in REGISTERS (MFMA C/D layout) over all its steps
//                           j = j0 .. c-1:  Lr = S'[r, j] X_jj^T,  Lc = S'[c, j] X_jj^T,  tile -= Lr Lc^T;  stores the tile once,
//                           when it is final (it is then the panel tile of step c for everybody else), and -- blocks of up
//                           to 16 tiles -- goes on to assemble tile (r, c) of the block inverse,
//                           X[r, c] = -X_rr sum_{p = c}^{r-1} L[r, p] X[p, c]   (the sums in the order of xrow_strip).
//
// Hand-offs between workgroups follow the guide's write-through form (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement
// & inter-workgroup visibility", first row of the table of measured hand-offs): every handed-off byte is stored `sc1` (write
// through), every storing wave drains `vmcnt(0)`, the workgroup meets at a barrier, ONE lane stores the flag (agent scope,
// relaxed); the consumer's lane 0 polls that word relaxed, the workgroup meets at a barrier, and EVERY load of handed-off bytes
// is an `sc1` buffer load (L1 bypassed; per-XCD L2s are not coherent).  Nothing depends on dispatch order or placement.
// Every workgroup of a problem must be resident at once (one per CU: 140 KB of LDS): the host launches this form only when
// 1 + tiles <= the CU count, every spin is bounded (`spin_limit` ticks of the 100 MHz clock), and a spin that gives up sets
// the abort word, on which every other wait returns too: the kernel always drains, the host sees the word after the
// factorisation and repeats it with the launch-per-step form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "potrf_step.hpp"

namespace gmrf {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// words in LDS that waves of one workgroup pass to each other: address space 3 (a generic `volatile int*` became FLAT loads /
// stores with sc0 sc1, which wait for every outstanding global store of the wave: 1 200 cycles per look, round 4)
typedef __attribute__((address_space(3))) volatile int lds_word;

struct PersistArgs {
    double* S; double* L; double* X;
    int64_t ld;
    int nt;                 // tiles per dimension of the block
    int j0, j1;             // column tiles [j0, j1) of this launch (blocks of up to 16 tiles: 0, nt)
    int xrows;              // 1, 2: the rows of the block inverse are assembled here too (j0 = 0, j1 = nt); 2: column j0's tiles by the diagonal tiles' workgroups
    int tail_panel;         // 1 (larger blocks): the workgroups of the last column also form L[r, j1-1] for the rows below the panel
    int* info; int blk;
    int64_t pS, pL, pX; int blk_per_problem;
    unsigned* flags;        // [problems][flag_stride]; zero at the start of a launch (left so by the last workgroup out of the previous one)
    int flag_stride;
    unsigned* abort_word;   // zeroed at the start of a factorisation, read by the host at its end
    unsigned spin_limit;
    unsigned long long* stamps;   // diagnostic (tests / tuning): s_memtime of the chain workgroup per step, else nullptr
};

// flag words of one problem: D[j] (L_jj, X_jj stored) | F[r][c] (tile (r,c) of S final) | PL[r][c] (L[r,c] stored) |
// XF[r][c] (X[r,c] stored)
// followed by ONE more word: the count of workgroups of this problem that have left the launch.  The last one out zeroes every
// word again (nobody is left to read them), so the words are zero when the next launch starts: the memset node that used to
// precede every launch (4.5 - 4.9 us each, 8 192 per C5 factorisation = 40 of its 1 604 ms) is gone.  Zeroed once, by the
// host, when they are allocated.
__host__ __device__ inline int persist_flag_count(int nt) { return nt + 3 * nt * nt; }
__host__ __device__ inline int persist_flag_words(int nt) { return (persist_flag_count(nt) + 1 + 3) & ~3; }

// tiles owned by worker workgroups, column by column: column j0 (inverse only, when xrows), then c = j0+1 .. j1-1 with the rows
// r = c+1 .. nt-1 and, from c = j0+2 on, the diagonal tile (c, c) (steps j0 .. c-2; step c-1 is the chain's).
// xrows == 2 (the 4 x 4 panel blocks of small batches): column j0 gets ONE workgroup, for the last row -- the inverse tiles
// (c-1, j0) of the rows above are assembled by the workgroups of the diagonal tiles (c, c) once those are final (they have no
// inverse tile of their own): 7 workgroups per problem instead of 9, so that four handles x batch 8 ask for 224 of the 256 CUs
// and not for 288 (elliptic512 4 x 8: 6.54 k -> 6.74 k solves/s).  Not for the blocks of one problem: there the sums of column
// j0 are the long ones (c - 1 terms), and a workgroup that starts them only after its diagonal tile's last step ends the
// launch late (darcy256: factor 16.1 -> 17.5 ms, measured).
__host__ __device__ inline int persist_tiles(int nt, int j0, int j1, int xrows) {
    int n = 0;
    if (xrows == 2) n = (nt - 1 > j0) ? 1 : 0;
    else if (xrows) n = nt - 1 - j0;
    for (int c = j0 + 1; c < j1; ++c) n += nt - 1 - c + ((c >= j0 + 2) ? 1 : 0);
    return n;
}
__device__ __forceinline__ void persist_tile_of(int t, int nt, int j0, int j1, int xrows, int& r, int& c) {
    if (xrows == 2) {
        if (nt - 1 > j0) {
            if (t == 0) { r = nt - 1; c = j0; return; }
            t -= 1;
        }
    } else if (xrows) {
        const int cnt = nt - 1 - j0;
        if (t < cnt) { r = j0 + 1 + t; c = j0; return; }
        t -= cnt;
    }
    for (c = j0 + 1; c < j1; ++c) {
        const int first = (c >= j0 + 2) ? c : c + 1;
        const int cnt = nt - first;
        if (t < cnt) { r = first + t; return; }
        t -= cnt;
    }
    r = -1; c = -1;
}

__device__ __forceinline__ unsigned ld_flag(const unsigned* f) {
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(unsigned* f, unsigned v) {
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: poll `f` until it is set; false when the abort word is set or the time limit passes (then sets it)
__device__ __forceinline__ bool spin_until(const unsigned* f, unsigned* abort_w, unsigned limit) {
    if (ld_flag(f)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (ld_flag(f)) return true;
        if ((++n & 31u) == 0u) {
            if (ld_flag(abort_w)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)limit) { st_flag(abort_w, 1u); return false; }
        }
    }
}

// The workgroup waits for up to three flags (nullptr: none): lane 0 polls, everybody meets at the barrier.  `okw` are two LDS
// words used alternately (`phase` counts the waits of this workgroup) so that a word is never rewritten while a slower wave
// still reads the previous verdict.
__device__ __forceinline__ bool wg_wait(const unsigned* a, const unsigned* b, const unsigned* c, const PersistArgs& pa,
                                        lds_word* okw, int& phase, int tid) {
    lds_word* w = okw + (phase & 1);
    ++phase;
    if (tid < 64) {
        // lanes 0, 1, 2 of wave 0 poll one flag each, side by side (three round trips to the fabric in the time of one)
        const unsigned* f = tid == 0 ? a : (tid == 1 ? b : (tid == 2 ? c : nullptr));
        bool ok = true;
        if (f) ok = spin_until(f, pa.abort_word, pa.spin_limit);
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        if (tid == 0) *w = all_ok ? 1 : 0;
    }
    __syncthreads();
    return *w != 0;
}

// every wave has issued its sc1 stores: drain them, meet, ONE lane raises the flags
__device__ __forceinline__ void wg_publish(unsigned* f1, unsigned* f2, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (f1) st_flag(f1, 1u);
        if (f2) st_flag(f2, 1u);
    }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000);
}

// 64 x 64 tile, global (row stride ld) -> LDS (row stride TLD), every load sc1; 256 threads
__device__ __forceinline__ void tile_g2s_sc1(const double* g, int64_t ld, double* s, int tid) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g);
    v4u v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 5, c = (idx & 31) * 2;
        v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((r * ld + c) * 8), 0, 16);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 5, c = (idx & 31) * 2;
        *reinterpret_cast<v4u*>(s + r * TLD + c) = v[it];
    }
}
// LDS -> global, every store sc1 (write through)
__device__ __forceinline__ void tile_s2g_sc1(const double* s, double* g, int64_t ld, int tid) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 5, c = (idx & 31) * 2;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u*>(s + r * TLD + c), rs, (int)((r * ld + c) * 8), 0, 16);
    }
}
// this wave's 16-row strip of a tile in the MFMA C/D layout (row lq + 4 q of block row `wave`, column 16 Jb + li)
__device__ __forceinline__ void strip_load_sc1(const double* g, int64_t ld, v4d (&t)[4], int jb_end, int wave, int li, int lq) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (Jb < jb_end) {
                const v2u u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(((16 * wave + lq + 4 * q) * ld + 16 * Jb + li) * 8), 0, 16);
                t[Jb][q] = __hiloint2double((int)u.y, (int)u.x);
            } else {
                t[Jb][q] = 0.0;
            }
        }
}
__device__ __forceinline__ void strip_store_sc1(double* g, int64_t ld, const v4d (&t)[4], int jb_end, bool negate, int wave, int li, int lq) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g);
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (Jb < jb_end) {
                const double x = negate ? -t[Jb][q] : t[Jb][q];
                v2u u;
                u.x = (unsigned)__double2loint(x); u.y = (unsigned)__double2hiint(x);
                __builtin_amdgcn_raw_buffer_store_b64(u, rs, (int)(((16 * wave + lq + 4 * q) * ld + 16 * Jb + li) * 8), 0, 16);
            }
        }
}

constexpr size_t POTRF_PERSIST_LDS = POTRF_STEP_LDS + 64;      // + words in LDS: two verdicts of wg_wait, four of the prefetch, the publish count, the compute waves' barrier count

// A barrier of the four computing waves of the chain workgroup alone (s_barrier would wait for waves 4 - 7, which are busy
// publishing the last tile): every wave adds one to an LDS word behind its LDS work (a wave's LDS operations are served in order)
// and waits until the word shows that all four have.  `epoch` counts this wave's barriers.
__device__ __forceinline__ void compute_waves_barrier(lds_word* cnt, int& epoch, int lane) {
    asm volatile("" ::: "memory");
    epoch += 4;
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (*cnt < epoch) {}
    asm volatile("" ::: "memory");
}

// Diagonal tile update  T = N - L L^T  (lower 16 x 16 blocks) with the ten blocks dealt 3 / 3 / 2 / 2 over the four waves instead of
// strip by strip (1 / 2 / 3 / 4: wave 3's 64 MFMAs were the length of the step, 4 450 cycles; now 48).  Every block is summed as
// strip_nt sums it (k groups ascending, the two k of a group in order): bitwise the same tile.
template <int W>
__device__ __forceinline__ void diag_update_balanced(const double* Ls, const double* Ns, double* Ts, int li, int lq) {
    constexpr int BI[4][3] = {{0, 3, 3}, {1, 1, 3}, {2, 2, -1}, {2, 3, -1}};
    constexpr int BJ[4][3] = {{0, 0, 1}, {0, 1, 2}, {0, 1, -1}, {2, 3, -1}};
    v4d acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[b] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (BI[W][b] >= 0) {
                const v2d av = *reinterpret_cast<const v2d*>(Ls + (16 * BI[W][b] + li) * TLD + k);
                const v2d bv = *reinterpret_cast<const v2d*>(Ls + (16 * BJ[W][b] + li) * TLD + k);
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc[b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (BI[W][b] >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (16 * BI[W][b] + lq + 4 * q) * TLD + 16 * BJ[W][b] + li;
                Ts[off] = Ns[off] - acc[b][q];
            }
        }
    }
}
// one 16 x 16 block (BI, BJ) of  N - L L^T  in the MFMA C/D layout, summed as above (bitwise the block diag_update_balanced gives)
__device__ __forceinline__ v4d diag_update_block(int BI, int BJ, const double* Ls, const double* Ns, int li, int lq) {
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
        const int k = 8 * kg + 2 * lq;
        const v2d av = *reinterpret_cast<const v2d*>(Ls + (16 * BI + li) * TLD + k);
        const v2d bv = *reinterpret_cast<const v2d*>(Ls + (16 * BJ + li) * TLD + k);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc, 0, 0, 0);
    }
    return load_d16(Ns + (16 * BI) * TLD + 16 * BJ, TLD, li, lq) - acc;
}
__device__ __forceinline__ void diag_update_balanced_w(int w, const double* Ls, const double* Ns, double* Ts, int li, int lq) {
    switch (w) {
        case 0: diag_update_balanced<0>(Ls, Ns, Ts, li, lq); break;
        case 1: diag_update_balanced<1>(Ls, Ns, Ts, li, lq); break;
        case 2: diag_update_balanced<2>(Ls, Ns, Ts, li, lq); break;
        default: diag_update_balanced<3>(Ls, Ns, Ts, li, lq); break;
    }
}

// The chain workgroup has EIGHT waves (round 5): waves 0 - 3 compute (the tile factorisation keeps all four busy now: every one owns
// a panel), waves 4 - 7 move tiles beside them, between the two barriers of tile_potrf_inv (`extra`), each for itself:
//   waves 4, 5   store their half of L[j+1, j] (in As since the last product), then wait for the flag of the NEXT step's panel tile
//                S'[j+2, j+1] and bring their half of it into As;
//   waves 6, 7   wait for the flag of tile (j+2, j+2) and bring their half into Bs;
// each leaves `tag` in its LDS word when its half is there.  A wait ends when the flag is up or when the tile is done (X_33 is
// flagged: tile_potrf_inv's word TF_I + 3) -- the chain takes a prefetched operand only if both words for it carry the step's tag
// after the factorisation; what is missing it waits for and loads as before (1 450 + 2 400 cycles a step when it has to do both).
// (Round 4 did this on waves 1 and 2 in the shadow of wave 0's panels, with looks that never waited.)
struct ChainSide {
    const double* sL; double* gL;             // LDS tile -> global (nullptr: nothing to store)
    const unsigned* f1; const unsigned* f2;   // flags the prefetch needs (nullptr: none)
    const double* gA; double* sA;             // nullptr: no next step
    const double* gN; double* sN;
    int64_t ld;
    lds_word* done;                           // LDS: [0], [1] As halves;  [2], [3] Bs halves
    int tag;
    unsigned long long* stamps;
    unsigned long long* dbg;                  // diagnostic: 1 + the 64-column quarter of the tile in which As was there (tools/persist_stamps.py), else nullptr
    bool split_update;                        // the caller left only the first 16 columns of the updated tile in Ts (As = L[j+1, j], Bs = tile (j+1, j+1))
    lds_word* used;                           // LDS: counts the waves 1 - 3 that are done with As and Bs (3 per tile: `used_target`)
    int used_target;
    __device__ __forceinline__ void idle(int, int, int) {}
    // Waves 1 - 3, beside wave 0's first panel: the 16 x 16 blocks of  tile (j+1, j+1) - L[j+1, j] L[j+1, j]^T  of their own panels,
    // straight into the accumulator layout they wait in -- wave 1: (1,1), (2,1) and, from wave 3 through Ts, (3,1); wave 2: (2,2),
    // (3,2); wave 3: (3,1) for wave 1, then (3,3).  Only the first 16 columns of the update (one block per wave, before the
    // routine) are on the chain: 1 300 cycles where the whole update was 4 300.
    __device__ __forceinline__ void load_panel(int w, v4d (&s0)[4], double* Ts, tile_word* fl, int li, int lq, int& spins) {
        if (!split_update) {
#pragma unroll
            for (int I = 0; I < 4; ++I) s0[I] = load_d16(Ts + (16 * I) * TLD + 16 * w, TLD, li, lq);
            return;
        }
        const v4d z = (v4d){0.0, 0.0, 0.0, 0.0};
        s0[0] = z; s0[1] = z; s0[2] = z; s0[3] = z;
        if (w == 1) {
            s0[1] = diag_update_block(1, 1, sL, sN, li, lq);
            s0[2] = diag_update_block(2, 1, sL, sN, li, lq);
        } else if (w == 2) {
            s0[2] = diag_update_block(2, 2, sL, sN, li, lq);
            s0[3] = diag_update_block(3, 2, sL, sN, li, lq);
        } else {
            const v4d b31 = diag_update_block(3, 1, sL, sN, li, lq);
            store_d16(Ts + 48 * TLD + 16, TLD, b31, li, lq);
            tile_set(fl + TF_SIDE + 0, li + 16 * lq);
            s0[3] = diag_update_block(3, 3, sL, sN, li, lq);
        }
        asm volatile("" ::: "memory");
        if (li + 16 * lq == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)used, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (w == 1) {
            tile_wait(fl + TF_SIDE + 0, spins);
            s0[3] = load_d16(Ts + 48 * TLD + 16, TLD, li, lq);
        }
    }
    // waves 4 - 7, before they overwrite As / Bs with the next step's operands: waves 1 - 3 have read L and the tile out of them
    __device__ __forceinline__ bool operands_free(tile_word* fl) const {
        if (!split_update) return true;
        for (int n = 0; n < (1 << 16); ++n) {
            if (*used >= used_target) return true;
            if (fl[TF_I + 3] != 0) return false;
            __builtin_amdgcn_s_sleep(2);
        }
        return false;
    }
    __device__ __forceinline__ void half_load(const double* g, double* s, int half, int lane) const {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g);
        v4u v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = half * 1024 + lane + it * 64;
            const int r = idx >> 5, c = (idx & 31) * 2;
            v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((r * ld + c) * 8), 0, 16);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = half * 1024 + lane + it * 64;
            const int r = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<v4u*>(s + r * TLD + c) = v[it];
        }
    }
    // one wave: true when `f` is up; false when the tile factorisation is over first (or a bound of looks is reached)
    __device__ __forceinline__ bool wait_flag(const unsigned* f, tile_word* fl, int lane) const {
        if (!f) return true;
        for (int n = 0; n < (1 << 14); ++n) {
            unsigned v = 0u;
            if (lane == 0) v = ld_flag(f);
            if (__builtin_amdgcn_readfirstlane(v)) return true;
            if (fl[TF_I + 3] != 0) return false;
            __builtin_amdgcn_s_sleep(4);
        }
        return false;
    }
    __device__ __forceinline__ void extra(int w, int lane, tile_word* fl) {
        const int half = w & 1;
        if (w < 2) {
            if (gL) {
                const __amdgpu_buffer_rsrc_t rs = tile_rsrc(gL);
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int idx = half * 1024 + lane + it * 64;
                    const int r = idx >> 5, c = (idx & 31) * 2;
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u*>(sL + r * TLD + c), rs, (int)((r * ld + c) * 8), 0, 16);
                }
            }
            if (!gA || !wait_flag(f1, fl, lane) || !operands_free(fl)) return;
            half_load(gA, sA, half, lane);             // (behind this wave's own reads of its half of As: the LDS serves them in order)
            if (lane == 0) done[half] = tag;
            if (dbg && half == 0 && lane == 0) *dbg = 1ull + (fl[TF_CS + 0] != 0) + (fl[TF_CS + 1] != 0) + (fl[TF_CS + 2] != 0);
        } else {
            if (!gN || !wait_flag(f2, fl, lane) || !operands_free(fl)) return;
            half_load(gN, sN, half, lane);
            if (lane == 0) done[2 + half] = tag;
        }
    }
};

__device__ __forceinline__ void potrf_persist_body(PersistArgs pa) {
    pa.S += (int64_t)blockIdx.y * pa.pS;
    pa.L += (int64_t)blockIdx.y * pa.pL;
    pa.X += (int64_t)blockIdx.y * pa.pX;
    pa.blk += (int)blockIdx.y * pa.blk_per_problem;
    pa.flags += (int64_t)blockIdx.y * pa.flag_stride;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ts = smem;
    double* Xs = Ts + TILE_ELEMS;
    double* Wk = Xs + TILE_ELEMS;               // WK_ELEMS
    double* rinvs = Wk + WK_ELEMS;           // 64
    double* As = rinvs + 64;
    double* Bs = As + TILE_ELEMS;
    lds_word* okw = (lds_word*)reinterpret_cast<int*>(Bs + TILE_ELEMS);
    int phase = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches, not exec masks, on `wave`)
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ld = pa.ld;
    const int nt = pa.nt;
    unsigned* const fD = pa.flags;
    unsigned* const fF = fD + nt;
    unsigned* const fPL = fF + nt * nt;
    unsigned* const fXF = fPL + nt * nt;
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    bool bad = false;

    if (blockIdx.x == 0) {
        // ------------------------------------------------------------------ the chain (eight waves: see ChainSide)
        const bool cw = wave < 4;                            // the waves that compute; the others only join the barriers out here
        lds_word* done = okw + 2;
        if (tid < 4) done[tid] = 0;
        ChainSide cs;
        cs.ld = ld; cs.done = done; cs.stamps = nullptr; cs.dbg = nullptr; cs.sA = As; cs.sN = Bs; cs.sL = As;
        cs.split_update = false; cs.used = okw + 9; cs.used_target = 0;
        if (tid == 0) *cs.used = 0;
        auto prefetch_for = [&](int jn) {                    // operands of step jn (tile jn + 1) -> cs, or none past the last step
            cs.tag = jn + 1;
            if (jn + 1 < pa.j1) {
                const int64_t oj = (int64_t)jn * 64, o1 = oj + 64;
                cs.gA = pa.S + o1 * ld + oj; cs.gN = pa.S + o1 * ld + o1;
                cs.f1 = (jn > pa.j0) ? fF + (jn + 1) * nt + jn : nullptr;
                cs.f2 = (jn + 1 >= pa.j0 + 2) ? fF + (jn + 1) * nt + (jn + 1) : nullptr;
            } else {
                cs.gA = nullptr; cs.gN = nullptr; cs.f1 = nullptr; cs.f2 = nullptr;
            }
        };
        const int64_t o0 = (int64_t)pa.j0 * 64;
        if (cw) tile_g2s(pa.S + o0 * ld + o0, ld, Ts, tid);
        cs.gL = nullptr;
        prefetch_for(pa.j0);
        __syncthreads();
        // X_jj is what every workgroup of the next step waits for.  Waves 4 - 7 store it (write through), issue the stores of L_jj
        // (which nobody inside the launch reads) behind it, wait for all but those eight youngest stores -- a wave's stores are
        // counted in issue order: X_jj and, issued long before, this wave's half of L[j, j-1] have arrived -- count themselves in an
        // LDS word, and the one whose count is the fourth raises the flags (the guide's form: every storing wave waits for its
        // stores, the wave whose add is last signals).  Waves 0 - 3 are in the next step's first product meanwhile: the publish
        // (2 000 cycles of store issue and drain per step in round 4) and the 600 cycles of L_jj's store issue have left the chain.
        // (Measured on the way: the step's first barrier met with `vmcnt(0)` by every wave -- waves 0 - 3 waited there for the
        //  drain of L_jj, 2 460 cycles where 880 had been; L_jj held in 32 registers until Ts was free -- the kernel is at the 256
        //  registers an eight-wave workgroup can have, 38 of them spilled, and the diagonal update took 6 600 cycles instead of 4 200.)
        lds_word* pubcnt = okw + 6;
        lds_word* cwcnt = okw + 7;
        lds_word* tsread = okw + 8;                          // waves 4 - 7 have read L_jj out of Ts (its place is the next diagonal tile's)
        int cw_epoch = 0, ts_epoch = 0;
        if (tid == 0) { *pubcnt = 0; *cwcnt = 0; *tsread = 0; }
        auto publish_tile = [&](int jt, unsigned* also) {    // waves 4 - 7
            const int64_t ot = (int64_t)jt * 64;
            tile_s2g_sc1(Xs, pa.X + ot * ld + ot, ld, tid - 256);
            tile_s2g_sc1(Ts, pa.L + ot * ld + ot, ld, tid - 256);          // (eight stores per lane)
            asm volatile("" ::: "memory");                  // (the stores are issued: their data has left LDS)
            if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)tsread, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (lane == 0) {
                const int before = __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((before & 3) == 3) {
                    st_flag(fD + jt, 1u);
                    if (also) st_flag(also, 1u);
                }
            }
        };
        tile_potrf_inv(Ts, Xs, Wk, rinvs, tid, bad, cs);
        if (bad && tid == 0) atomicCAS(pa.info, 0, pa.blk);
        if (!cw) publish_tile(pa.j0, nullptr);
        // (The step's prefetch words were written before tile_potrf_inv's last barrier.)
        if (pa.stamps && tid == 0) pa.stamps[0] = __builtin_amdgcn_s_memtime();
        for (int j = pa.j0; j + 1 < pa.j1; ++j) {
            const int64_t oj = (int64_t)j * 64, o1 = oj + 64;
            // (tile_potrf_inv ended with a barrier: the words are what waves 4 - 7 left during the last factorisation)
            const bool pre_a = done[0] == j + 1 && done[1] == j + 1, pre_n = done[2] == j + 1 && done[3] == j + 1;
            if (pre_a && pre_n) {
                if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 1] = __builtin_amdgcn_s_memtime();
            } else {
                const unsigned* w1 = (j > pa.j0) ? fF + (j + 1) * nt + j : nullptr;
                const unsigned* w2 = (j + 1 >= pa.j0 + 2) ? fF + (j + 1) * nt + (j + 1) : nullptr;
                if (!wg_wait(w1, w2, nullptr, pa, okw, phase, tid)) return;
                if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 1] = __builtin_amdgcn_s_memtime();
                if (cw && !pre_a) tile_g2s_sc1(pa.S + o1 * ld + oj, ld, As, tid);
                if (cw && !pre_n) tile_g2s_sc1(pa.S + o1 * ld + o1, ld, Bs, tid);
                __syncthreads();
            }
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 2] = __builtin_amdgcn_s_memtime();
            // L[j+1, j] = As Xs^T   (X lower triangular: column block Jb needs the k groups 0 .. 2 Jb + 1)
            v4d lr[4];
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb) lr[Jb] = zero;
            if (cw) {
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    const int k = 8 * kg + 2 * lq;
                    const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
#pragma unroll
                    for (int Jb = kg / 2; Jb < 4; ++Jb) {
                        const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
                        lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, lr[Jb], 0, 0, 0);
                        lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, lr[Jb], 0, 0, 0);
                    }
                }
            }
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 3] = __builtin_amdgcn_s_memtime();
            // (waves 4 - 7 are publishing the last tile: the next two meetings are of the four computing waves alone)
            if (cw) {
                compute_waves_barrier(cwcnt, cw_epoch, lane);    // every wave has read As
#pragma unroll
                for (int Jb = 0; Jb < 4; ++Jb) store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, lr[Jb], li, lq);
                compute_waves_barrier(cwcnt, cw_epoch, lane);
            }
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 4] = __builtin_amdgcn_s_memtime();
            // (L[j+1, j] leaves for global memory beside the tile factorisation below: ChainSide::extra)
            // tile (j+1, j+1) - L[j+1, j] L[j+1, j]^T: product from zero, ONE subtraction (as potrf_step)
            // (the ten lower blocks dealt evenly over the waves; L_jj has left Ts: its stores were drained by wg_publish; the tile
            //  (j+1, j+1) waits in Bs, prefetched or just loaded)
            // tile (j+1, j+1) - L[j+1, j] L[j+1, j]^T: only its first 16 columns here, one block per wave -- the other six blocks are
            // formed by waves 1 - 3 inside the tile routine, beside wave 0's first panel (ChainSide::load_panel)
            ts_epoch += 4;
            if (cw) {
                while (*tsread < ts_epoch) {}               // (long true: waves 4 - 7 read L_jj out of Ts right after the tile)
                asm volatile("" ::: "memory");
                const v4d b0 = diag_update_block(wave, 0, As, Bs, li, lq);
                store_d16(Ts + (16 * wave) * TLD, TLD, b0, li, lq);
            }
            cs.split_update = true; cs.used_target += 3;
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 6] = __builtin_amdgcn_s_memtime();
            cs.gL = pa.L + o1 * ld + oj;
            prefetch_for(j + 1);
            cs.dbg = pa.stamps ? pa.stamps + 8 * (j - pa.j0) + 5 : nullptr;
            tile_potrf_prepare(Wk, tid);
            __syncthreads();
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 7] = __builtin_amdgcn_s_memtime();
            tile_potrf_inv<true>(Ts, Xs, Wk, rinvs, tid, bad, cs);
            if (bad && tid == 0) atomicCAS(pa.info, 0, pa.blk);
            if (!cw) publish_tile(j + 1, fPL + (j + 1) * nt + j);      // X_{j+1,j+1} and (stored beside the tile) L[j+1, j]: drained, flagged
            if (pa.stamps && tid == 0) pa.stamps[8 * (j - pa.j0) + 8] = __builtin_amdgcn_s_memtime();
        }
        wg_publish(nullptr, nullptr, tid);                   // (every store of this workgroup has arrived before it counts itself out)
        return;
    }

    // ---------------------------------------------------------------------- a tile of the trailing block
    int r, c;
    persist_tile_of((int)blockIdx.x - 1, nt, pa.j0, pa.j1, pa.xrows, r, c);
    if (r < 0) return;
    const int nsteps = (c > pa.j0) ? ((r == c) ? c - 1 - pa.j0 : c - pa.j0) : 0;
    const int jb_end = (c == r) ? wave + 1 : 4;
    double* Sg = pa.S + (int64_t)r * 64 * ld + (int64_t)c * 64;
    v4d cpre[4];
    if (nsteps > 0) {
        // nobody else writes this tile in this launch: plain loads
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cpre[Jb][q] = (Jb < jb_end) ? Sg[(int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li] : 0.0;
    }
    for (int s = 0; s < nsteps; ++s) {
        const int j = pa.j0 + s;
        const int64_t oj = (int64_t)j * 64;
        const unsigned* w1 = (s > 0) ? fF + r * nt + j : nullptr;
        const unsigned* w2 = (s > 0 && c != r) ? fF + c * nt + j : nullptr;
        // the panel tiles of this step were final a step ago; X_jj is what arrives last: load them while waiting for it
        if (!wg_wait(w1, w2, nullptr, pa, okw, phase, tid)) return;
        tile_g2s_sc1(pa.S + (int64_t)r * 64 * ld + oj, ld, As, tid);
        if (c != r) tile_g2s_sc1(pa.S + (int64_t)c * 64 * ld + oj, ld, Bs, tid);
        if (!wg_wait(fD + j, nullptr, nullptr, pa, okw, phase, tid)) return;
        tile_g2s_sc1(pa.X + oj * ld + oj, ld, Xs, tid);
        __syncthreads();
        v4d lr[4], lc[4];
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) { lr[Jb] = zero; lc[Jb] = zero; }
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const int k = 8 * kg + 2 * lq;
            const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
            v2d bv = av;
            if (c != r) bv = *reinterpret_cast<const v2d*>(Bs + (16 * wave + li) * TLD + k);
#pragma unroll
            for (int Jb = kg / 2; Jb < 4; ++Jb) {
                const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
                lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, lr[Jb], 0, 0, 0);
                lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, lr[Jb], 0, 0, 0);
                if (c != r) {
                    lc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.x, xv.x, lc[Jb], 0, 0, 0);
                    lc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv.y, xv.y, lc[Jb], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // every wave is done reading As / Bs
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) {
            store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, lr[Jb], li, lq);
            if (c != r) store_d16(Bs + (16 * wave) * TLD + 16 * Jb, TLD, lc[Jb], li, lq);
        }
        __syncthreads();
        const bool panel_owner = (c == j + 1);             // (then r > c: the chain stores L[j+1, j] itself)
        if (panel_owner) tile_s2g_sc1(As, pa.L + (int64_t)r * 64 * ld + oj, ld, tid);
        const double* Lcs = (c != r) ? Bs : As;
        v4d pacc[4];
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) pacc[Jb] = zero;
        if (c != r) strip_nt<4>(As + (16 * wave + li) * TLD, Lcs, pacc, li, lq);
        else strip_nt_diag(wave, As + (16 * wave + li) * TLD, Lcs, pacc, li, lq);
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) cpre[Jb] = cpre[Jb] - pacc[Jb];
        if (s + 1 == nsteps) {
            // the tile is final: it is the panel tile of step c (diagonal tile: what the chain updates once more and factors)
            strip_store_sc1(Sg, ld, cpre, jb_end, false, wave, li, lq);
            wg_publish(fF + r * nt + c, panel_owner ? fPL + r * nt + j : nullptr, tid);
        } else if (panel_owner) {
            wg_publish(fPL + r * nt + j, nullptr, tid);    // (not reached: the panel owner's step c-1 is its last)
        } else {
            __syncthreads();                               // As / Bs are re-filled by the next step
        }
    }
    if (pa.tail_panel && c == pa.j1 - 1 && r > c) {
        // Larger blocks, the panel's last column: L[r, c] = S'[r, c] X_cc^T for the rows below the panel, here instead of in a
        // potrf_panel launch of its own (8 192 launches of a C5 factorisation).  The final tile waits in registers.
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) store_d16(As + (16 * wave) * TLD + 16 * Jb, TLD, cpre[Jb], li, lq);
        if (!wg_wait(fD + c, nullptr, nullptr, pa, okw, phase, tid)) return;
        const int64_t oc = (int64_t)c * 64;
        tile_g2s_sc1(pa.X + oc * ld + oc, ld, Xs, tid);
        __syncthreads();
        v4d lr[4];
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb) lr[Jb] = zero;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const int k = 8 * kg + 2 * lq;
            const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
#pragma unroll
            for (int Jb = kg / 2; Jb < 4; ++Jb) {
                const v2d xv = *reinterpret_cast<const v2d*>(Xs + (16 * Jb + li) * TLD + k);
                lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, xv.x, lr[Jb], 0, 0, 0);
                lr[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, xv.y, lr[Jb], 0, 0, 0);
            }
        }
        double* Lg = pa.L + (int64_t)r * 64 * ld + oc;
#pragma unroll
        for (int Jb = 0; Jb < 4; ++Jb)
#pragma unroll
            for (int q = 0; q < 4; ++q) Lg[(int64_t)(16 * wave + lq + 4 * q) * ld + 16 * Jb + li] = lr[Jb][q];
        return;
    }
    if (!pa.xrows) return;
    if (r == c) {
        if (pa.xrows != 2) return;
        r = c - 1;                                         // a diagonal tile's workgroup: the inverse tile (c - 1, j0) is its second job
        c = pa.j0;
    }

    // ---------------------------------------------------------------------- tile (r, c) of the block inverse
    //   T = sum_{p = c}^{r-1} L[r, p] X[p, c]  (p ascending, k in the slot order of xrow_strip), X[r, c] = -X_rr T
    v4d acc[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) acc[Jb] = zero;
    for (int p = c; p < r; ++p) {
        const int64_t op = (int64_t)p * 64;
        const unsigned* wx = (p == c) ? fD + c : fXF + p * nt + c;
        if (!wg_wait(fPL + r * nt + p, wx, nullptr, pa, okw, phase, tid)) return;
        tile_g2s_sc1(pa.L + (int64_t)r * 64 * ld + op, ld, As, tid);
        tile_g2s_sc1(pa.X + op * ld + (int64_t)c * 64, ld, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
            const int k = 8 * kg + 2 * lq;
            const v2d av = *reinterpret_cast<const v2d*>(As + (16 * wave + li) * TLD + k);
#pragma unroll
            for (int Jb = 0; Jb < 4; ++Jb) {
                acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, Bs[k * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
                acc[Jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, Bs[(k + 1) * TLD + 16 * Jb + li], acc[Jb], 0, 0, 0);
            }
        }
        __syncthreads();                                   // As / Bs are re-filled by the next term
    }
    if (!wg_wait(fD + r, nullptr, nullptr, pa, okw, phase, tid)) return;
    tile_g2s_sc1(pa.X + (int64_t)r * 64 * ld + (int64_t)r * 64, ld, As, tid);          // X_rr
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) store_d16(Bs + (16 * wave) * TLD + 16 * Jb, TLD, acc[Jb], li, lq);      // T
    __syncthreads();
    v4d res[4];
#pragma unroll
    for (int Jb = 0; Jb < 4; ++Jb) res[Jb] = zero;
    strip_tri_nn_w(wave, As + (16 * wave + li) * TLD, Bs, res, li, lq);           // X_rr lower triangular: the k groups 0 .. 2 wave + 1
    strip_store_sc1(pa.X + (int64_t)r * 64 * ld + (int64_t)c * 64, ld, res, 4, true, wave, li, lq);
    wg_publish(fXF + r * nt + c, nullptr, tid);
}

constexpr int POTRF_PERSIST_THREADS = 512;
template <bool UNUSED>
__global__ __launch_bounds__(POTRF_PERSIST_THREADS, 2) void potrf_persist(PersistArgs pa) {
    // (only the chain workgroup uses its waves 4 - 7; a barrier counts the waves that have not ended)
    if (blockIdx.x != 0 && threadIdx.x >= 256) return;
    potrf_persist_body(pa);                               // (every return in there is taken by the whole workgroup)
    // The last workgroup of a problem to get here leaves the flag words zero for the next launch: all its own stores are
    // acknowledged (vmcnt(0)), then one lane counts the workgroup out; whoever counts gridDim.x - 1 others before it knows that
    // nobody will look at a word of this launch again.
    __shared__ int last_out;
    unsigned* const flags = pa.flags + (int64_t)blockIdx.y * pa.flag_stride;
    const int nflags = persist_flag_count(pa.nt);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        last_out = __hip_atomic_fetch_add(flags + nflags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (last_out && threadIdx.x < 256)                     // (a worker workgroup has only its first four waves left)
        for (int i = threadIdx.x; i <= nflags; i += 256) st_flag(flags + i, 0u);
}

}  // namespace gmrf
