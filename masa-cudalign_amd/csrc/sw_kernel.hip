// MI355X (gfx950 / CDNA4) Stage-1 affine-gap SW/NW strip-wavefront kernel.
//
// Replaces the reference's device code (X/CUDAligner.cu: kernel_long_phase :941-1007,
// kernel_short_phase :745-824, kernel_single_phase :1100-1156, kernel_sw :276-289,
// kernel_check_max4 :396-410, kernel_load :441-459, kernel_flush :502-540) with a
// from-scratch design for 64-lane wavefronts:
//
//   * one wavefront owns a STRIP of 64*R consecutive DP rows (lane k: rows k*R..k*R+R-1, all
//     state in VGPRs) and sweeps it left->right over the whole partition width; lane k is k
//     columns behind lane k-1 (systolic skew), so every step each lane computes R cells of one
//     column and hands its bottom (H,F) to lane k+1 with a single DPP wave_shr:1 move;
//   * the horizontal bus row (H,F per column, HBM) is read in coalesced 64-column chunks,
//     staged through LDS and fed to lane 0; lane 63's outputs are collected in LDS and written
//     back in coalesced chunks (in place: a strip overwrites the bus row it consumed);
//   * strips are claimed dynamically (atomic ticket) by persistent wavefronts; strip s+1
//     follows strip s through a per-strip progress counter (write-through sc1 stores + drained
//     flag, MI355X_MICROARCH.md "Valid forms") -- no kernel launch per external diagonal;
//   * arithmetic is kept in the "t = H - (open+ext)" domain so that one subtraction per cell
//     feeds both E of the right neighbour and F of the lower neighbour, the substitution score
//     is one v_bfe_i32 from a per-row 8x4-bit profile register, H is v_max3 + v_max, and the
//     running best is folded with v_max3 -- ~9.5 int32 VALU ops per SW cell.
//
// Recurrence (bit-identical to CPUBlockProcessor.cpp:66-93 and CUDAligner.cu:276-289):
//   E = max(Hleft-3, Eleft)-2 ; F = max(Hup-3, Fup)-2 ; v = Hdiag + (c0!=c1 ? -3 : +1)
//   H = SW ? max(0,v,E,F) : max(v,E,F)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_kernel.h"

namespace mi355sw {

#define DBG(k, v) do { if (a->dbg != nullptr && lane == 0) st_agent(&a->dbg[k], (v)); } while (0)

#define GAP_FIRST 5   // DNA_GAP_OPEN + DNA_GAP_EXT (CUDAligner.hpp:92-98)
#define GAP_EXT 2
#define NEG_INF (-999999999)   // libmasaTypes.hpp:46

typedef unsigned int u32;
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ int wave_shr1(int old, int src) {
    // lane k <- src of lane k-1 ; lane 0 keeps `old` (DPP wave_shr:1, bound_ctrl off)
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int max3(int a, int b, int c) {
    return max(max(a, b), c);   // folds to v_max3_i32
}
__device__ __forceinline__ int ld_agent(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int2 ld_agent2(const int2* p) {
    unsigned long long x = __hip_atomic_load((const unsigned long long*) p, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    return make_int2((int) (u32) x, (int) (u32) (x >> 32));
}
__device__ __forceinline__ void st_agent2(int2* p, int2 v) {
    unsigned long long x = ((unsigned long long) (u32) v.y << 32) | (u32) v.x;
    __hip_atomic_store((unsigned long long*) p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-uniform polls: the loaded word is the same in every lane; readfirstlane tells the compiler so
// (scalar branches instead of EXEC-masked loops)
__device__ __forceinline__ int poll_agent(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int poll_sys(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ int ld_sys(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int2 ld_sys2(const int2* p) {
    unsigned long long x = __hip_atomic_load((const unsigned long long*) p, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM);
    return make_int2((int) (u32) x, (int) (u32) (x >> 32));
}

// Per-wave LDS staging area (all values in the t = H-5 domain).
struct __attribute__((aligned(16))) WaveLds {
    int2 in_tf[CHUNK + 1];    // (t,F) of the row above, columns [64c, 64c+64) (+1: prefetch slot)
    int2 out_tf[CHUNK];       // (t,F) of the emit row, written by the emit lane at step u
    unsigned char c1[2 * CHUNK + 8];   // seq1 shift codes: [0,64) previous chunk, [64,128) current
    int red[3 * 64];          // strip-end best reduction
};

template <int R>
struct LaneState {
    int tl[R];      // t (= H-5) of the cell to the left, per row
    int e[R];       // E of the cell to the left, per row
    int prof[R];    // PROFILE: 8 nibbles (c1 code -> score+5) ; else raw seq0 byte
    int tup_prev;   // t of (row above, previous column)
    int tbot, fbot; // bottom (t,F) produced at the previous step
    int best_t, best_r, best_j;
};

// One systolic step: every lane advances one column.  `u` is the step index inside the chunk.
template <int R, bool SW, bool PROFILE, bool MASKED, bool TRACK, bool EMIT_ANY, bool CMAX = false>
__device__ __forceinline__ void wave_step(LaneState<R>& st, WaveLds* lds, const int u, const int lane,
                                          const int jl /* column of this lane at u=0 */, const int n,
                                          const int nvalid, const int emit_lane, const int emit_row,
                                          int2& feed_io, int& c1_io, int* cmax = nullptr) {
    // ---- hand-off from the lane above (full EXEC); LDS reads are software-pipelined by one step ----
    const int2 feed = feed_io;                             // (t,F) of the row above: lane 0 only
    const int c1 = c1_io;
    feed_io = lds->in_tf[u + 1];                           // broadcast read for the next step
    c1_io = lds->c1[CHUNK + u + 1 - lane];
    const int tup = wave_shr1(feed.x, st.tbot);
    const int fup = wave_shr1(feed.y, st.fbot);

    bool active = true;
    if (MASKED) active = (u32) (jl + u) < (u32) n;
    if (active) {
        int diag = st.tup_prev;
        int upt = tup, upf = fup;
        int t_emit = 0, f_emit = 0;
        int mx = NEG_INF, tprev = NEG_INF;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int E = max(st.tl[r], st.e[r] - GAP_EXT);
            const int F = max(upt, upf - GAP_EXT);
            int v;
            if (PROFILE) {
                v = diag + __builtin_amdgcn_sbfe(st.prof[r], c1, 4);   // score+5 in {2,6}
            } else {
                v = diag + ((st.prof[r] != c1) ? 2 : 6);
            }
            int h = max3(v, E, F);
            if (SW) h = max(h, 0);
            const int t = h - GAP_FIRST;
            diag = st.tl[r];
            st.tl[r] = t;
            st.e[r] = E;
            upt = t;
            upf = F;
            if (TRACK || CMAX) { if (r & 1) mx = max3(mx, tprev, t); tprev = t; }
            if (EMIT_ANY) {
                t_emit = (r == emit_row) ? t : t_emit;
                f_emit = (r == emit_row) ? F : f_emit;
            }
        }
        if ((TRACK || CMAX) && (R & 1)) mx = max(mx, tprev);
        if (CMAX) *cmax = max(*cmax, mx);                   // the lane's maximum over the chunk (block pruning: what it can hand on)
        st.tup_prev = tup;
        st.tbot = upt;
        st.fbot = upf;
        if (!EMIT_ANY) { t_emit = upt; f_emit = upf; }
        if (lane == emit_lane) lds->out_tf[u] = make_int2(t_emit, f_emit);

        if (TRACK) {
            // rare path: exact canonical (max score, min i, min j) bookkeeping
            if (__any(mx >= st.best_t)) {
                const int j = jl + u;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int t = st.tl[r];
                    const bool upd = (r < nvalid) && ((t > st.best_t) || (t == st.best_t && r < st.best_r));
                    st.best_t = upd ? t : st.best_t;
                    st.best_r = upd ? r : st.best_r;
                    st.best_j = upd ? j : st.best_j;
                }
            }
        }
    }
}

// PRUNE (round 5; local alignments): block pruning in the int32 family too -- the reference prunes in every instantiation of
// its kernels (X/CUDAligner.cu:950-960); here a 64-step slab of the strip is skipped when nothing that enters it (the lanes'
// chunk maxima, the bus cells above it) can still reach the running best, exactly the packed kernel's local test
// (sw_kernel_pk16.inc, SW_PRUNE_MARGIN included).  Skipped cells read H = 0, E = F = -INF.  No window, no fast-forward: this
// family takes the pairs the packed kernel cannot (more than 14 common byte values) and its reruns.
#define SW32_PRUNE_MARGIN 8
template <int R, bool SW, bool PROFILE, bool TRACK, bool PRUNE = false>
__device__ __attribute__((noinline)) void process_strip(const KernelArgs* ap, const int s_in, WaveLds* lds, const int lane) {
    const UniformArgs a = uniform_args(ap);
    const int s = __builtin_amdgcn_readfirstlane(s_in);
    const int n = a->n;
    const int SH = 64 * R;
    const int nchunks = (n + 63 + CHUNK - 1) / CHUNK;
    const int row0 = a->strip_row0 + s * SH;          // first DP row of this strip (0-based)
    const int lrow0 = row0 + lane * R;                // first row of this lane
    const int rows_left = a->m - lrow0;
    const int nvalid = rows_left < 0 ? 0 : (rows_left > R ? R : rows_left);
    const int* prog_in = &a->progress[s];              // progress of the strip above
    // wait budget of the in-kernel polls: a band fed by another GPU may legitimately stand still for as long as
    // its first column takes to arrive (the first strip waits on the host counter, all others on that strip)
    const int spin_limit = a->first_col_ready != nullptr ? (1 << 30) : (1 << 24);
    int* prog_out = &a->progress[s + 1];

    // which (lane,row) is the row handed to the next strip / flushed as special row
    int emit_lane = 63, emit_row = R - 1;
    const bool ragged = (row0 + SH > a->m);
    if (ragged) {
        const int last = a->m - 1 - row0;              // last valid row inside the strip
        emit_lane = last / R;
        emit_row = last - emit_lane * R;
    }
    const bool last_strip = (row0 + SH >= a->m);
    int2* special = nullptr;
    if (a->special_interval_strips > 0 && a->special_rows != nullptr) {
        const int sg = a->strip_index0 + s + 1;       // strips completed once this one ends
        if (sg % a->special_interval_strips == 0 && (long long) sg * SH < a->m)
            special = a->special_rows + (long long) (sg / a->special_interval_strips - 1) * a->special_pitch;
    }
    int2* lastrow = (last_strip && a->last_row != nullptr) ? a->last_row : nullptr;

    // ---- per-lane state from the first column (InitialCellsReader semantics on device) ----
    // (a column delivered from outside has been awaited in claim_strip_common)
    LaneState<R> st;
#pragma unroll
    for (int r = 0; r < R; r++) {
        int h = 0, e = NEG_INF;
        if (a->first_col != nullptr) {
            const int g = lrow0 + r;
            if (g < a->m) {
                const int2 c = ld_sys2(&a->first_col[g + 1]);
                h = c.x; e = c.y;
            }
        }
        st.tl[r] = h - GAP_FIRST;
        st.e[r] = e;
        const int g = lrow0 + r;
        const int c0 = (g < a->m) ? (int) a->seq0[g] : a->pad_code;
        if (PROFILE) {
            // nibble k = score(c0, code k) + 5 : 6 on match, 2 otherwise; pad/foreign codes never match
            u32 p = 0x22222222u;
            if (c0 < a->n_match_codes) p += (4u << (4 * c0));
            st.prof[r] = (int) p;
        } else {
            st.prof[r] = c0 << a->seq0_shift;
        }
    }
    {
        int hd = 0;
        if (a->first_col != nullptr && lrow0 <= a->m) hd = ld_sys2(&a->first_col[lrow0]).x;
        st.tup_prev = hd - GAP_FIRST;
    }
    st.tbot = NEG_INF; st.fbot = NEG_INF;
    st.best_t = NEG_INF; st.best_r = R; st.best_j = -1;
    // block pruning: what the lane holds, bounded from above (t domain); the running best as of the last look
    int lane_entry = NEG_INF, gseen = NEG_INF, pruned_slabs = 0;
    if (PRUNE) {
#pragma unroll
        for (int r = 0; r < R; r++) lane_entry = max(lane_entry, st.tl[r]);
    }
    const int pr_rows = PRUNE ? a->prune_rows - 1 - row0 : 0, pr_cols = PRUNE ? a->prune_cols - 1 : 0;

    DBG(1, 1);
    // ---- sweep the strip ----
    for (int c = 0; c < nchunks; c++) {
        const int col0 = c * CHUNK;
        bool skip = false;
        DBG(2, c); DBG(3, 10);
        // (1) input chunk: wait for the strip above, then stage bus + seq1 codes into LDS
        {
            int need = col0 + CHUNK;
            if (need > n) need = n;
            if (col0 < n) {
                int spins = 0;
                while (poll_agent(prog_in) < need && spins < spin_limit) {
                    __builtin_amdgcn_s_sleep(2);
                    spins++;
                }
                if (spins >= spin_limit && lane == 0) atomicExch(a->error_flag, 1);
            }
            const int col = col0 + lane;
            int2 hf = make_int2(0, NEG_INF);
            unsigned char code = 0;
            if (col < n) {
                hf = ld_agent2(&a->bus[col]);
                code = a->seq1[col];
            }
            if (PRUNE) {
                gseen = max(gseen, poll_agent(a->gbest_in));
                if (!ragged && col0 >= 2 * CHUNK && col0 + CHUNK <= n) {
                    // (lane k is k columns behind lane 0: the slab touches columns col0 - 63 .. col0 + 63)
                    const int left = min(pr_rows + 2, pr_cols + 1 - (col0 - CHUNK));
                    const int e = max(lane_entry, hf.x - GAP_FIRST);
                    skip = !__any(e + left + SW32_PRUNE_MARGIN >= gseen);
                }
            }
            // shift the seq1 window: [64,128) -> [0,64), then the new chunk
            const unsigned char prev = lds->c1[CHUNK + lane];
            lds->c1[lane] = prev;
            lds->c1[CHUNK + lane] = code;
            lds->in_tf[lane] = make_int2(hf.x - GAP_FIRST, hf.y);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        DBG(3, 20);
        // (2) 64 systolic steps
        const int jl = col0 - lane;
        const bool masked = (c == 0) || (col0 + CHUNK - 1 >= n);
        int2 feed = lds->in_tf[0];
        int c1 = lds->c1[CHUNK - lane];
        int cm = NEG_INF;
        if (PRUNE && skip) {
            // the slab is not computed: every cell in it counts as H = 0 (t = -5), E = F = -INF
#pragma unroll
            for (int r = 0; r < R; r++) { st.tl[r] = -GAP_FIRST; st.e[r] = NEG_INF; }
            st.tup_prev = -GAP_FIRST; st.tbot = -GAP_FIRST; st.fbot = NEG_INF;
            lds->out_tf[lane] = make_int2(-GAP_FIRST, NEG_INF);
            cm = -GAP_FIRST;
            pruned_slabs++;
        } else if (ragged) {
#pragma unroll 2
            for (int u = 0; u < CHUNK; u++)
                wave_step<R, SW, PROFILE, true, TRACK, true, PRUNE>(st, lds, u, lane, jl, n, nvalid, emit_lane, emit_row, feed, c1, &cm);
        } else if (masked) {
#pragma unroll 2
            for (int u = 0; u < CHUNK; u++)
                wave_step<R, SW, PROFILE, true, TRACK, false, PRUNE>(st, lds, u, lane, jl, n, nvalid, 63, R - 1, feed, c1, &cm);
        } else {
#pragma unroll 4
            for (int u = 0; u < CHUNK; u++)
                wave_step<R, SW, PROFILE, false, TRACK, false, PRUNE>(st, lds, u, lane, jl, n, nvalid, 63, R - 1, feed, c1, &cm);
        }
        if (PRUNE) {
            // what the lane can hand on (a chunk it sat out entirely -- masked steps -- keeps what it held), and what the others learn
            lane_entry = max(cm, skip ? -GAP_FIRST : (masked ? lane_entry : NEG_INF));
            if (!skip && __any(cm > gseen)) {
                int w = cm;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) w = max(w, __shfl_xor(w, d));
                w = __builtin_amdgcn_readfirstlane(w);
                if (lane == 0) atomicMax(a->gbest, w);
                gseen = max(gseen, w);
            }
        }
        DBG(3, 30);
        // (3) output chunk: columns col0-emit_lane .. col0-emit_lane+63
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int2 tf = lds->out_tf[lane];
            const int col = col0 - emit_lane + lane;
            if (col >= 0 && col < n) {
                const int2 hf = make_int2(tf.x + GAP_FIRST, tf.y);
                st_agent2(&a->bus[col], hf);
                if (special != nullptr) special[col] = hf;
                if (lastrow != nullptr) lastrow[col] = hf;
            }
            // every store of this wave must have left before the flag (R1: drain, then flag)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int done = col0 - emit_lane + CHUNK;
            if (done > n) done = n;
            if (done < 0) done = 0;
            if (lane == 0) st_agent(prog_out, done);
        }
    }

    if (PRUNE && pruned_slabs > 0 && lane == 0 && a->pruned_slabs != nullptr) atomicAdd(a->pruned_slabs, (unsigned long long) pruned_slabs);
    DBG(3, 40);
    // ---- strip epilogue: last column, best score ----
    if (a->last_col != nullptr) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int g = lrow0 + r;
            if (g < a->m) a->last_col[g + 1] = make_int2(st.tl[r] + GAP_FIRST, st.e[r]);
        }
    }
    if (TRACK) {
        // canonical reduction over lanes: max t, then min row, then min j
        lds->red[lane] = st.best_t;
        lds->red[64 + lane] = (st.best_j >= 0) ? (lrow0 + st.best_r) : 0x7fffffff;
        lds->red[128 + lane] = st.best_j;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            int bt = NEG_INF, bi = 0x7fffffff, bj = -1;
            for (int k = 0; k < 64; k++) {
                const int t = lds->red[k], i = lds->red[64 + k], j = lds->red[128 + k];
                if (j >= 0 && (t > bt || (t == bt && (i < bi || (i == bi && j < bj))))) {
                    bt = t; bi = i; bj = j;
                }
            }
            int4 rec;
            rec.x = (bj >= 0) ? bt + GAP_FIRST : NEG_INF;
            rec.y = bi; rec.z = bj; rec.w = 1;
            a->strip_best[s] = rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// TWO wavefronts per SIMD for this family (the packed kernels keep one): its inner loop is v_max_i32 / v_add_u32 /
// v_bfe, and a non-packed "slow" instruction of one wavefront co-issues with a "fast" one of another
// (profiles/r02_valu_issue_rate.md, result 2: v_max_i32 next to v_add_u32 = full overlap) -- which no single wavefront
// can do for itself.  The register allocation is held between a third and a half of the SIMD's 512-entry file, so that
// exactly two of these wavefronts fit a SIMD wherever the dispatcher puts them (launch = 2 x SIMDs wavefronts): the
// strip chain runs at the speed of its slowest member, a SIMD with three would hold everybody up.
// Measured (tools/int32_perf.py, 4 M x 3 M SW, GCUPS one / two wavefronts per SIMD): 256-row strips 1543 / 2244 (related),
// 512-row 1779 / 2060 (related), 2023(1024-row) / 2468 (unrelated); 1024-row strips need 248 registers and LOSE with two
// (1718 / 1486): they keep a SIMD to themselves, and the runtime prefers the short strips for this family.
#ifndef SW32_WAVES_PER_SIMD
#define SW32_WAVES_PER_SIMD 2
#endif
template <int R, bool SW, bool PROFILE, bool TRACK, bool PRUNE = false>
__global__ void __launch_bounds__(64)
#if SW32_WAVES_PER_SIMD == 2
__attribute__((amdgpu_waves_per_eu(R == 16 ? 1 : 2, R == 16 ? 1 : 2)))
#endif
sw_strip_kernel(const KernelArgs* __restrict__ ap) {
    __shared__ WaveLds lds_store;
    WaveLds* lds = &lds_store;
    const int lane = threadIdx.x;
    const UniformArgs a = uniform_args(ap);
    const int num_strips = a->num_strips;
#if SW32_WAVES_PER_SIMD == 2
    // vector + accumulation registers together in (170, 256]: R = 4 uses ~85 vector registers, R = 8 ~135, R = 16 ~250
    if (R == 4) asm volatile("" ::: "a127");
    else if (R == 8) asm volatile("" ::: "a63");
    else asm volatile("" ::: "a255");               // 1024-row strips: one wavefront per SIMD, as the packed kernels
#else
    // one wavefront per SIMD, enforced: see sw_kernel_pk16.inc
    asm volatile("" ::: "a255");
#endif
    for (;;) {
        const int s = __builtin_amdgcn_readfirstlane(claim_strip_common(ap, lane, 64 * R));
        if (s >= num_strips) break;
        if (poll_agent(a->abort_flag) != 0 || (a->host_abort != nullptr && poll_sys(a->host_abort) != 0)) {
            // stopped: this wavefront is done (every strip still in flight holds an earlier ticket; see sw_kernel_pk16.inc)
            if (lane == 0) st_agent(&a->progress[s + 1], a->n);
            __builtin_amdgcn_wave_barrier();
            break;
        } else {
            process_strip<R, SW, PROFILE, TRACK, PRUNE>(ap, s, lds, lane);
        }
        complete_strip_common(ap, s, lane, 64 * R, true);
    }
}

// Device-side border initialisation: InitialCellsReader::read (InitialCellsReader.cpp:84-108)
__global__ void fill_bus_kernel(int2* bus, int n, int init_type, int start_offset) {
    const int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        int h = 0;
        if (init_type == INIT_WITH_GAPS) h = -GAP_EXT * (start_offset + j + 1) - 3;
        else if (init_type == INIT_WITH_GAPS_OPENED) h = -GAP_EXT * (start_offset + j + 1);
        bus[j] = make_int2(h, NEG_INF);
    }
}

__global__ void fill_int_kernel(int* p, long long count, int value) {
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) p[k] = value;
}

__global__ void fill_cells_kernel(int2* p, long long count, int2 value) {
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) p[k] = value;
}

template <int R>
static hipError_t launch_r(const KernelArgs* a, int grid, hipStream_t stream, bool sw, bool profile, bool track, bool prune) {
#define LAUNCH(SWV, PRV, TRV) \
    hipLaunchKernelGGL((sw_strip_kernel<R, SWV, PRV, TRV>), dim3(grid), dim3(64), 0, stream, a)
    if (sw && prune) {        // block pruning: local alignments with their best score tracked
        if (profile) hipLaunchKernelGGL((sw_strip_kernel<R, true, true, true, true>), dim3(grid), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((sw_strip_kernel<R, true, false, true, true>), dim3(grid), dim3(64), 0, stream, a);
    } else if (sw) {
        if (profile) { if (track) LAUNCH(true, true, true); else LAUNCH(true, true, false); }
        else         { if (track) LAUNCH(true, false, true); else LAUNCH(true, false, false); }
    } else {
        if (profile) { if (track) LAUNCH(false, true, true); else LAUNCH(false, true, false); }
        else         { if (track) LAUNCH(false, false, true); else LAUNCH(false, false, false); }
    }
#undef LAUNCH
    return hipGetLastError();
}

hipError_t launch_strip_kernel(const KernelArgs& a, KernelArgs* dargs, int rows_per_lane, int grid, hipStream_t stream,
                               bool sw, bool profile, bool track) {
    hipError_t e = hipMemcpyAsync(dargs, &a, sizeof(KernelArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);      // `a` is a host temporary
    if (e != hipSuccess) return e;
    const bool prune = a.prune != 0 && sw && track;
    switch (rows_per_lane) {
    case 4: return launch_r<4>(dargs, grid, stream, sw, profile, track, prune);
    case 8: return launch_r<8>(dargs, grid, stream, sw, profile, track, prune);
    case 16: return launch_r<16>(dargs, grid, stream, sw, profile, track, prune);
    default: return hipErrorInvalidValue;
    }
}

int strip_kernel_waves_per_simd(int rows_per_lane) { return (SW32_WAVES_PER_SIMD == 2 && rows_per_lane < 16) ? 2 : 1; }

hipError_t launch_fill_bus(int2* bus, int n, int init_type, int start_offset, hipStream_t stream) {
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_bus_kernel, dim3(blocks), dim3(256), 0, stream, bus, n, init_type, start_offset);
    return hipGetLastError();
}

hipError_t launch_fill_int(int* p, long long count, int value, hipStream_t stream) {
    long long b = (count + 255) / 256;
    int blocks = (int) (b > 2048 ? 2048 : (b < 1 ? 1 : b));
    hipLaunchKernelGGL(fill_int_kernel, dim3(blocks), dim3(256), 0, stream, p, count, value);
    return hipGetLastError();
}

hipError_t launch_fill_cells(int2* p, long long count, int h, int f, hipStream_t stream) {
    long long b = (count + 255) / 256;
    int blocks = (int) (b > 8192 ? 8192 : (b < 1 ? 1 : b));
    hipLaunchKernelGGL(fill_cells_kernel, dim3(blocks), dim3(256), 0, stream, p, count, make_int2(h, f));
    return hipGetLastError();
}

}  // namespace mi355sw
