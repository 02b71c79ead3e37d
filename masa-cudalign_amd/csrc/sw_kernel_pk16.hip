// MI355X (gfx950) Stage-1 SW strip kernel, PACKED 16-BIT form: two DP cells per VALU instruction.
//
// Why: a wave64 int32 VALU op issues once per 4 cycles per SIMD on this chip whatever the occupancy
// (tools/micro_valu.hip), so the int32 kernel (sw_kernel.hip, ~10.8 ops/cell) is pinned at its VALU
// ceiling.  v_pk_{add,max}_i16 / v_pk_min_u16 issue at the same rate and carry two cells.
//
// Geometry: every lane owns TWO row blocks of R rows, packed lo/hi in each 32-bit register: the LO
// block (rows 2kR..2kR+R-1 of the strip) and the HI block (the next R rows), HI running ONE column behind
// LO.  The wave is a 128-stage systolic array ("virtual lane" v = 2*lane+half is v columns behind virtual
// lane 0): the LO block receives the bottom (T,F) of the previous lane's HI block by DPP wave_shr:1, the HI
// block receives its own lane's LO bottom of the previous step; one v_alignbit_b32 merges the two.
// Strip height = 128*R rows.
//
// Arithmetic (exact restatement of CPUBlockProcessor.cpp:66-93 for SW).  Row r of a block is kept in the
// domain X^ = X + 2r - bias ("row-shifted"): stepping DOWN one row then costs nothing for the gap extension
// (F^[r] = max(T^[r-1], F^[r-1]) is max(Hu-5, Fu-2) in that domain), the SW floor becomes a per-row
// constant Z^[r] = 2r - bias, and the diagonal score absorbs the shift:
//   x = mask0[r] & mask1        one-hot base masks (bit 2+code) | bit 1, both halves at once
//   y = pk_min_u16(x, 6)        = 6 on match, 2 otherwise        (score + 3 + 2)
//   E = sat(pk_max(TL,E) - 2)                                    (max(Hl-5,E-2) = max(Tl,E)-2, same row)
//   g = pk_max(pk_max(diag+y, E), Z^[r])                         off the row-to-row dependency chain
//   F = pk_max(upT, upF) ; H = pk_max(g, F) ; T = H - 3          the chain: 3 dependent ops per row
// 10 packed ops per 2 cells (+1 for the running maximum).  Bottoms are handed to the next block converted
// to its "row -1" domain (-2R).  Values are 16-bit RELATIVE to a wave-uniform
// int32 bias that is re-centred at chunk boundaries (every 64 columns) on the wavefront's running maximum:
// H is Lipschitz (|dH| <= 5 per row/column step), so everything a wavefront touches in one chunk lies
// within a few thousand of that maximum while the 16-bit window is 65536 wide.  A per-chunk guard on the
// chunk maximum still reports an overflow (the host then re-runs with the int32 kernel) long before a wrap
// is possible.  -INF borders saturate at -32768 and stay there (saturating adds).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_kernel.h"

namespace mi355sw {

#define DBG16(k, v) do { if (a->dbg != nullptr && lane == 0) __hip_atomic_store(&a->dbg[k], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)

#define NEG_INF (-999999999)
#define T_OFF 3            // T = H - 3
#define PRIO_CHUNKS 4       // chunks of every strip that run at raised wave priority
#define REBASE_HI 20000     // chunk maximum (relative) above which the window is moved up ...
#define REBASE_TO 8000      // ... so that the maximum sits here
#define REBASE_LO (-8000)   // NW only: chunk maximum below which the window is moved down
#ifndef PK16_HALFTRACK
#define PK16_HALFTRACK 1  // fast pass accumulates the chunk maximum on odd rows only (even rows: bound +5)
#endif
#ifndef PK16_SVC_A
#define PK16_SVC_A 16     // step of a chunk at which the previous chunk's stores are published and the next inputs requested
#define PK16_SVC_B 0      // second chance for the input prefetch (0: none)
#endif
#ifndef PK16_EXACT_MODE
#define PK16_EXACT_MODE 1  // after a replayed chunk, run the following chunks in the exact code directly
#endif
#ifndef PK16_UNROLL
#define PK16_UNROLL 8     // steps per loop body of the 64-step chunk loop (measured: 8 beats 2 and 4 by 2-4 %)
#endif
#define GUARD16 30000      // chunk maximum above this => overflow report (wrap needs 32767)

typedef unsigned int u32;
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 as_s2(int x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ int as_i(s2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 padd_sat(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s2 splat(int v) { s2 r = {(short) v, (short) v}; return r; }
__device__ __forceinline__ int pack(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int clamp16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
__device__ __forceinline__ int lo16(int x) { return __builtin_amdgcn_sbfe(x, 0, 16); }
__device__ __forceinline__ int hi16(int x) { return x >> 16; }

__device__ __forceinline__ int wave_shr1_16(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int poll_agent16(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int poll_sys16(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_agent16(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int2 ld_agent2_16(const int2* p) {
    unsigned long long x = __hip_atomic_load((const unsigned long long*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_int2((int) (u32) x, (int) (u32) (x >> 32));
}
__device__ __forceinline__ int2 ld_sys2_16(const int2* p) {
    unsigned long long x = __hip_atomic_load((const unsigned long long*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_int2((int) (u32) x, (int) (u32) (x >> 32));
}
__device__ __forceinline__ int ld_u8_16(const unsigned char* p) {
    return (int) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent2_16(int2* p, int2 v) {
    unsigned long long x = ((unsigned long long) (u32) v.y << 32) | (u32) v.x;
    __hip_atomic_store((unsigned long long*) p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum { WIN = 128 };   // seq1 window: 128 columns of history in front of the 64-column chunk

struct __attribute__((aligned(16))) WaveLds16 {
    int2 in_tf[CHUNK + 1];      // (T16<<16, F16<<16) of the row above (value in the HIGH half: DPP `old` of lane 0)
    int2 out_tf[CHUNK];         // packed (T,F) words of the emit lane, one per step
    int2 dump[64];              // write-only slots of the non-emitting lanes (no EXEC toggling per step)
    int c1w[WIN + CHUNK + 8];   // per column j: mask(j) | mask(j-1)<<16
    int c1s[WIN + CHUNK + 8];   // per column j: v_perm_b32 selector {code(j), zero, 4+code(j-1), zero} (table form)
    int red[3 * 64];
};

template <int R>
struct Lane16 {
    s2 TL[R];       // T of the cell to the left      (lo: LO block row r, hi: HI block row r)
    s2 E[R];        // E of the cell to the left
    int M0[R];      // one-hot base masks of the two rows
    int TLO[R], THI[R];   // table form of the same: byte c = score of the row against column code c (c < 4)
    s2 tup_prev;    // T of (row above the block, previous column)
    s2 tbot, fbot;  // bottoms produced at the previous step
    int best_t, best_r, best_j;   // best T (true, 32-bit), row index inside the lane (0..2R-1), column
};

template <int R, bool MASKED, bool TRACK, bool EMIT_ANY, bool HALF, bool PERM, bool SWF>
__device__ __forceinline__ void wave_step16(Lane16<R>& st, WaveLds16* lds, const int u, const int lane,
                                            const int jl /* LO column of this lane at u=0 */, const int n,
                                            const int nvalid_lo, const int nvalid_hi, const int emit_lane,
                                            const int emit_row, const s2 (&Z)[R], const int bias, int2& feed_io, int& c1_io,
                                            s2 (&cm)[R], int2* out_base, const int out_stride, const bool all_rows_valid) {
    const int2 feed = feed_io;
    const int c1p = c1_io;
    feed_io = lds->in_tf[u + 1];
    c1_io = PERM ? lds->c1s[WIN + u + 1 - 2 * lane] : lds->c1w[WIN + u + 1 - 2 * lane];
    // hand-off: LO <- previous lane's HI bottom (DPP), HI <- own LO bottom; lane 0 LO <- bus feed
    const int dT = wave_shr1_16(feed.x, as_i(st.tbot));
    const int dF = wave_shr1_16(feed.y, as_i(st.fbot));
    const s2 tup = as_s2(__builtin_amdgcn_alignbit(as_i(st.tbot), dT, 16));
    s2 upT = tup;
    s2 upF = as_s2(__builtin_amdgcn_alignbit(as_i(st.fbot), dF, 16));
    s2 diag = st.tup_prev;

    int vmask = -1;
    if (MASKED) {
        const int j = jl + u;                                  // LO column; HI is at j-1
        const bool vlo = (u32) j < (u32) n;
        const bool vhi = (u32) (j - 1) < (u32) n;
        vmask = (vlo ? 0xffff : 0) | (vhi ? 0xffff0000 : 0);
    }
    const s2 m2 = splat(-2), m3 = splat(-T_OFF);
    const us2 six = {6, 6};
    s2 t_emit = splat(0), f_emit = splat(0);
    s2 ms = splat(-32768);                                   // TRACK only: true (unshifted) maximum of the step
    s2 newT[R];
    s2 Hbot = splat(0);
#pragma unroll
    for (int r = 0; r < R; r++) {
        // score + 5 of the two cells: one byte permute when every column in reach is one of <= 4 plain
        // codes (table form), else one-hot AND + min
        us2 y;
        if (PERM) {
            y = __builtin_bit_cast(us2, __builtin_amdgcn_perm((u32) st.THI[r], (u32) st.TLO[r], (u32) c1p));
        } else {
            const int x = st.M0[r] & c1p;
            y = __builtin_elementwise_min(__builtin_bit_cast(us2, x), six);
        }
        s2 Ev = padd_sat(pmax(st.TL[r], st.E[r]), m2);
        const s2 v = diag + __builtin_bit_cast(s2, y);
        const s2 g = SWF ? pmax(pmax(v, Ev), Z[r]) : pmax(v, Ev);   // off the row-to-row critical chain (Z: SW floor)
        const s2 Fv = pmax(upT, upF);                      // chain: max, max, add per row
        const s2 H = pmax(g, Fv);
        s2 T = H + m3;
        diag = st.TL[r];
        if (MASKED) {
            T = as_s2((as_i(T) & vmask) | (as_i(st.TL[r]) & ~vmask));
            Ev = as_s2((as_i(Ev) & vmask) | (as_i(st.E[r]) & ~vmask));
        }
        st.TL[r] = T;
        st.E[r] = Ev;
        newT[r] = T;
        upT = T;
        upF = Fv;
        Hbot = H;
        // HALF: H(i,j) <= H(i+1,j) + 5 (the row below can always open a gap), so the odd rows bound the
        // even ones; the chunk test adds the 5 and an exact replay decides
        if (!HALF || (r & 1)) {
            if (MASKED) cm[r] = pmax(cm[r], as_s2((as_i(T) & vmask) | (0x80008000 & ~vmask)));
            else cm[r] = pmax(cm[r], T);
        }
        if (TRACK) {
            const s2 tt = T + splat(-2 * r);
            if (MASKED) ms = pmax(ms, as_s2((as_i(tt) & vmask) | (0x80008000 & ~vmask)));
            else ms = pmax(ms, tt);
        }
        if (EMIT_ANY) {
            t_emit = (r == emit_row) ? T : t_emit;
            f_emit = (r == emit_row) ? Fv : f_emit;
        }
    }
    // bottoms for the next block, converted to its "row -1" domain (2(R-1) -> -2); saturating, so the
    // -INF image stays put.  tbot comes straight from H (parallel to T = H - 3, not behind it).
    const s2 tb = MASKED ? padd_sat(upT, splat(-2 * R)) : padd_sat(Hbot, splat(-T_OFF - 2 * R));
    const s2 fb = padd_sat(upF, splat(-2 * R));
    if (MASKED) {
        st.tup_prev = as_s2((as_i(tup) & vmask) | (as_i(st.tup_prev) & ~vmask));
        st.tbot = as_s2((as_i(tb) & vmask) | (as_i(st.tbot) & ~vmask));
        st.fbot = as_s2((as_i(fb) & vmask) | (as_i(st.fbot) & ~vmask));
    } else {
        st.tup_prev = tup;
        st.tbot = tb;
        st.fbot = fb;
    }
    if (!EMIT_ANY) { t_emit = upT; f_emit = upF; }
    // all lanes store (no s_and_saveexec / s_or per step): lane `emit_lane` into out_tf[u], the others into dump[lane]
    out_base[out_stride * u] = make_int2(as_i(t_emit), as_i(f_emit));

    if (TRACK) {
        // cheap per-step test on max(lo,hi); rare exact path keeps the canonical (max, min i, min j) cell
        const int mi = as_i(ms);
        const int m = max(lo16(mi), hi16(mi)) + bias;
        if (__any(m >= st.best_t)) {
            const int j = jl + u;
            if (all_rows_valid) {
                // Only the step's best cell of the lane can displace the running best: the larger half
                // maximum (LO on a tie: smaller row), at the FIRST row that reaches it.  Packed search for
                // that row: (ms - tt) is 0 exactly there.
                us2 fr = {0x7fff, 0x7fff};
                const us2 one = {1, 1}, big = {0x4000, 0x4000};
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const s2 tt = newT[r] + splat(-2 * r);
                    const us2 d = __builtin_bit_cast(us2, ms - tt);
                    const us2 z = __builtin_elementwise_min(d, one);
                    const us2 rv = {(unsigned short) r, (unsigned short) r};
                    fr = __builtin_elementwise_min(fr, (us2) (z * big + rv));
                }
                const int fi = __builtin_bit_cast(int, fr);
                const int mlo = lo16(mi), mhi = hi16(mi);
                const bool vlo = MASKED ? ((vmask & 1) != 0) : true;
                const bool vhi = MASKED ? (((vmask >> 16) & 1) != 0) : true;
                const bool use_hi = vhi && (!vlo || mhi > mlo);
                const int t = (use_hi ? mhi : mlo) + bias;
                const int rr = use_hi ? (R + (int) ((u32) fi >> 16)) : (fi & 0xffff);
                const bool upd = (use_hi ? vhi : vlo) && ((t > st.best_t) || (t == st.best_t && rr < st.best_r));
                st.best_t = upd ? t : st.best_t;
                st.best_r = upd ? rr : st.best_r;
                st.best_j = upd ? (j - (use_hi ? 1 : 0)) : st.best_j;
            } else
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int nv = half ? nvalid_hi : nvalid_lo;
                const bool hv = MASKED ? (((vmask >> (16 * half)) & 1) != 0) : true;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int w = as_i(newT[r]);
                    const int t = (half ? hi16(w) : lo16(w)) + bias - 2 * r;
                    const int rr = half * R + r;
                    const bool upd = hv && (r < nv) && ((t > st.best_t) || (t == st.best_t && rr < st.best_r));
                    st.best_t = upd ? t : st.best_t;
                    st.best_r = upd ? rr : st.best_r;
                    st.best_j = upd ? (j - half) : st.best_j;
                }
            }
        }
    }
}

// true (unshifted) maximum T of a chunk from the per-row accumulators
template <int R, bool HALF>
__device__ __forceinline__ int chunk_max16(const s2 (&cm)[R]) {
    int v = -32768 - 2 * R;
#pragma unroll
    for (int r = HALF ? 1 : 0; r < R; r += HALF ? 2 : 1) {
        const int w = as_i(cm[r]);
        v = max(v, max(lo16(w), hi16(w)) - 2 * r);
    }
    return v;
}

// 64 systolic steps of one chunk (reads the staged inputs from LDS, leaves the emit row in out_tf)
struct NoService16 { __device__ __forceinline__ void operator()(int) const {} };

template <int R, bool MASKED, bool TRACKSTEP, bool EMIT_ANY, bool HALF, bool PERM, bool SWF, typename Svc>
__device__ __forceinline__ void run_chunk16(Lane16<R>& st, WaveLds16* lds, const int lane, const int jl, const int n,
                                            const int nvalid_lo, const int nvalid_hi, const int emit_lane,
                                            const int emit_row, const s2 (&Z)[R], const int bias, s2 (&cmax)[R],
                                            Svc&& service, const bool all_rows_valid = true) {
    int2 feed = lds->in_tf[0];
    int c1 = PERM ? lds->c1s[WIN - 2 * lane] : lds->c1w[WIN - 2 * lane];
    int2* out_base = (lane == emit_lane) ? &lds->out_tf[0] : &lds->dump[lane];
    const int out_stride = (lane == emit_lane) ? 1 : 0;
#pragma unroll 1
    for (int ub = 0; ub < CHUNK; ub += PK16_UNROLL) {
        // in-chunk service points: wave-uniform branches at a loop-body boundary, where the schedule is cut anyway
        if (ub == PK16_SVC_A) service(0);
        if (PK16_SVC_B > 0 && ub == PK16_SVC_B) service(1);
#pragma unroll
        for (int k = 0; k < PK16_UNROLL; k++)
            wave_step16<R, MASKED, TRACKSTEP, EMIT_ANY, HALF, PERM, SWF>(st, lds, ub + k, lane, jl, n, nvalid_lo, nvalid_hi, emit_lane, emit_row, Z, bias, feed, c1, cmax, out_base, out_stride, all_rows_valid);
    }
}

template <int R, bool TRACK, bool SWF>
__device__ __attribute__((noinline)) void process_strip16(const KernelArgs* ap, const int s_in, WaveLds16* lds, const int lane) {
    const UniformArgs a = uniform_args(ap);
    const int s = __builtin_amdgcn_readfirstlane(s_in);
    const int n = a->n;
    const int SH = 128 * R;
    const int nchunks = (n + 127 + CHUNK - 1) / CHUNK;
    int bias = 0;                                         // wave-uniform int32 bias of the 16-bit window

    const int row0 = a->strip_row0 + s * SH;
    const int lrow_lo = row0 + (2 * lane) * R;            // first row of the LO block
    const int lrow_hi = lrow_lo + R;                      // first row of the HI block
    int nvalid_lo = a->m - lrow_lo; nvalid_lo = nvalid_lo < 0 ? 0 : (nvalid_lo > R ? R : nvalid_lo);
    int nvalid_hi = a->m - lrow_hi; nvalid_hi = nvalid_hi < 0 ? 0 : (nvalid_hi > R ? R : nvalid_hi);
    const int* prog_in = &a->progress[s];
    int* prog_out = &a->progress[s + 1];

    // emitting virtual lane / row: the strip's bottom row, or DP row m-1 for the ragged last strip
    int emit_v = 127, emit_row = R - 1;
    const bool ragged = (row0 + SH > a->m);
    const bool emit_any = ragged && a->last_row != nullptr;   // otherwise nobody reads a ragged strip's bus row
    if (emit_any) {
        const int last = a->m - 1 - row0;
        emit_v = last / R;
        emit_row = last - emit_v * R;
    }
    const int emit_lane = emit_v >> 1, emit_half = emit_v & 1;
    const bool last_strip = (row0 + SH >= a->m);
    int2* special = nullptr;
    if (a->special_interval_strips > 0 && a->special_rows != nullptr) {
        const int sg = a->strip_index0 + s + 1;
        if (sg % a->special_interval_strips == 0 && (long long) sg * SH < a->m)
            special = a->special_rows + (long long) (sg / a->special_interval_strips - 1) * a->special_pitch;
    }
    int2* lastrow = (last_strip && a->last_row != nullptr) ? a->last_row : nullptr;
    int2* ckpt = nullptr;
    if (a->ckpt_rows != nullptr && a->ckpt_interval_strips > 0) {
        const int sg = a->strip_index0 + s + 1;
        if (sg % a->ckpt_interval_strips == 0 && !last_strip)
            ckpt = a->ckpt_rows + (long long) (sg / a->ckpt_interval_strips) * a->ckpt_pitch;
    }

    bool overflow = false;
    // ---- first column ----
    if (a->first_col != nullptr && a->first_col_ready != nullptr) {
        int need = row0 + SH;
        if (need > a->m) need = a->m;
        int spins = 0;
        while (poll_sys16(a->first_col_ready) < need && poll_agent16(a->abort_flag) == 0 && spins < (1 << 26)) {
            __builtin_amdgcn_s_sleep(32);
            spins++;
        }
        if (spins >= (1 << 26) && lane == 0) atomicExch(a->error_flag, 2);
    }
    Lane16<R> st;
    int h0[2 * R], e0[2 * R];
    int hmax = SWF ? 0 : NEG_INF;
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int g = (half ? lrow_hi : lrow_lo) + r;
            // rows past m: 0 is a harmless start under the SW floor; without a floor it could sit far above
            // the window, so they start at the -INF image instead
            int h = (SWF || a->first_col == nullptr) ? 0 : NEG_INF, ee = NEG_INF;
            if (a->first_col != nullptr && g < a->m) {
                const int2 c = ld_sys2_16(&a->first_col[g + 1]);
                h = c.x; ee = c.y;
            }
            h0[2 * r + half] = h; e0[2 * r + half] = ee;
            if (SWF || g < a->m) hmax = max(hmax, h);
        }
    }
    // initial window: centred on the largest first-column score of the strip (0 for zero borders)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hmax = max(hmax, __shfl_xor(hmax, d));
    // SW: the window never goes below the floor (bias >= 0); NW/semi-global: it follows the scores down too
    if (SWF) bias = __builtin_amdgcn_readfirstlane(hmax > REBASE_HI ? hmax - REBASE_TO : 0);
    else bias = __builtin_amdgcn_readfirstlane((hmax > REBASE_HI || hmax < REBASE_LO) ? hmax - REBASE_TO : 0);
    s2 Z[R];                                              // SW floor (H = 0) of every row in its shifted domain
#pragma unroll
    for (int r = 0; r < R; r++) Z[r] = splat(clamp16(2 * r - bias));
#pragma unroll
    for (int r = 0; r < R; r++) {
        int mk[2], tb[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int g = (half ? lrow_hi : lrow_lo) + r;
            const int c0 = (g < a->m) ? (int) a->seq0[g] : a->pad_code;
            mk[half] = ((c0 < a->n_match_codes) ? (4 << c0) : 0) | 2;   // bit 1: the constant part of the score
            tb[half] = 0x02020202 | ((c0 < a->n_match_codes && c0 < 4) ? (4 << (8 * c0)) : 0);
        }
        st.TL[r] = as_s2(pack(clamp16(h0[2 * r] - T_OFF - bias + 2 * r), clamp16(h0[2 * r + 1] - T_OFF - bias + 2 * r)));
        st.E[r] = as_s2(pack(clamp16(e0[2 * r] - bias + 2 * r), clamp16(e0[2 * r + 1] - bias + 2 * r)));
        st.M0[r] = pack(mk[0], mk[1]);
        st.TLO[r] = tb[0];
        st.THI[r] = tb[1];
    }
    {
        int hd[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int g0 = half ? lrow_hi : lrow_lo;
            int h = (SWF || a->first_col == nullptr) ? 0 : NEG_INF;
            if (a->first_col != nullptr && g0 <= a->m) h = ld_sys2_16(&a->first_col[g0]).x;
            hd[half] = clamp16(h - T_OFF - bias - 2);        // "row -1" of the block
        }
        st.tup_prev = as_s2(pack(hd[0], hd[1]));
    }
    st.tbot = splat(-32768);
    st.fbot = splat(-32768);
    st.best_t = NEG_INF; st.best_r = 2 * R; st.best_j = -1;
    int lane_max = NEG_INF;                               // best H-3 of this lane, true 32-bit
    int gseen = NEG_INF;

    // seq1 window starts empty
    lds->c1w[lane] = 0x00020002; lds->c1w[64 + lane] = 0x00020002; lds->c1w[128 + lane] = 0x00020002;
    if (lane < 8) lds->c1w[192 + lane] = 0x00020002;
    lds->c1s[lane] = 0x0c040c00; lds->c1s[64 + lane] = 0x0c040c00; lds->c1s[128 + lane] = 0x0c040c00;
    if (lane < 8) lds->c1s[192 + lane] = 0x0c040c00;
    bool simple1 = false, simple2 = false;                // the previous two chunks held only plain codes (< 4)

    DBG16(1, 1);
    long long tr_start = 0, tr_first = 0; int tr_spins = 0;
    if (a->trace != nullptr) tr_start = __builtin_amdgcn_s_memrealtime();
    // The first chunks of a strip are the critical path of the pipeline's start-up: the next strip cannot
    // begin before they are published.  Run them at raised priority so that they proceed at single-wave
    // speed instead of a quarter of the SIMD (measured: hop 250 us -> ~50 us with 4 waves per SIMD).
    __builtin_amdgcn_s_setprio(3);
    bool simple0 = false;
    bool exact_mode = false;                              // wave-uniform: run the exact-tracking code without a fast pass first
    bool pf_valid = false;                                // next chunk's inputs are in pf_* (wave-uniform)
    int2 pf_hf = make_int2(0, NEG_INF);
    int pf_code = 255, pf_codep = 255;
    int prog_async = 0;                                   // predecessor's progress, requested at the chunk start
    int gb_async = TRACK ? __hip_atomic_load(a->gbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int flag_pending = -1;                                // progress value whose bus stores are still in flight
    for (int c = 0; c < nchunks; c++) {
        const int col0 = c * CHUNK;
        // graded: the younger the strip, the closer it is to the start-up front of the pipeline
        if (c == 8) __builtin_amdgcn_s_setprio(2);
        else if (c == 48) __builtin_amdgcn_s_setprio(1);
        else if (c == 256) __builtin_amdgcn_s_setprio(0);
        DBG16(2, c); DBG16(3, 10);
        const bool trc = (a->trace != nullptr) && (c == 1000 || c == 1);
        long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
        if (trc) q0 = __builtin_amdgcn_s_memrealtime();
        // (1) stage the input chunk: prefetched during the previous chunk when the predecessor was far enough
        //     ahead (the usual case), otherwise wait for it here
        {
            const int col = col0 + lane;
            int2 hf = make_int2(0, NEG_INF);
            int code = 255, codep = 255;
            if (pf_valid) {
                hf = pf_hf; code = pf_code; codep = pf_codep;
            } else {
                int need = col0 + CHUNK;
                if (need > n) need = n;
                if (col0 < n) {
                    int spins = 0;
                    while (poll_agent16(prog_in) < need && spins < (1 << 24)) {
                        __builtin_amdgcn_s_sleep(2);
                        spins++;
                    }
                    tr_spins += spins;
                    if (spins >= (1 << 24) && lane == 0) atomicExch(a->error_flag, 1);
                }
                if (col < n) {
                    hf = ld_agent2_16(&a->bus[col]);
                    code = ld_u8_16(&a->seq1[col]) >> 2;      // seq1 holds code*4 (shift form of the int32 kernel)
                }
                if (col >= 1 && col - 1 < n) codep = ld_u8_16(&a->seq1[col - 1]) >> 2;
            }
            pf_valid = false;
            if (TRACK) {
                // Seed: a cell below the best score already found anywhere can never be the answer, so every
                // lane starts from the global running best (ties are still taken: the test below is >=).
                // Without it each strip spends its first chunks in the exact path and the start-up delay of
                // every hop of the strip pipeline triples.  The value was requested one chunk ago.
                gseen = __builtin_amdgcn_readfirstlane(gb_async);
                if (gseen > st.best_t) { st.best_t = gseen; st.best_r = 2 * R; st.best_j = -1; }
                gb_async = __hip_atomic_load(a->gbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (trc) q1 = __builtin_amdgcn_s_memrealtime();
            const int mk = ((code < a->n_match_codes) ? (4 << code) : 0) | 2;
            const int mkp = ((codep < a->n_match_codes) ? (4 << codep) : 0) | 2;
            if (col < n && hf.x - T_OFF - bias > GUARD16) overflow = true;
            if (!SWF && col < n && hf.x - T_OFF - bias < -GUARD16) overflow = true;   // no floor to hide behind
            // shift the window by one chunk, then append
            const int w0 = lds->c1w[64 + lane];
            const int w1 = lds->c1w[128 + lane];
            lds->c1w[lane] = w0;
            lds->c1w[64 + lane] = w1;
            lds->c1w[128 + lane] = mk | (mkp << 16);
            const int v0 = lds->c1s[64 + lane];
            const int v1 = lds->c1s[128 + lane];
            lds->c1s[lane] = v0;
            lds->c1s[64 + lane] = v1;
            lds->c1s[128 + lane] = (code & 3) | 0x0c000c00 | ((4 + (codep & 3)) << 16);
            simple2 = simple1; simple1 = simple0;
            simple0 = (col0 + CHUNK <= n) && !__any(code >= 4 || code >= a->n_match_codes);
            lds->in_tf[lane] = make_int2(clamp16(hf.x - T_OFF - bias - 2) << 16, clamp16(hf.y - bias - 2) << 16);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ask for the predecessor's progress now, look at the answer in the middle of the chunk
            prog_async = __hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (trc) q2 = __builtin_amdgcn_s_memrealtime();
        DBG16(3, 20);
        // (2) 64 systolic steps
        const int jl = col0 - 2 * lane;
        const bool masked = (col0 - 127 < 0) || (col0 + CHUNK - 1 >= n);
        s2 cmax[R];
#pragma unroll
        for (int r = 0; r < R; r++) cmax[r] = splat(-32768);
        int rebias_to = bias;
        // Fast pass: no position bookkeeping at all.  Only if some lane's chunk maximum reaches its
        // (globally seeded) threshold is the chunk replayed from a register snapshot with the exact
        // per-step bookkeeping -- a rare event once the running best is above the background level.
        // While chunks keep producing new candidates (start of the run, a ridge of a real alignment) the
        // fast pass would be thrown away every time: such stretches run the exact code directly
        // (exact_mode), one pass per chunk, and return to the fast pass after a chunk without any update.
        Lane16<R> snap;
        if (TRACK && !exact_mode) {
#pragma unroll
            for (int r = 0; r < R; r++) { snap.TL[r] = st.TL[r]; snap.E[r] = st.E[r]; }
            snap.tup_prev = st.tup_prev; snap.tbot = st.tbot; snap.fbot = st.fbot;
        }
        constexpr bool HALF = TRACK && (PK16_HALFTRACK != 0);
        constexpr int SLACK = HALF ? 5 : 0;                // what an untracked (even) row can exceed its neighbour by
        // The ragged last strip runs the ordinary code (rows past m are ordinary cells that never match);
        // only when its DP row m-1 is wanted (last row) does the emit position have to move off the bottom.
        // ---- mid-chunk service: everything that needs a memory round trip happens here, half a chunk after
        //      it was requested, so that no latency is exposed at the chunk boundary ----
        auto service = [&](const int second) {
            // (a) the previous chunk's bus stores have landed by now: publish them
            if (!second) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (flag_pending >= 0) {
                    if (lane == 0) st_agent16(prog_out, flag_pending);
                    flag_pending = -1;
                }
            }
            // (b) is the next input chunk already there?  Then fetch it now (the loads are issued after the
            //     progress value was observed, so they see the published data); otherwise ask again for the
            //     second service point
            const int ncol0 = col0 + CHUNK;
            if (c + 1 < nchunks && !pf_valid) {
                int need = ncol0 + CHUNK;
                if (need > n) need = n;
                const int avail = __builtin_amdgcn_readfirstlane(prog_async);
                if (ncol0 >= n || avail >= need) {
                    const int col = ncol0 + lane;
                    pf_hf = make_int2(0, NEG_INF); pf_code = 255; pf_codep = 255;
                    if (col < n) {
                        pf_hf = ld_agent2_16(&a->bus[col]);
                        pf_code = ld_u8_16(&a->seq1[col]) >> 2;
                    }
                    if (col >= 1 && col - 1 < n) pf_codep = ld_u8_16(&a->seq1[col - 1]) >> 2;
                    pf_valid = true;
                } else if (!second) {
                    prog_async = __hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        const bool use_perm = !masked && !emit_any && simple0 && simple1 && simple2;
        if (TRACK && exact_mode) {
            const int bt0 = st.best_t, bj0 = st.best_j, br0 = st.best_r;
            if (emit_any) run_chunk16<R, true, true, true, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, emit_lane, emit_row, Z, bias, cmax, service, !ragged);
            else if (masked) run_chunk16<R, true, true, false, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax, service, !ragged);
            else run_chunk16<R, false, true, false, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax, service, !ragged);
            exact_mode = __any(st.best_t != bt0 || st.best_j != bj0 || st.best_r != br0);
        } else {
        if (use_perm) run_chunk16<R, false, false, false, HALF, true, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax, service);
        else if (emit_any && masked) run_chunk16<R, true, false, true, HALF, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, emit_lane, emit_row, Z, bias, cmax, service);
        else if (emit_any) run_chunk16<R, false, false, true, HALF, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, emit_lane, emit_row, Z, bias, cmax, service);
        else if (masked) run_chunk16<R, true, false, false, HALF, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax, service);
        else run_chunk16<R, false, false, false, HALF, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax, service);
        if (TRACK) {
            if (__any(chunk_max16<R, HALF>(cmax) + SLACK + bias >= st.best_t)) {
#pragma unroll
                for (int r = 0; r < R; r++) { st.TL[r] = snap.TL[r]; st.E[r] = snap.E[r]; }
                st.tup_prev = snap.tup_prev; st.tbot = snap.tbot; st.fbot = snap.fbot;
                s2 cmax2[R];
#pragma unroll
                for (int r = 0; r < R; r++) cmax2[r] = splat(-32768);
                if (emit_any) run_chunk16<R, true, true, true, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, emit_lane, emit_row, Z, bias, cmax2, NoService16(), !ragged);
                else if (masked) run_chunk16<R, true, true, false, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax2, NoService16(), !ragged);
                else run_chunk16<R, false, true, false, false, false, SWF>(st, lds, lane, jl, n, nvalid_lo, nvalid_hi, 63, R - 1, Z, bias, cmax2, NoService16(), !ragged);
                exact_mode = (PK16_EXACT_MODE != 0);
            }
        }
        }
        if (trc) q3 = __builtin_amdgcn_s_memrealtime();
        DBG16(3, 30);
        // chunk maximum: range guard, publication of a new global best, re-centring of the 16-bit window
        {
            const int cmv = chunk_max16<R, HALF>(cmax);    // exact, or a lower bound within SLACK of it
            if (cmv + SLACK > GUARD16) overflow = true;
            lane_max = max(lane_max, cmv + bias);
            // the wave-wide maximum (six cross-lane steps) is only needed when one of its three consumers
            // can fire; three ballots decide that
            const bool need_wmax = __any(cmv + SLACK > REBASE_HI) || (TRACK && __any(cmv + bias > gseen)) ||
                                   (SWF ? (bias > 0 && !__any(cmv >= 0)) : !__any(cmv >= REBASE_LO));
            if (need_wmax) {
                int w = cmv;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) w = max(w, __shfl_xor(w, d));
                const int wmax = __builtin_amdgcn_readfirstlane(w);
                if (TRACK && wmax + bias > gseen) {
                    if (lane == 0) atomicMax(a->gbest, wmax + bias);
                }
                int nb = bias;
                if (wmax + SLACK > REBASE_HI) nb = bias + (wmax - REBASE_TO);
                else if (SWF && wmax < 0 && bias > 0 && wmax > -32768) nb = max(0, bias + max(wmax - REBASE_TO, -30000));
                else if (!SWF && wmax < REBASE_LO && wmax > -32768) nb = bias + max(wmax - REBASE_TO, -30000);
                if (nb != bias) {
                    // shift every live 16-bit value by the same amount (saturating: the -INF image stays put
                    // when the window moves up); outputs of this chunk were produced with the old bias
                    rebias_to = nb;
                }
            }
        }
        // (3) output chunk: columns col0-emit_v .. col0-emit_v+63
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int2 tf = lds->out_tf[lane];
            const int col = col0 - emit_v + lane;
            if (col >= 0 && col < n) {
                const int t16 = emit_half ? hi16(tf.x) : lo16(tf.x);
                const int f16 = emit_half ? hi16(tf.y) : lo16(tf.y);
                // -32768 is the sticky image of -INF (only border rows can carry it)
                const int2 hf = make_int2(t16 + T_OFF + bias - 2 * emit_row,
                                          f16 == -32768 ? NEG_INF : f16 + bias - 2 * emit_row);
                st_agent2_16(&a->bus[col], hf);
                if (special != nullptr) special[col] = hf;
                if (lastrow != nullptr) lastrow[col] = hf;
                if (ckpt != nullptr) ckpt[col] = hf;
            }
            // the progress flag follows at the next mid-chunk service (or after the last chunk), when the
            // stores have drained without anybody waiting for them
            int done = col0 - emit_v + CHUNK;
            if (done > n) done = n;
            if (done < 0) done = 0;
            flag_pending = done;
        }
        if (rebias_to != bias) {
            const s2 dlt = splat(rebias_to - bias);
#pragma unroll
            for (int r = 0; r < R; r++) {
                st.TL[r] = __builtin_elementwise_sub_sat(st.TL[r], dlt);
                st.E[r] = __builtin_elementwise_sub_sat(st.E[r], dlt);
            }
            st.tup_prev = __builtin_elementwise_sub_sat(st.tup_prev, dlt);
            st.tbot = __builtin_elementwise_sub_sat(st.tbot, dlt);
            st.fbot = __builtin_elementwise_sub_sat(st.fbot, dlt);
            bias = rebias_to;
#pragma unroll
            for (int r = 0; r < R; r++) Z[r] = splat(clamp16(2 * r - bias));
        }
        if (trc && lane == 0) {
            q4 = __builtin_amdgcn_s_memrealtime();
            long long* tr = a->trace + 4 * s;
            const long long v = ((q1 - q0) & 0xffff) | (((q2 - q1) & 0xffff) << 16) | (((q3 - q2) & 0xffff) << 32) | (((q4 - q3) & 0xffff) << 48);
#ifdef PK16_TRACE_ABS   // absolute times (after the input wait) of chunk 1 and chunk 1000 instead of the phase split
            if (c == 1) tr[2] = q1; else tr[3] = q1;
#else
            if (c == 1) tr[2] = v; else tr[3] = v;
#endif
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (flag_pending >= 0 && lane == 0) st_agent16(prog_out, flag_pending);
    __builtin_amdgcn_s_setprio(0);
    if (__any(overflow)) {
        if (lane == 0) { atomicExch(a->error_flag, 16); st_agent16(a->abort_flag, 1); }
    }
    if (a->trace != nullptr && lane == 0) {
        a->trace[4 * s + 0] = tr_start; a->trace[4 * s + 1] = __builtin_amdgcn_s_memrealtime();
    }
    DBG16(3, 40);
    // ---- strip epilogue ----
    if (a->last_col != nullptr) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int g = (half ? lrow_hi : lrow_lo) + r;
                if (g < a->m) {
                    const int t16 = half ? hi16(as_i(st.TL[r])) : lo16(as_i(st.TL[r]));
                    const int e16 = half ? hi16(as_i(st.E[r])) : lo16(as_i(st.E[r]));
                    a->last_col[g + 1] = make_int2(t16 + T_OFF + bias - 2 * r, e16 == -32768 ? NEG_INF : e16 + bias - 2 * r);
                }
            }
        }
    }
    if (TRACK) {
        lds->red[lane] = st.best_t;
        lds->red[64 + lane] = (st.best_j >= 0) ? (lrow_lo + st.best_r) : 0x7fffffff;   // LO rows then HI rows are consecutive
        lds->red[128 + lane] = st.best_j;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            int bt = NEG_INF, bi = 0x7fffffff, bj = -1;
            for (int k = 0; k < 64; k++) {
                const int t = lds->red[k], i = lds->red[64 + k], j = lds->red[128 + k];
                if (j >= 0 && (t > bt || (t == bt && (i < bi || (i == bi && j < bj))))) { bt = t; bi = i; bj = j; }
            }
            int4 rec;
            rec.x = (bj >= 0) ? bt + T_OFF : NEG_INF;
            rec.y = bi; rec.z = bj; rec.w = 1;
            a->strip_best[s] = rec;
        }
        __builtin_amdgcn_wave_barrier();
    } else {
        // value-only record: the strip's best score; the exact canonical cell of the winning strip is
        // recomputed afterwards from the nearest checkpoint row by the exact-tracking kernel (runtime.cpp)
        lds->red[lane] = lane_max;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            int bt = NEG_INF;
            for (int k = 0; k < 64; k++) bt = max(bt, lds->red[k]);
            int4 rec;
            rec.x = (bt > NEG_INF) ? bt + T_OFF : NEG_INF;
            rec.y = -1; rec.z = -1; rec.w = 2;
            a->strip_best[s] = rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// see sw_kernel.hip: claim/complete are separate noinline functions so that the persistent loop body
// contains no lane-divergent statement for the structurizer to peel.
__device__ __attribute__((noinline)) void complete_strip16(const KernelArgs* ap, const int s_in, const int lane) {
    const UniformArgs a = uniform_args(ap);
    const int s = __builtin_amdgcn_readfirstlane(s_in);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    int spins = 0;
    while (poll_agent16(a->strips_done_dev) != s && spins < (1 << 24)) {
        __builtin_amdgcn_s_sleep(8);
        spins++;
    }
    if (lane == 0) {
        if (spins >= (1 << 24)) atomicExch(a->error_flag, 3);
        if (a->strips_done_host != nullptr)
            __hip_atomic_store(a->strips_done_host, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        st_agent16(a->strips_done_dev, s + 1);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __attribute__((noinline)) int claim_strip16(const KernelArgs* ap, const int lane) {
    const UniformArgs a = uniform_args(ap);
    int s = 0;
    if (lane == 0) s = atomicAdd(a->ticket, 1);
    return __builtin_amdgcn_readfirstlane(s);
}

template <int R, bool TRACK, bool SWF>
__global__ void __launch_bounds__(64) sw_strip_kernel_pk16(const KernelArgs* __restrict__ ap) {
    __shared__ WaveLds16 lds_store;
    WaveLds16* lds = &lds_store;
    const int lane = threadIdx.x;
    const UniformArgs a = uniform_args(ap);
    const int num_strips = a->num_strips;
    // One wavefront per SIMD is the design point and the strip chain runs at the speed of its
    // slowest member, so placement must not be left to the dispatcher: claiming the top accumulation
    // register makes the wavefront's register allocation exceed half of the SIMD's 512-entry file, and the
    // hardware then cannot co-schedule two of them on one SIMD.
    asm volatile("" ::: "a255");
    for (;;) {
        const int s = __builtin_amdgcn_readfirstlane(claim_strip16(ap, lane));
        if (s >= num_strips) break;
        if (poll_agent16(a->abort_flag) != 0) {
            if (lane == 0) st_agent16(&a->progress[s + 1], a->n);
            __builtin_amdgcn_wave_barrier();
        } else {
            process_strip16<R, TRACK, SWF>(ap, s, lds, lane);
        }
        complete_strip16(ap, s, lane);
    }
}

hipError_t launch_strip_kernel_pk16(const KernelArgs& a, KernelArgs* dargs, int rows_per_half, int grid, hipStream_t stream, bool track, bool sw) {
    hipError_t e = hipMemcpyAsync(dargs, &a, sizeof(KernelArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
#define LAUNCH16(RV, TRV, SWV) hipLaunchKernelGGL((sw_strip_kernel_pk16<RV, TRV, SWV>), dim3(grid), dim3(64), 0, stream, (const KernelArgs*) dargs)
#define LAUNCH16R(RV) do { if (sw) { if (track) LAUNCH16(RV, true, true); else LAUNCH16(RV, false, true); } \
                           else    { if (track) LAUNCH16(RV, true, false); else LAUNCH16(RV, false, false); } } while (0)
    switch (rows_per_half) {
    case 2: LAUNCH16R(2); break;
    case 4: LAUNCH16R(4); break;
    case 6: LAUNCH16R(6); break;
    case 8: LAUNCH16R(8); break;
    case 12: LAUNCH16R(12); break;
    case 16: LAUNCH16R(16); break;
    default: return hipErrorInvalidValue;
    }
#undef LAUNCH16R
#undef LAUNCH16
    return hipGetLastError();
}

}  // namespace mi355sw
