// MI355X stage 4: the Myers-Miller refinement of the stage-3 crosspoints, batched on the GPU.
//
// Replaces MASA-Core's CPU stage 4 (M/stage4/sw_stage4.cpp: stage4() :880-960, reduce_partitions :809-860 with four
// pthreads, split_thread :86-222, ort_split_2 :293-380, processCol :250-271, match :273-291, merge_partitions
// :786-807) -- after stage 1 the longest stage of the reference's pipeline on the engine (242 s of the 612 s of the
// 48 M x 46 M run).  The arithmetic and every tie-break are the reference's; what changes is the schedule:
//
//   * one ITERATION of the reference (every partition larger than the limit is cut once, in the middle of its longer
//     side) is two launches: `mm_half_kernel` computes, for every partition, the middle row of the forward half-matrix
//     and of the reverse half-matrix -- one wavefront per half, thousands of halves resident together; then
//     `mm_match_kernel` walks the candidate columns of every partition in the reference's order (from the middle
//     outwards, forward side first, aligned before gapped) and takes the first one whose forward + reverse scores
//     add up to the partition's score difference.
//   * a half-matrix is swept by ONE wavefront as a systolic array: lane k owns 4 rows, lane k+1 is one column
//     behind lane k and receives its bottom (H,F) by DPP; 256 rows per pass, the last row of a pass is the bus row
//     of the next one (in place in the output array, 64 columns prefetched per chunk).  No inter-wavefront
//     synchronisation at all: the batch is embarrassingly parallel, which a single pair of sequences never is.
//
// The reference evaluates the two halves column by column and stops at the first matching column; computing both
// middle rows completely and then choosing by the same order gives the same crosspoint (the DP values do not depend
// on the order they are computed in).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "sw_kernel.h"

namespace mi355sw {

#define S4_INF 999999999
#define S4_GAP_OPEN 3
#define S4_GAP_EXT 2
#define S4_GAP_FIRST 5
#define S4_RB 4                 // rows per lane
#define S4_PASS (64 * S4_RB)    // rows per pass

struct HalfProblem {
    long long a_off, b_off;     // element of row r / column c: A[a_off + r*a_stride], B[b_off + c*b_stride]
    int a_stride, b_stride;
    int a_is_seq1;              // which comparable array is A (0: seq0, 1: seq1); B is the other one
    int rows, cols;
    int col_open, row_open;     // first column: -(r+1)*EXT - col_open ; top border: -(c+1)*EXT - row_open
    int corner;                 // H of the corner: 0 or -INF
    long long out_off;          // (cols+1) cells (H,F) of the last row; cell 0 = first-column value
};

struct MatchProblem {
    long long fwd_off, rev_off; // output arrays of the two halves
    int lenB, diff;
    int result_type, result_col, result_score, status;   // status: 0 found, -3 not found, -4 sum above the difference
};

__device__ __forceinline__ int shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }

// cmp0[i] = seq0[i] << shift (coded sequences keep seq1 as code*4); afterwards equal bytes <=> matching residues
__global__ void s4_make_comparable(const unsigned char* in, unsigned char* out, long long n, int shift) {
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) out[k] = (unsigned char) (in[k] << shift);
}

__global__ void __launch_bounds__(64) mm_half_kernel(const HalfProblem* __restrict__ problems, const unsigned char* __restrict__ cmp0,
                                                    const unsigned char* __restrict__ cmp1, int2* __restrict__ cells) {
    const HalfProblem P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    const unsigned char* A = P.a_is_seq1 ? cmp1 : cmp0;
    const unsigned char* B = P.a_is_seq1 ? cmp0 : cmp1;
    int2* out = cells + P.out_off;
    const int rows = P.rows, cols = P.cols;
    if (lane == 0) {
        const int v = -rows * S4_GAP_EXT - P.col_open;      // r0[0].h = r0[0].f = c0[imid0-1].h
        out[0] = make_int2(v, v);
    }
    const int npass = (rows + S4_PASS - 1) / S4_PASS;
    for (int p = 0; p < npass; p++) {
        const int row0 = p * S4_PASS + lane * S4_RB;          // first row (0-based) of this lane
        int nv = rows - row0; nv = nv < 0 ? 0 : (nv > S4_RB ? S4_RB : nv);
        const int last = min(rows, (p + 1) * S4_PASS) - p * S4_PASS - 1;   // last row of the pass, pass-relative
        const int k_last = last / S4_RB;
        int hl[S4_RB], el[S4_RB], a[S4_RB];
#pragma unroll
        for (int r = 0; r < S4_RB; r++) {
            hl[r] = -(row0 + r + 1) * S4_GAP_EXT - P.col_open;           // column 0 of the half-matrix
            el[r] = -S4_INF;
            a[r] = (r < nv) ? (int) A[P.a_off + (long long) (row0 + r) * P.a_stride] : 0x100;
        }
        // H of (row above this lane, column 0): the corner for the very first row, else the first-column value
        int up_prev = (row0 == 0) ? P.corner : (-row0 * S4_GAP_EXT - P.col_open);
        int bot_h = -S4_INF, bot_f = -S4_INF;
        int2 bus_chunk = make_int2(0, 0);
        int bcur = 0x200, bprev = 0x200;
        const int steps = cols + k_last;
        for (int t = 0; t < steps; t++) {
            const int t0 = t & ~63;
            if ((t & 63) == 0) {
                // 64 columns of the row above (previous pass, in place in `out`) and of sequence B
                const int c = t0 + lane;
                bprev = bcur;
                bcur = (c < cols) ? (int) B[P.b_off + (long long) c * P.b_stride] : 0x200;
                if (p > 0 && c < cols) {
                    // written by another lane of this wavefront during the previous pass: read past the L1
                    const unsigned long long x = __hip_atomic_load((const unsigned long long*) &out[c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bus_chunk = make_int2((int) (unsigned) x, (int) (unsigned) (x >> 32));
                }
            }
            // hand-off from the lane above: (H,F) of its bottom row at the column this lane works on now
            int feed_h, feed_f;
            if (p == 0) {
                feed_h = -(t + 1) * S4_GAP_EXT - P.row_open;           // top border, lane 0 is at column t
                feed_f = -S4_INF;
            } else {
                feed_h = __shfl(bus_chunk.x, t - t0);
                feed_f = __shfl(bus_chunk.y, t - t0);
            }
            int up_h = shr1(feed_h, bot_h);
            int up_f = shr1(feed_f, bot_f);
            const int c = t - lane;                                        // 0-based column of this lane
            const int bidx = c - t0;                                       // >= -63
            const int b_now = __shfl(bcur, bidx & 63);
            const int b_old = __shfl(bprev, bidx & 63);
            const int b = bidx >= 0 ? b_now : b_old;
            if (c >= 0 && c < cols && nv > 0) {
                int diag = up_prev;
                const int up_keep = up_h;
                int oh = 0, of = 0;
#pragma unroll
                for (int r = 0; r < S4_RB; r++) {
                    const int e = max(hl[r] - S4_GAP_FIRST, el[r] - S4_GAP_EXT);
                    const int f = max(up_h - S4_GAP_FIRST, up_f - S4_GAP_EXT);
                    const int h = max(max(diag + ((a[r] == b) ? 1 : -3), e), f);
                    diag = hl[r];
                    if (r < nv) { hl[r] = h; el[r] = e; up_h = h; up_f = f; oh = h; of = f; }
                }
                up_prev = up_keep;
                bot_h = oh; bot_f = of;
                if (lane == k_last) out[c + 1] = make_int2(oh, of);
            }
        }
        // the pass's last row is the next pass's bus row: its stores must have landed before they are read again
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
}

// sw_stage4.cpp:351-372: candidate columns from the middle outwards; (a) forward cell of column jmid1+q against the
// reverse cell of the same column, then (b) the mirrored column; match(): aligned first, then gapped.
__global__ void __launch_bounds__(64) mm_match_kernel(MatchProblem* __restrict__ problems, const int2* __restrict__ cells) {
    MatchProblem& P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    const int2* fwd = cells + P.fwd_off;
    const int2* rev = cells + P.rev_off;
    const int lenB = P.lenB, diff = P.diff;
    const int jmid1 = lenB - lenB / 2;
    const int ncand = 2 * (lenB - jmid1 + 1);
    for (int q0 = 0; q0 < ncand; q0 += 64) {
        const int q = q0 + lane;
        int found = 0, err = 0, type = 0, score = 0, col = 0;
        if (q < ncand) {
            const int jp1 = jmid1 + (q >> 1);                  // j+1 of the reference's loop
            col = (q & 1) ? lenB - jp1 : jp1;
            const int2 a = fwd[col];
            const int2 b = rev[lenB - col];
            const int sum_match = a.x + b.x;
            const int sum_gap = a.y + b.y + S4_GAP_OPEN;
            if (sum_match == diff) { found = 1; type = 0; score = a.x; }
            else if (sum_gap == diff) { found = 1; type = 2; score = a.y; }
            else if (sum_match > diff || sum_gap > diff) err = 1;
        }
        const unsigned long long fm = __ballot(found), em = __ballot(err);
        const int ff = fm ? __builtin_ctzll(fm) : 64, fe = em ? __builtin_ctzll(em) : 64;
        if (fe < ff) { if (lane == 0) P.status = -4; return; }
        if (ff < 64) {
            if (lane == ff) { P.result_type = type; P.result_col = col; P.result_score = score; P.status = 0; }
            return;
        }
    }
    if (lane == 0) P.status = -3;
}

namespace {
struct Buf { void* p = nullptr; size_t cap = 0; };
hipError_t ensure_buf(Buf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return hipSuccess;
    if (b.p) (void) hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
    hipError_t e = hipMalloc(&b.p, want);
    if (e == hipSuccess) b.cap = want;
    return e;
}
int largest_partition(const std::vector<Stage4Crosspoint>& cp) {      // CrosspointsFile::getLargestPartitionSize
    int mi = 0, mj = 0;
    for (size_t k = 1; k < cp.size(); k++) {
        const int di = std::abs(cp[k - 1].i - cp[k].i), dj = std::abs(cp[k - 1].j - cp[k].j);
        if (di != 0 && dj != 0) { mi = std::max(mi, di); mj = std::max(mj, dj); }
    }
    return std::max(mi, mj);
}
}  // namespace

// Returns 0, or a negative code: -1 HIP error (*hip_err), -2 a partition exceeds the reference's H_MAX, -3 a partition
// without a matching column ("NOT FOUND" in the reference), -4 a column whose scores exceed the difference ("Error Match").
int stage4_refine(const unsigned char* d_seq0, long long len0, const unsigned char* d_seq1, long long len1, int seq0_shift,
                  hipStream_t stream, std::vector<Stage4Crosspoint>& list, int max_size, Stage4Stats* stats, hipError_t* hip_err) {
    static const int inv_type[] = {0, 2, 1};
    const int H_MAX = 2 * 64 * 1024;
    Buf cmp0, cmp1, d_half, d_match, d_cells;
    hipError_t e = hipSuccess;
    int rc = 0;
#define S4CHK(x) do { e = (x); if (e != hipSuccess) { rc = -1; goto done; } } while (0)
    {
    S4CHK(ensure_buf(cmp0, (size_t) len0 + 64));
    S4CHK(ensure_buf(cmp1, (size_t) len1 + 64));
    hipLaunchKernelGGL(s4_make_comparable, dim3(1024), dim3(256), 0, stream, d_seq0, (unsigned char*) cmp0.p, len0, seq0_shift);
    hipLaunchKernelGGL(s4_make_comparable, dim3(1024), dim3(256), 0, stream, d_seq1, (unsigned char*) cmp1.p, len1, 0);
    S4CHK(hipGetLastError());
    hipEvent_t ev0, ev1;
    S4CHK(hipEventCreate(&ev0)); S4CHK(hipEventCreate(&ev1));
    while (largest_partition(list) > max_size) {
        std::vector<HalfProblem> halves;
        std::vector<MatchProblem> matches;
        std::vector<int> owner;                          // partition index k of every match problem
        std::vector<char> inverse;
        long long cells = 0, dp_cells = 0;
        for (size_t k = 1; k < list.size(); k++) {
            const Stage4Crosspoint &s = list[k - 1], &t = list[k];
            const int di = t.i - s.i, dj = t.j - s.j;
            if (di == 0 || dj == 0) continue;
            const bool inv = di < dj;
            // split_thread :121-203: the longer side is cut; partitions within the limit are left alone
            if (inv ? !(s.j < t.j - max_size) : !(s.i < t.i - max_size)) continue;
            const int i0 = inv ? s.j : s.i, i1 = inv ? t.j : t.i, j0 = inv ? s.i : s.j, j1 = inv ? t.i : t.j;
            const int type_s = inv ? inv_type[s.type] : s.type, type_e = inv ? inv_type[t.type] : t.type;
            const int lenA = i1 - i0, lenB = j1 - j0;
            if (lenB >= H_MAX || lenA / 2 + 1 >= H_MAX) { rc = -2; goto done_ev; }
            const int imid0 = lenA / 2, imid1 = lenA - imid0;
            HalfProblem f{}, r{};
            f.a_is_seq1 = r.a_is_seq1 = inv ? 1 : 0;
            f.a_off = i0; f.a_stride = 1; f.b_off = j0; f.b_stride = 1;
            f.rows = imid0; f.cols = lenB;
            f.col_open = S4_GAP_OPEN * (type_s != 2); f.row_open = S4_GAP_OPEN * (type_s != 1);
            f.corner = (type_s != 0) ? -S4_INF : 0;
            f.out_off = cells; cells += lenB + 1;
            r.a_off = (long long) i1 - 1; r.a_stride = -1; r.b_off = (long long) j1 - 1; r.b_stride = -1;
            r.rows = imid1; r.cols = lenB;
            r.col_open = S4_GAP_OPEN; r.row_open = S4_GAP_OPEN;
            r.corner = (type_e != 0) ? -S4_INF : 0;
            r.out_off = cells; cells += lenB + 1;
            MatchProblem mp{};
            mp.fwd_off = f.out_off; mp.rev_off = r.out_off; mp.lenB = lenB; mp.diff = t.score - s.score; mp.status = -3;
            halves.push_back(f); halves.push_back(r);
            matches.push_back(mp);
            owner.push_back((int) k);
            inverse.push_back(inv ? 1 : 0);
            dp_cells += (long long) lenA * lenB;
        }
        if (matches.empty()) break;
        S4CHK(ensure_buf(d_half, halves.size() * sizeof(HalfProblem)));
        S4CHK(ensure_buf(d_match, matches.size() * sizeof(MatchProblem)));
        S4CHK(ensure_buf(d_cells, (size_t) cells * sizeof(int2)));
        S4CHK(hipMemcpyAsync(d_half.p, halves.data(), halves.size() * sizeof(HalfProblem), hipMemcpyHostToDevice, stream));
        S4CHK(hipMemcpyAsync(d_match.p, matches.data(), matches.size() * sizeof(MatchProblem), hipMemcpyHostToDevice, stream));
        S4CHK(hipEventRecord(ev0, stream));
        hipLaunchKernelGGL(mm_half_kernel, dim3((unsigned) halves.size()), dim3(64), 0, stream, (const HalfProblem*) d_half.p,
                           (const unsigned char*) cmp0.p, (const unsigned char*) cmp1.p, (int2*) d_cells.p);
        hipLaunchKernelGGL(mm_match_kernel, dim3((unsigned) matches.size()), dim3(64), 0, stream, (MatchProblem*) d_match.p,
                           (const int2*) d_cells.p);
        S4CHK(hipGetLastError());
        S4CHK(hipEventRecord(ev1, stream));
        S4CHK(hipMemcpyAsync(matches.data(), d_match.p, matches.size() * sizeof(MatchProblem), hipMemcpyDeviceToHost, stream));
        S4CHK(hipStreamSynchronize(stream));
        float ms = 0.f;
        S4CHK(hipEventElapsedTime(&ms, ev0, ev1));
        if (stats) { stats->steps++; stats->kernel_ms += ms; stats->dp_cells += dp_cells; stats->partitions += (long long) matches.size(); }
        // merge_partitions :786-807
        std::vector<Stage4Crosspoint> merged;
        merged.reserve(list.size() + matches.size());
        merged.push_back(list[0]);
        bool has_new = false;
        size_t q = 0;
        for (size_t k = 1; k < list.size(); k++) {
            if (q < owner.size() && owner[q] == (int) k) {
                const MatchProblem& mp = matches[q];
                if (mp.status != 0) { rc = mp.status; goto done_ev; }
                const Stage4Crosspoint &s = list[k - 1], &t = list[k];
                const bool inv = inverse[q] != 0;
                Stage4Crosspoint n;
                const int lenA = inv ? t.j - s.j : t.i - s.i;
                const int mid = (inv ? s.j : s.i) + lenA / 2;              // cross.i = imid0 + i0
                const int col = (inv ? s.i : s.j) + mp.result_col;        // cross.j = j0 + column
                n.type = inv ? inv_type[mp.result_type] : mp.result_type;
                n.i = inv ? col : mid;
                n.j = inv ? mid : col;
                n.score = mp.result_score + s.score;
                if (n.i != s.i || n.j != s.j) { has_new = true; merged.push_back(n); }
                q++;
            }
            merged.push_back(list[k]);
        }
        if (!has_new) break;                               // "Didn't reduce partition."
        list.swap(merged);
    }
done_ev:
    (void) hipEventDestroy(ev0); (void) hipEventDestroy(ev1);
    }
done:
#undef S4CHK
    for (Buf* b : {&cmp0, &cmp1, &d_half, &d_match, &d_cells}) if (b->p) (void) hipFree(b->p);
    if (hip_err) *hip_err = e;
    return rc;
}

}  // namespace mi355sw
