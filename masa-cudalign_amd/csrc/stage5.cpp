// Stage 5 on the host: the exact alignment of every partition between two consecutive crosspoints of crosspoint_04
// (at most 16 x 16 after stage 4), by a full-matrix Gotoh pass and a traceback that honours the crosspoint types.
// Replaces M/stage5/sw_stage5.cpp (sw() :83-319, the loop of stage5() :322-485) -- single-threaded CPU code in the
// reference too; a 10 M-column alignment is ~700 k partitions of <= 256 cells, milliseconds here.
// The caller (masa-cudalign_amd/stage56.py) turns the gap events into Alignment.cpp's gap lists.
// Second half of this file: stage 6's text (mi355sw_stage6_text) -- host code of the same kind, kept in the same
// translation unit so that the build recipe, which is part of the kernels' build identity, stays as it is.
#include "../../include/mi355sw.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

enum { GAP_OPEN = 3, GAP_EXT = 2, MATCH = 1, MISMATCH = -3, GAP_FIRST = GAP_OPEN + GAP_EXT, NINF = -999999999 };
enum { TYPE_MATCH = 0, TYPE_GAP_1 = 1, TYPE_GAP_2 = 2 };

struct Out {
    std::vector<int32_t> g0, g1;     // gap events: DP row i of a gap in sequence 0's list, DP column j of one in sequence 1's
    mi355sw_stage5_totals tot;
};

// _dot (sw_stage5.cpp:64-80): type 1 = the path moves along S0 only -> a gap character in S1 at column j; 2 = the reverse
inline void dot(Out& o, int i, int j, int typ) {
    if (typ == 1) o.g1.push_back(j);
    else if (typ == 2) o.g0.push_back(i);
}

// sw() (:83-319): rows (i0, i1], columns (j0, j1]; returns 0, or -1 when the traceback finds no predecessor
int sw(Out& o, const unsigned char* d0, const unsigned char* d1, int i0, int j0, int i1, int j1, int type_s, int type_e,
       std::vector<int>& H, std::vector<int>& E, std::vector<int>& F) {
    mi355sw_stage5_totals& t = o.tot;
    if (i0 == i1) {
        long long s = (long long) (j1 - j0) * -GAP_EXT;
        if (type_s != TYPE_GAP_1) { t.gap_open++; s += -GAP_OPEN; }
        for (int j = j1; j > j0; j--) { dot(o, i0, j, 2); t.gap_extensions++; }
        t.score += s;
        return 0;
    }
    if (j0 == j1) {
        long long s = (long long) (i1 - i0) * -GAP_EXT;
        if (type_s != TYPE_GAP_2) { t.gap_open++; s += -GAP_OPEN; }
        for (int i = i1; i > i0; i--) { dot(o, i, j0, 1); t.gap_extensions++; }
        t.score += s;
        return 0;
    }
    const int rows = i1 - i0, cols = j1 - j0, W = cols + 1;
    const unsigned char* a = d0 + i0;
    const unsigned char* b = d1 + j0;
    H.assign((size_t) (rows + 1) * W, 0);
    E.assign((size_t) (rows + 1) * W, NINF);
    F.assign((size_t) (rows + 1) * W, NINF);
    for (int j = 1; j <= cols; j++) H[j] = -j * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_1);
    H[0] = type_s != 0 ? NINF : 0;
    for (int i = 1; i <= rows; i++) {
        int* hi = &H[(size_t) i * W]; const int* hp = hi - W;
        int* ei = &E[(size_t) i * W]; const int* ep = ei - W;
        int* fi = &F[(size_t) i * W];
        hi[0] = -i * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_2);
        const unsigned char s = a[i - 1];
        for (int j = 1; j <= cols; j++) {
            const int ev = std::max(hp[j] - GAP_FIRST, ep[j] - GAP_EXT);
            const int fv = std::max(hi[j - 1] - GAP_FIRST, fi[j - 1] - GAP_EXT);
            ei[j] = ev; fi[j] = fv;
            hi[j] = std::max(std::max(hp[j - 1] + (s == b[j - 1] ? MATCH : MISMATCH), ev), fv);
        }
    }
    int i = rows, j = cols;
    int c = type_e;                       // 0 aligned, TYPE_GAP_2, TYPE_GAP_1
    long long total = 0;
    while (i > 0 && j > 0) {
        const int eh = H[(size_t) (i - 1) * W + j] - GAP_FIRST;
        const int fh = H[(size_t) i * W + j - 1] - GAP_FIRST;
        const int h11 = H[(size_t) (i - 1) * W + j - 1] + (a[i - 1] == b[j - 1] ? MATCH : MISMATCH);
        const int h10 = E[(size_t) i * W + j], h01 = F[(size_t) i * W + j], h00 = H[(size_t) i * W + j];
        int d;
        if (c == 0) {
            if (h00 == h11) { d = 0; c = TYPE_MATCH; }
            else if (h00 == h10) { d = 1; c = (h10 == eh) ? TYPE_MATCH : TYPE_GAP_2; }
            else if (h00 == h01) { d = 2; c = (h01 == fh) ? TYPE_MATCH : TYPE_GAP_1; }
            else return -1;
        } else if (c == TYPE_GAP_2) { d = 1; c = (h10 == eh) ? TYPE_MATCH : TYPE_GAP_2; }
        else { d = 2; c = (h01 == fh) ? TYPE_MATCH : TYPE_GAP_1; }
        dot(o, i0 + i, j0 + j, d);
        if (d == 0) {
            if (a[i - 1] == b[j - 1]) { t.matches++; total += MATCH; }
            else { t.mismatches++; total += MISMATCH; }
            i--; j--;
        } else {
            if (c == TYPE_MATCH) { t.gap_open++; total += -GAP_FIRST; }
            else total += -GAP_EXT;
            t.gap_extensions++;
            if (d == 1) i--; else j--;
        }
    }
    while (i > 0) { dot(o, i0 + i, j0 + j, 1); i--; t.gap_extensions++; c = TYPE_GAP_2; total += -GAP_EXT; }
    while (j > 0) { dot(o, i0 + i, j0 + j, 2); j--; t.gap_extensions++; c = TYPE_GAP_1; total += -GAP_EXT; }
    if (type_s == TYPE_MATCH && c != TYPE_MATCH) total -= GAP_OPEN;
    t.score += total;
    return 0;
}

int32_t* give(const std::vector<int32_t>& v) {
    int32_t* p = (int32_t*) malloc(sizeof(int32_t) * (v.size() ? v.size() : 1));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(int32_t) * v.size());
    return p;
}

}  // namespace

extern "C" int mi355sw_stage5(const char* seq0, int32_t len0, const char* seq1, int32_t len1, const mi355sw_crosspoint* cps,
                              int32_t count, int32_t** gaps0, int64_t* n0, int32_t** gaps1, int64_t* n1,
                              mi355sw_stage5_totals* totals, int32_t* failed_at) {
    if (!seq0 || !seq1 || !cps || count < 1 || !gaps0 || !n0 || !gaps1 || !n1 || !totals) return MI355SW_EINVAL;
    Out o;
    memset(&o.tot, 0, sizeof(o.tot));
    std::vector<int> H, E, F;
    for (int k = 0; k + 1 < count; k++) {
        const mi355sw_crosspoint &p = cps[k], &q = cps[k + 1];
        if (p.i < 0 || p.j < 0 || q.i > len0 || q.j > len1 || q.i < p.i || q.j < p.j) { if (failed_at) *failed_at = k; return MI355SW_EINVAL; }
        // W_MAX of the reference (sw_stage5.cpp:40): stage 4 must have run first
        if (q.i != p.i && q.j != p.j && (q.i - p.i > 8192 || q.j - p.j > 8192)) { if (failed_at) *failed_at = k; return MI355SW_ETOOLARGE; }
        if (sw(o, (const unsigned char*) seq0, (const unsigned char*) seq1, p.i, p.j, q.i, q.j, p.type, q.type, H, E, F) != 0) {
            if (failed_at) *failed_at = k;
            return MI355SW_ETRACEBACK;
        }
    }
    *gaps0 = give(o.g0); *n0 = (int64_t) o.g0.size();
    *gaps1 = give(o.g1); *n1 = (int64_t) o.g1.size();
    if (!*gaps0 || !*gaps1) { free(*gaps0); free(*gaps1); return MI355SW_ENOMEM; }
    *totals = o.tot;
    return MI355SW_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stage 6 on the host: the body of alignment.NN.txt -- 60-column blocks of Query / match marks / Sbjct with positions and
// running scores, then the summary -- from the forward sequence data and the alignment's two gap lists.  Replaces
// printText of M/stage6/sw_stage6.cpp:60-262 from the first block on (the caller writes the three header lines, which
// need the sequences' descriptions and trim positions).  Plain host code like the reference's; 200 MB of text for a
// 46 M-column alignment is a second here.
#include <cstdio>
#include <string>

namespace stage6 {

enum { COLS = 60 };      // (the scores are stage 5's, above)

// one sequence's side of a block (:96-132 / :134-170): walks from *pos towards `last` in direction `dir`, emitting a
// '-' wherever the gap list says so (an entry (p, len) sits before position p walking forwards, after it backwards)
struct Walker {
    const unsigned char* d;
    const int32_t* gaps;     // (position, length) pairs
    int n_gaps, dir, pos, last, cur, left_pos, left_len;
    bool done;
    void load() {
        if (cur >= 0 && cur < n_gaps) { left_pos = gaps[2 * cur]; left_len = gaps[2 * cur + 1]; }
        else { left_pos = -1; left_len = -1; }
    }
    int fill(unsigned char* out) {
        int k = 0, len = 0;
        while (k < COLS && !done) {
            if (left_pos == pos + (dir > 0 ? 0 : 1)) {
                out[len++] = '-';
                if (--left_len == 0) { cur += dir; load(); }
            } else {
                out[len++] = d[pos - 1];
                if (pos == last) { done = true; break; }
                pos += dir;
            }
            k++;
        }
        return len;
    }
};
}  // namespace stage6

extern "C" int mi355sw_stage6_text(const char* seq0, int32_t seq0_len, const char* seq1, int32_t seq1_len, int32_t i0, int32_t j0,
                                   int32_t i1, int32_t j1, const int32_t* gaps0, int32_t n_gaps0, const int32_t* gaps1,
                                   int32_t n_gaps1, int64_t raw_score, char** text, int64_t* text_len,
                                   mi355sw_stage5_totals* totals) {
    if (!seq0 || !seq1 || !text || !text_len || (n_gaps0 > 0 && !gaps0) || (n_gaps1 > 0 && !gaps1)) return MI355SW_EINVAL;
    *text = nullptr; *text_len = 0;
    const bool none = (i0 == -1 && j0 == -1 && i1 == -1 && j1 == -1);
    if (!none && (i0 < 1 || i0 > seq0_len || i1 < 1 || i1 > seq0_len || j0 < 1 || j0 > seq1_len || j1 < 1 || j1 > seq1_len))
        return MI355SW_EINVAL;
    stage6::Walker q = {(const unsigned char*) seq0, gaps0, n_gaps0, i1 > i0 ? 1 : -1, i0, i1, 0, -1, -1, none};
    stage6::Walker s = {(const unsigned char*) seq1, gaps1, n_gaps1, j1 > j0 ? 1 : -1, j0, j1, 0, -1, -1, none};
    q.cur = q.dir > 0 ? 0 : n_gaps0 - 1;
    s.cur = s.dir > 0 ? 0 : n_gaps1 - 1;
    q.load(); s.load();
    std::string out;
    out.reserve((size_t) (none ? 256 : ((long long) (abs(i1 - i0) + abs(j1 - j0)) / 60 + 2) * 230));
    long long score = 0, gap_openings = 0, gap_extentions = 0, matches = 0, mismatches = 0;
    int qgap = 0, sgap = 0;
    if (none) out += "There was no alignment produced!\n\n";
    using namespace stage6;
    unsigned char qb[COLS + 1], sb[COLS + 1], marks[COLS + 1];
    char line[64];
    while (!q.done || !s.done) {
        const int qp = q.pos, sp = s.pos;
        int ql = q.fill(qb), sl = s.fill(sb);
        while (sl < ql) sb[sl++] = '-';
        while (ql < sl) qb[ql++] = '-';
        snprintf(line, sizeof line, "Query: %8d ", qp);
        out += line;
        out.append((const char*) qb, ql);
        snprintf(line, sizeof line, " %8d\n", q.pos);
        out += line;
        out += "                ";
        long long temp = 0;
        for (int k = 0; k < ql; k++) {
            const unsigned char a = qb[k], b = sb[k];
            marks[k] = (a == b) ? '|' : ' ';
            if (a == '-') {
                if (qgap) temp += -GAP_EXT; else { temp += -GAP_OPEN - GAP_EXT; gap_openings++; }
                gap_extentions++;
                qgap = 1; sgap = 0;
            } else if (b == '-') {
                if (sgap) temp += -GAP_EXT; else { temp += -GAP_OPEN - GAP_EXT; gap_openings++; }
                gap_extentions++;
                qgap = 0; sgap = 1;
            } else {
                if (a == b) { temp += MATCH; matches++; } else { temp += MISMATCH; mismatches++; }
                qgap = sgap = 0;
            }
        }
        score += temp;
        out.append((const char*) marks, ql);
        snprintf(line, sizeof line, " [%lld/%lld]\n", temp, score);
        out += line;
        snprintf(line, sizeof line, "Sbjct: %8d ", sp);
        out += line;
        out.append((const char*) sb, sl);
        snprintf(line, sizeof line, " %8d\n", s.pos);
        out += line;
        out += "\n\n";
    }
    if (totals) {
        totals->score = score; totals->matches = matches; totals->mismatches = mismatches;
        totals->gap_open = gap_openings; totals->gap_extensions = gap_extentions;
    }
    if (score != raw_score) return MI355SW_ETRACEBACK;      // "Stage6 error: Alignment score is different" (:243-247)
    out += "Summary:\n\n";
    snprintf(line, sizeof line, "Total Score:    %10lld\n", score); out += line;
    snprintf(line, sizeof line, "Matches:        %10lld (+%d)\n", matches, MATCH); out += line;
    snprintf(line, sizeof line, "Mismatches:     %10lld (%d)\n", mismatches, MISMATCH); out += line;
    snprintf(line, sizeof line, "Gap Openings:   %10lld (%d)\n", gap_openings, -GAP_OPEN); out += line;
    snprintf(line, sizeof line, "Gap Extentions: %10lld (%d)\n", gap_extentions, -GAP_EXT); out += line;
    char* buf = (char*) malloc(out.size() + 1);
    if (!buf) return MI355SW_ENOMEM;
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *text = buf;
    *text_len = (int64_t) out.size();
    return MI355SW_OK;
}

// ---- crosspoint files as text (CrosspointsFile::save, M/common/CrosspointsFile.cpp:99-160) ----------------------------------
// "START\n", one "type,i,j,score\n" line per crosspoint, "END\n": crosspoint_04 of a 46 M-column alignment is 4.6 M lines
// (130 MB), 2-4 s of formatting in the Python host and a tenth of a second here.
extern "C" int mi355sw_crosspoints_text(const mi355sw_crosspoint* points, int64_t count, char** text, int64_t* text_len) {
    if (!text || !text_len || count < 0 || (count > 0 && !points)) return MI355SW_EINVAL;
    *text = nullptr; *text_len = 0;
    // at most 4 numbers of 11 characters, 3 commas and a newline per line
    char* buf = (char*) malloc((size_t) count * 48 + 16);
    if (!buf) return MI355SW_ENOMEM;
    char* w = buf;
    memcpy(w, "START\n", 6); w += 6;
    auto put = [&](long long v) {
        char tmp[24];
        int n = 0;
        unsigned long long u = v < 0 ? (unsigned long long) (-v) : (unsigned long long) v;
        do { tmp[n++] = (char) ('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *w++ = '-';
        while (n) *w++ = tmp[--n];
    };
    for (int64_t k = 0; k < count; k++) {
        put(points[k].type); *w++ = ',';
        put(points[k].i); *w++ = ',';
        put(points[k].j); *w++ = ',';
        put(points[k].score); *w++ = '\n';
    }
    memcpy(w, "END\n", 4); w += 4;
    *text = buf;
    *text_len = (int64_t) (w - buf);
    return MI355SW_OK;
}
