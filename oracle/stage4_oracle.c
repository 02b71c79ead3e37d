/* oracle/stage4_oracle.c -- TEST INFRASTRUCTURE (see oracle/sw_oracle.h): plain-C restatement of MASA-Core's stage 4,
 * the Myers-Miller refinement of the stage-3 crosspoints down to partitions of at most `max_size` rows/columns
 * (M/stage4/sw_stage4.cpp).  Only tests/ and smoke() use it, as the checker of the HIP stage 4.
 *
 * Restated: the default strategy STAGE_4_STRATEGY_OPTIMIZED = ort_split_2 (:293-380) with processCol (:250-271) and
 * match (:273-291); the partition loop of split_thread (:86-222: orientation so that the split dimension is the larger
 * one, inv_type, "no split" markers); merge_partitions (:786-807) and the step loop of stage4() (:923-946) with
 * CrosspointsFile::getLargestPartitionSize (M/common/CrosspointsFile.cpp:71-92).
 * Not restated: the two other strategies and the fall-back to ORIGINAL_MM for half-partitions >= H_MAX (131072): such
 * a partition makes oc_stage4() return -2.  A partition without a matching column ("NOT FOUND", exit(1) in the
 * reference) returns -3.                                                                                           */
#include <stdlib.h>
#include <string.h>

#define OC_INF 999999999
#define GAP_OPEN 3
#define GAP_EXT 2
#define GAP_FIRST (GAP_OPEN + GAP_EXT)
#define SC_MATCH 1
#define SC_MISMATCH (-3)
#define TYPE_MATCH 0
#define TYPE_GAP_1 1
#define TYPE_GAP_2 2
#define H_MAX (2 * 64 * 1024)

typedef struct { int type, i, j, score; } oc_crosspoint;
typedef struct { int h, f; } oc_cell4;

static int max2(int a, int b) { return a > b ? a : b; }
static int max3(int a, int b, int c) { return max2(max2(a, b), c); }

/* sw_stage4.cpp:250-271.  s0 is walked with stride ds0 (forward +1, reversed -1). */
static oc_cell4 process_col(const unsigned char* s0, int ds0, unsigned char c, int h11, int h10, oc_cell4* col, int len) {
    int f0 = -OC_INF;
    for (int j = 0; j < len; j++) {
        col[j].f = max2(col[j].h - GAP_FIRST, col[j].f - GAP_EXT);          /* the column's E (horizontal gap) */
        f0 = max2(h10 - GAP_FIRST, f0 - GAP_EXT);
        h10 = max3(h11 + ((c == s0[j * ds0]) ? SC_MATCH : SC_MISMATCH), col[j].f, f0);
        h11 = col[j].h;
        col[j].h = h10;
    }
    oc_cell4 cell;
    cell.f = f0;
    cell.h = h10;
    return cell;
}

/* sw_stage4.cpp:273-291; returns 1 found, 0 not, -1 "Error Match" */
static int match4(oc_cell4 a, oc_cell4 b, int diff, oc_crosspoint* pt) {
    int sum_match = a.h + b.h;
    int sum_gap = a.f + b.f + GAP_OPEN;
    if (sum_match == diff) { pt->type = TYPE_MATCH; pt->score = a.h; return 1; }
    if (sum_gap == diff) { pt->type = TYPE_GAP_2; pt->score = a.f; return 1; }
    if (sum_match > diff || sum_gap > diff) return -1;
    return 0;
}

/* ort_split_2, sw_stage4.cpp:293-380: seqA is the split ("vertical") sequence, rows (i0,i1], seqB columns (j0,j1] */
static int ort_split_2(const unsigned char* seqA, const unsigned char* seqB, int i0, int j0, int i1, int j1,
                       int type_s, int type_e, int score_s, int score_e, oc_crosspoint* out) {
    const int lenA = i1 - i0, lenB = j1 - j0;
    if (lenB >= H_MAX || lenA / 2 + 1 >= H_MAX) return -2;
    const int diff = score_e - score_s;
    const int imid0 = lenA / 2, imid1 = lenA - imid0;
    const int jmid1 = lenB - lenB / 2;
    const unsigned char* s0 = seqA + i0;            /* s0[k]  = row i0+k+1      */
    const unsigned char* s1 = seqB + j0;
    const unsigned char* s0r = seqA + (i1 - 1);     /* s0r[-k] = row i1-k       */
    const unsigned char* s1r = seqB + (j1 - 1);
    oc_cell4* c0 = (oc_cell4*) malloc(sizeof(oc_cell4) * (size_t) (imid0 + 1));
    oc_cell4* c1 = (oc_cell4*) malloc(sizeof(oc_cell4) * (size_t) (imid1 + 1));
    oc_cell4* r0 = (oc_cell4*) malloc(sizeof(oc_cell4) * (size_t) (jmid1 + 2));
    oc_cell4* r1 = (oc_cell4*) malloc(sizeof(oc_cell4) * (size_t) (jmid1 + 2));
    for (int i = 0; i < imid0; i++) { c0[i].h = -(i + 1) * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_2); c0[i].f = -OC_INF; }
    for (int i = 0; i < imid1; i++) { c1[i].h = -(i + 1) * GAP_EXT - GAP_OPEN; c1[i].f = -OC_INF; }
    r0[0].h = r0[0].f = c0[imid0 - 1].h;
    r1[0].h = r1[0].f = c1[imid1 - 1].h;
    int d0 = (type_s != TYPE_MATCH) ? -OC_INF : 0;
    int d1 = (type_e != TYPE_MATCH) ? -OC_INF : 0;
    int rc = -3;
    oc_crosspoint cross;
    for (int j = 0; j < lenB && rc == -3; j++) {
        int h0 = -(j + 1) * GAP_EXT - GAP_OPEN * (type_s != TYPE_GAP_1);
        oc_cell4 rr0 = process_col(s0, 1, s1[j], d0, h0, c0, imid0);
        d0 = h0;
        int h1 = -(j + 1) * GAP_EXT - GAP_OPEN;
        oc_cell4 rr1 = process_col(s0r, -1, s1r[-j], d1, h1, c1, imid1);
        d1 = h1;
        if (j + 1 <= jmid1) { r0[j + 1] = rr0; r1[j + 1] = rr1; }
        if (j + 1 >= jmid1) {
            int mt = match4(rr0, r1[lenB - (j + 1)], diff, &cross);
            if (mt < 0) { rc = -4; break; }
            if (mt) { cross.j = j0 + (j + 1); cross.i = imid0 + i0; cross.score += score_s; *out = cross; rc = 0; break; }
            mt = match4(r0[lenB - (j + 1)], rr1, diff, &cross);
            if (mt < 0) { rc = -4; break; }
            if (mt) { cross.j = j0 + (lenB - (j + 1)); cross.i = imid0 + i0; cross.score += score_s; *out = cross; rc = 0; break; }
        }
    }
    free(c0); free(c1); free(r0); free(r1);
    return rc;
}

/* one pass over all partitions: split_thread (:86-222) + merge_partitions (:786-807); returns the new count, 0 when
 * nothing was reduced, < 0 on error */
static int reduce_partitions(const unsigned char* seq0, const unsigned char* seq1, oc_crosspoint** list, int count, int max_size) {
    static const int inv_type[] = {0, 2, 1};
    oc_crosspoint* cp = *list;
    oc_crosspoint* np = (oc_crosspoint*) malloc(sizeof(oc_crosspoint) * (size_t) count);
    for (int k = 1; k < count; k++) {
        const int i0 = cp[k - 1].i, j0 = cp[k - 1].j, type0 = cp[k - 1].type, score0 = cp[k - 1].score;
        const int i1 = cp[k].i, j1 = cp[k].j, type1 = cp[k].type, score1 = cp[k].score;
        const int di = i1 - i0, dj = j1 - j0;
        np[k].type = -1;
        if (di == 0 || dj == 0) continue;
        int rc = 0;
        if (di < dj) {                                  /* "inverse": split seq1 */
            if (j0 < j1 - max_size) {
                oc_crosspoint t;
                rc = ort_split_2(seq1, seq0, j0, i0, j1, i1, inv_type[type0], inv_type[type1], score0, score1, &t);
                if (rc == 0) { np[k].i = t.j; np[k].j = t.i; np[k].type = inv_type[t.type]; np[k].score = t.score; }
            }
        } else if (i0 < i1 - max_size) {
            oc_crosspoint t;
            rc = ort_split_2(seq0, seq1, i0, j0, i1, j1, type0, type1, score0, score1, &t);
            if (rc == 0) np[k] = t;
        }
        if (rc != 0) { free(np); return rc; }
    }
    oc_crosspoint* merged = (oc_crosspoint*) malloc(sizeof(oc_crosspoint) * (size_t) (2 * count));
    int n = 0, has_new = 0;
    merged[n++] = cp[0];
    for (int k = 1; k < count; k++) {
        const int diff_pos = (np[k].i != cp[k - 1].i || np[k].j != cp[k - 1].j);
        if (np[k].type != -1 && diff_pos) { has_new = 1; merged[n++] = np[k]; }
        merged[n++] = cp[k];
    }
    free(np);
    if (!has_new) { free(merged); return 0; }
    free(cp);
    *list = merged;
    return n;
}

static int largest_partition(const oc_crosspoint* cp, int count) {
    int mi = 0, mj = 0;
    for (int k = 1; k < count; k++) {
        int di = abs(cp[k - 1].i - cp[k].i), dj = abs(cp[k - 1].j - cp[k].j);
        if (di != 0 && dj != 0) { if (mi < di) mi = di; if (mj < dj) mj = dj; }
    }
    return mi > mj ? mi : mj;
}

/* stage4(), sw_stage4.cpp:880-960.  in: the stage-3 crosspoints; *out: malloc'ed refined list (caller frees with
 * oc_stage4_free); returns the count or < 0. */
int oc_stage4(const unsigned char* seq0, const unsigned char* seq1, const oc_crosspoint* in, int count, int max_size,
              oc_crosspoint** out, int* steps) {
    oc_crosspoint* list = (oc_crosspoint*) malloc(sizeof(oc_crosspoint) * (size_t) (count > 0 ? count : 1));
    memcpy(list, in, sizeof(oc_crosspoint) * (size_t) count);
    int st = 0;
    while (largest_partition(list, count) > max_size) {
        int n = reduce_partitions(seq0, seq1, &list, count, max_size);
        if (n < 0) { free(list); return n; }
        if (n == 0) break;                              /* "Didn't reduce partition." */
        count = n;
        st++;
    }
    if (steps) *steps = st;
    *out = list;
    return count;
}

void oc_stage4_free(oc_crosspoint* p) { free(p); }
