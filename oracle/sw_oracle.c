/*
 * oracle/sw_oracle.c -- TEST INFRASTRUCTURE ONLY (see sw_oracle.h header note).
 * CPU restatement of the reference's Stage-1 path; parity PINNED against
 * oracle/_ref (real MASA-Core) by tests/test_oracle_vs_reference.py and
 * against the tests/golden fixtures.
 *
 * Citations: "M/" = /root/reference/masa-cudalign-4.0.2.1028/libs/masa-core/src/
 */
#include "sw_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAX2(a, b) (((a) > (b)) ? (a) : (b))

/* ------------------------------------------------------------------------- *
 * CPUBlockProcessor::processBlock  (M/libmasa/processors/CPUBlockProcessor.cpp:113-174)
 * with the sw()/nw() recurrences of :66-93:
 *   E = max(Hleft - OPEN, E) - EXT ; F = max(Hup - OPEN, F) - EXT
 *   v = Hdiag + (c1 != c0 ? MISMATCH : MATCH)
 *   H = SW ? max(0, v, E, F) : max(v, E, F)
 * row[k] = (H,F) of cell (i0-1, j0+k) in / last block row out
 * col[0] = diagonal H(i0-1,j0-1); col[k+1] = (H,E) of cell (i0+k, j0-1) in / last block column out
 * best = first strict maximum in row-major order (:151-155).
 * ------------------------------------------------------------------------- */
oc_score oracle_process_block(const unsigned char* seq0_, const unsigned char* seq1_,
        oc_cell* row, oc_cell* col, int i0, int j0, int i1, int j1, int recurrence) {
    oc_score best;
    best.i = -1;
    best.j = -1;
    best.score = -OC_INF;
    const unsigned char* seq0 = seq0_ + i0;
    const unsigned char* seq1 = seq1_ + j0;
    const int sw = (recurrence == OC_SMITH_WATERMAN);
    int h11 = col[0].h;
    for (int i = 0; i < i1 - i0; i++) {
        int h01 = col[i + 1].h;
        int e00 = col[i + 1].f;
        const unsigned char c = seq0[i];
        for (int j = 0; j < j1 - j0; j++) {
            int h10 = row[j].h;
            int f10 = row[j].f;
            e00 = MAX2(h01 - OC_GAP_OPEN, e00) - OC_GAP_EXT;
            f10 = MAX2(h10 - OC_GAP_OPEN, f10) - OC_GAP_EXT;
            int v1 = h11 + ((seq1[j] != c) ? OC_MISMATCH : OC_MATCH);
            int h00 = MAX2(v1, MAX2(e00, f10));
            if (sw) h00 = MAX2(h00, 0);
            h11 = h10;
            h01 = h00;
            row[j].h = h00;
            row[j].f = f10;
            if (best.score < h00) {
                best.score = h00;
                best.i = i0 + i;
                best.j = j0 + j;
            }
        }
        if (i == 0) col[0].h = h11;
        h11 = col[i + 1].h;
        col[i + 1].h = h01;
        col[i + 1].f = e00;
    }
    return best;
}

/* InitialCellsReader::read (M/common/io/InitialCellsReader.cpp:84-108) */
void oracle_initial_cells(int type, int position, oc_cell* buffer, int len) {
    if (type == OC_INIT_WITH_GAPS || type == OC_INIT_WITH_GAPS_OPENED) {
        const int open = (type == OC_INIT_WITH_GAPS) ? OC_GAP_OPEN : 0;
        int k = 0;
        if (position == 0 && len > 0) {
            buffer[0].h = 0;
            buffer[0].f = -OC_INF;
            k++;
        }
        for (; k < len; k++) {
            buffer[k].h = -OC_GAP_EXT * (position + k) - open;
            buffer[k].f = -OC_INF;
        }
    } else {
        for (int k = 0; k < len; k++) {
            buffer[k].h = 0;
            buffer[k].f = -OC_INF;
        }
    }
}

/* AlignerUtils::matchColumn (M/libmasa/utils/AlignerUtils.cpp:50-107) */
int oracle_match_column(const oc_cell* buffer, const oc_cell* base, int len, int goal,
        int* k_out, int* score, int* type) {
    for (int k = 0; k < len; k++) {
        int sum_match = base[k].h + buffer[k].h;
        int sum_gap = base[k].f + buffer[k].f + OC_GAP_OPEN;
        if (sum_match == goal) {
            *k_out = k; *score = base[k].h; *type = 0; /* MATCH_ALIGNED */
            return 1;
        } else if (sum_gap == goal) {
            *k_out = k; *score = base[k].f; *type = 1; /* MATCH_GAPPED */
            return 1;
        } else if (sum_match > goal || sum_gap > goal) {
            *k_out = k;
            return sum_match > goal ? -1 : -2;      /* MATCH_ERROR_1 / _2 */
        }
    }
    return 0;
}

/* ---- border streams (AlignerManager::receiveFirstRow/Column, AlignerManager.cpp:318-332) ---- */
typedef struct {
    int type;
    int position;            /* absolute position for generated borders */
    const oc_cell* custom;   /* custom data, indexed from 0 = corner */
    int consumed;
    oc_cell tail;            /* AbstractAligner::firstRowTail / firstColumnTail (AbstractAligner.cpp:226-238) */
} border_t;

static void border_read(border_t* b, oc_cell* buf, int len) {
    if (b->type == OC_INIT_WITH_CUSTOM_DATA) {
        memcpy(buf, b->custom + b->consumed, sizeof(oc_cell) * (size_t) len);
    } else {
        oracle_initial_cells(b->type, b->position, buf, len);
    }
    b->position += len;
    b->consumed += len;
    if (len > 0) b->tail = buf[len - 1];
}

/* ---- BestScoreList with limit 1 (M/common/BestScoreList.hpp:30-38, .cpp:129-195) ---- *
 * order: score desc, then i asc, then j asc; scores < min_score ignored.          */
static void best_add(oc_score* best, int min_score, int i, int j, int score) {
    if (score < min_score) return;
    if (best->score == -OC_INF && best->i == -1) {
        best->i = i; best->j = j; best->score = score;
        return;
    }
    int d = score - best->score;
    if (d > 0 || (d == 0 && (i < best->i || (i == best->i && j < best->j)))) {
        best->i = i; best->j = j; best->score = score;
    }
}

/* ---- AbstractBlockPruning::isBlockPrunable (M/libmasa/pruning/AbstractBlockPruning.cpp:70-111) ---- */
typedef struct {
    int enabled, recurrence, max_i, max_j, best;
    int gw, gh;
    unsigned char* k;    /* (gh+1) x (gw+1), BlockPruningGenericN2.cpp:60-74 */
} pruner_t;

static int prunable(pruner_t* pr, int i0, int j0, int i1, int j1, int score) {
    int distI = pr->max_i - i0;
    int distJ = pr->max_j - j0;
    int distMin = distI < distJ ? distI : distJ;
    int inc = distMin * OC_MATCH;
    int dec = 0;
    if (pr->recurrence == OC_NEEDLEMAN_WUNSCH) {
        int bmax = MAX2(j1 - j0, i1 - i0);
        int gaps = abs(distJ - distI) - bmax;
        if (gaps > 0) inc -= OC_GAP_OPEN + gaps * OC_GAP_EXT;
        dec -= distMin * OC_MISMATCH;
        gaps = abs(distJ - distI) + bmax;
        dec += OC_GAP_OPEN + gaps * OC_GAP_EXT;
    }
    if (pr->best < score - dec) pr->best = score - dec;
    return (score + inc) <= pr->best;
}

static int grid_count(int len, int b) { return (len + b - 1) / b; }

int oracle_stage1(const oc_params* p, oc_result* r) {
    memset(r, 0, sizeof(*r));
    r->best.i = -1; r->best.j = -1; r->best.score = -OC_INF;
    const int m = p->m, n = p->n;
    if (m <= 0 || n <= 0) return 0;   /* AlignerManager.cpp:96-99: zero-area partition skipped */
    const int bh = p->block_h > 0 ? p->block_h : 1024;
    const int bw = p->block_w > 0 ? p->block_w : 1024;
    const int gw = grid_count(n, bw), gh = grid_count(m, bh);

    oc_cell* rowbuf = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) n);
    oc_cell* colbuf = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) (bh + 1));
    /* per block-row column buffers are independent in the reference (col[by]); since the
     * serial schedule finishes a block row before the next starts, one buffer suffices. */
    oc_score* scores = (oc_score*) malloc(sizeof(oc_score) * (size_t) gw * gh);

    border_t frow, fcol;
    memset(&frow, 0, sizeof(frow)); memset(&fcol, 0, sizeof(fcol));
    frow.type = p->first_row_type; frow.position = p->row_start_offset; frow.custom = p->custom_first_row;
    fcol.type = p->first_col_type; fcol.position = p->col_start_offset; fcol.custom = p->custom_first_col;
    frow.tail.h = fcol.tail.h = -OC_INF; frow.tail.f = fcol.tail.f = -OC_INF;

    /* special-row bookkeeping (AbstractBlockAligner.cpp:418-439) */
    int interval_blocks = 0;
    if (p->special_row_interval > 0) {
        interval_blocks = (p->special_row_interval + bh - 1) / bh;
        if (interval_blocks <= 0) interval_blocks = 1;
    }
    int cap_rows = 0;
    for (int by = 0; by < gh; by++)
        if ((interval_blocks && (by + 1) % interval_blocks == 0)) cap_rows++;
    if (cap_rows) {
        r->special_rows = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) cap_rows * (n + 1));
        r->special_row_ids = (int*) malloc(sizeof(int) * (size_t) cap_rows);
    }
    if (p->want_last_row) r->last_row = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) (n + 1));
    if (p->want_last_col) r->last_col = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) (m + 1));
    int last_col_pos = 0;

    pruner_t pr;
    memset(&pr, 0, sizeof(pr));
    pr.enabled = p->pruning;
    pr.recurrence = p->recurrence;
    pr.max_i = p->max_i > 0 ? p->max_i : m;
    pr.max_j = p->max_j > 0 ? p->max_j : n;
    pr.best = -OC_INF;
    pr.gw = gw; pr.gh = gh;
    if (pr.enabled) {
        pr.k = (unsigned char*) calloc((size_t) (gh + 1) * (gw + 1), 1);
        for (int i = 1; i <= gh; i++) pr.k[(size_t) i * (gw + 1)] = 1;
        for (int j = 1; j <= gw; j++) pr.k[j] = 1;
    }
#define K(by, bx) pr.k[(size_t) (by) * (gw + 1) + (bx)]

    /* AbstractBlockAligner.cpp:289-292: corner cell read from both borders */
    oc_cell dummy;
    border_read(&fcol, &dummy, 1);
    border_read(&frow, &dummy, 1);

    for (int by = 0; by < gh; by++) {
        const int i0 = by * bh, i1 = (i0 + bh > m) ? m : i0 + bh;
        const int special = (interval_blocks && (by + 1) % interval_blocks == 0);
        const int lastrow = (p->want_last_row && by == gh - 1);
        oc_cell* srow = NULL;
        if (special) {
            srow = r->special_rows + (size_t) r->n_special_rows * (n + 1);
            r->special_row_ids[r->n_special_rows++] = i1;
        }
        for (int bx = 0; bx < gw; bx++) {
            const int j0 = bx * bw, j1 = (j0 + bw > n) ? n : j0 + bw;
            oc_cell* row = rowbuf + j0;
            if (by == 0) border_read(&frow, row, j1 - j0);
            if (bx == 0) {
                colbuf[0] = fcol.tail;
                border_read(&fcol, colbuf + 1, i1 - i0);
                if (special || lastrow) {
                    oc_cell c = colbuf[i1 - i0];
                    c.f = -OC_INF;
                    if (special) srow[0] = c;
                    if (lastrow) r->last_row[0] = c;
                }
            }
            if (by == 0 && p->want_last_col && bx == gw - 1) {
                oc_cell c = row[j1 - j0 - 1];
                c.f = -OC_INF;
                r->last_col[last_col_pos++] = c;
            }
            oc_score s;
            s.i = -1; s.j = -1; s.score = -OC_INF;
            r->blocks_total++;
            int pruned = 0;
            if (pr.enabled) {
                /* BlockPruningGenericN2::isBlockPruned (BlockPruningGenericN2.cpp:48-56) */
                if (K(by + 1, bx) && K(by, bx) && K(by, bx + 1)) {
                    K(by + 1, bx + 1) = 1;
                    pruned = 1;
                }
            }
            if (!pruned) {
                s = oracle_process_block(p->seq0, p->seq1, row, colbuf, i0, j0, i1, j1, p->recurrence);
                if (pr.enabled) {
                    /* BlockPruningGenericN2::pruningUpdate (:39-46) */
                    if (pr.best < s.score) pr.best = s.score;
                    if (prunable(&pr, i0, j0, i1, j1, s.score)) K(by + 1, bx + 1) = 1;
                }
            } else {
                r->blocks_pruned++;
            }
            scores[(size_t) bx * gh + by] = s;
            if (special) memcpy(srow + 1 + j0, row, sizeof(oc_cell) * (size_t) (j1 - j0));
            if (lastrow) memcpy(r->last_row + 1 + j0, row, sizeof(oc_cell) * (size_t) (j1 - j0));
            if (p->want_last_col && bx == gw - 1) {
                memcpy(r->last_col + last_col_pos, colbuf + 1, sizeof(oc_cell) * (size_t) (i1 - i0));
                last_col_pos += i1 - i0;
            }
        }
    }

    /* dispatchScore loop (AbstractBlockAligner.cpp:310-315) + AlignerManager::dispatchScore (:411-450) */
    const int min_score = (p->best_mode == OC_BEST_ANYWHERE && p->recurrence == OC_SMITH_WATERMAN) ? 0 : -OC_INF;
    for (int bx = 0; bx < gw; bx++) {
        for (int by = 0; by < gh; by++) {
            oc_score s = scores[(size_t) bx * gh + by];
            if (s.score > -OC_INF && p->best_mode == OC_BEST_ANYWHERE)
                best_add(&r->best, min_score, s.i + 1, s.j + 1, s.score);
        }
    }
    if (p->best_mode == OC_BEST_LAST_CELL) {
        /* AbstractBlockAligner.cpp:317-323; AlignerManager.cpp:430-434 */
        best_add(&r->best, -OC_INF, m, n, rowbuf[n - 1].h);
    }
    if ((p->best_mode == OC_BEST_LAST_ROW || p->best_mode == OC_BEST_LAST_ROW_OR_COL) && r->last_row) {
        /* AlignerManager::dispatchRow (:386-393) + findBestCell (:604-616), per dispatched chunk:
         * first the border cell alone, then one chunk per block; j = J0 + lastRowPos + best_id */
        best_add(&r->best, -OC_INF, m, 0, r->last_row[0].h);
        for (int bx = 0; bx < gw; bx++) {
            int j0 = bx * bw, j1 = (j0 + bw > n) ? n : j0 + bw;
            int bid = 1 + j0, bsc = -OC_INF;
            for (int k = 1 + j0; k < 1 + j1; k++) if (bsc < r->last_row[k].h) { bsc = r->last_row[k].h; bid = k; }
            best_add(&r->best, -OC_INF, m, bid, bsc);
        }
    }
    if ((p->best_mode == OC_BEST_LAST_COL || p->best_mode == OC_BEST_LAST_ROW_OR_COL) && r->last_col) {
        /* AlignerManager::dispatchColumn (:340-357): i = I0 + lastColumnPos + best_id, j = J1 */
        best_add(&r->best, -OC_INF, 0, n, r->last_col[0].h);
        for (int by = 0; by < gh; by++) {
            int i0 = by * bh, i1 = (i0 + bh > m) ? m : i0 + bh;
            int bid = 1 + i0, bsc = -OC_INF;
            for (int k = 1 + i0; k < 1 + i1; k++) if (bsc < r->last_col[k].h) { bsc = r->last_col[k].h; bid = k; }
            best_add(&r->best, -OC_INF, bid, n, bsc);
        }
    }

    /* block scores as AbstractBlockAligner dispatches them (dispatchScore(score, bx, by), :343-346): kept for the
     * tests of the engine's block-score grid; column-major like the reference's score table: [bx * gh + by] */
    r->block_scores = scores; r->grid_w = gw; r->grid_h = gh;
    free(rowbuf); free(colbuf); free(pr.k);
#undef K
    return 0;
}

void oracle_free_result(oc_result* r) {
    free(r->special_row_ids); free(r->special_rows); free(r->last_row); free(r->last_col); free(r->block_scores);
    memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------- *
 * Multi-threaded anti-diagonal wavefront of blocks (no pruning).  Same cells,
 * same per-block best, same canonical reduction => same result as the serial
 * schedule.  The "port" CPU baseline in bench.py; the tests use it where the
 * serial schedule would take minutes (special rows, last row and last column are
 * the same cells whatever the block schedule).
 * ------------------------------------------------------------------------- */
typedef struct {
    const oc_params* p;
    int gw, gh, bh, bw;
    oc_cell* rowbuf;       /* n cells: shared horizontal bus */
    oc_cell* colbufs;      /* gh x (bh+1): per block-row vertical bus */
    oc_score* scores;
    int diag;              /* current anti-diagonal */
    volatile int next;     /* next block index on the diagonal */
    pthread_barrier_t bar;
    int threads;
    int interval_blocks;   /* special rows: every interval_blocks-th block row (0 = none) */
    oc_result* r;
} mt_ctx;

static void* mt_worker(void* arg) {
    mt_ctx* c = (mt_ctx*) arg;
    const oc_params* p = c->p;
    for (int d = 0; d < c->gw + c->gh - 1; d++) {
        for (;;) {
            int k = __sync_fetch_and_add(&c->next, 1);
            int by_lo = d - (c->gw - 1); if (by_lo < 0) by_lo = 0;
            int by = by_lo + k;
            int bx = d - by;
            if (by >= c->gh || bx < 0) break;
            int i0 = by * c->bh, i1 = (i0 + c->bh > p->m) ? p->m : i0 + c->bh;
            int j0 = bx * c->bw, j1 = (j0 + c->bw > p->n) ? p->n : j0 + c->bw;
            oc_cell* row = c->rowbuf + j0;
            oc_cell* col = c->colbufs + (size_t) by * (c->bh + 1);
            const int special = (c->interval_blocks && (by + 1) % c->interval_blocks == 0);
            const int lastrow = (c->r->last_row != NULL && by == c->gh - 1);
            oc_cell* srow = special ? c->r->special_rows + (size_t) ((by + 1) / c->interval_blocks - 1) * (p->n + 1) : NULL;
            if (bx == 0 && (special || lastrow)) {          /* the row's first cell: the first column's, F void (as the serial schedule) */
                oc_cell f = col[i1 - i0];
                f.f = -OC_INF;
                if (special) srow[0] = f;
                if (lastrow) c->r->last_row[0] = f;
            }
            if (by == 0 && bx == c->gw - 1 && c->r->last_col != NULL) {
                oc_cell f = row[j1 - j0 - 1];
                f.f = -OC_INF;
                c->r->last_col[0] = f;
            }
            c->scores[(size_t) bx * c->gh + by] = oracle_process_block(p->seq0, p->seq1, row, col, i0, j0, i1, j1, p->recurrence);
            if (special) memcpy(srow + 1 + j0, row, sizeof(oc_cell) * (size_t) (j1 - j0));
            if (lastrow) memcpy(c->r->last_row + 1 + j0, row, sizeof(oc_cell) * (size_t) (j1 - j0));
            if (bx == c->gw - 1 && c->r->last_col != NULL) memcpy(c->r->last_col + 1 + i0, col + 1, sizeof(oc_cell) * (size_t) (i1 - i0));
        }
        pthread_barrier_wait(&c->bar);
        if (__sync_bool_compare_and_swap(&c->diag, d, d + 1)) c->next = 0;
        pthread_barrier_wait(&c->bar);
    }
    return NULL;
}

int oracle_stage1_mt(const oc_params* p, oc_result* r, int threads) {
    memset(r, 0, sizeof(*r));
    r->best.i = -1; r->best.j = -1; r->best.score = -OC_INF;
    if (p->m <= 0 || p->n <= 0) return 0;
    if (threads < 1) threads = 1;
    mt_ctx c;
    memset(&c, 0, sizeof(c));
    c.p = p;
    c.bh = p->block_h > 0 ? p->block_h : 1024;
    c.bw = p->block_w > 0 ? p->block_w : 1024;
    c.gw = grid_count(p->n, c.bw); c.gh = grid_count(p->m, c.bh);
    c.rowbuf = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) p->n);
    c.colbufs = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) c.gh * (c.bh + 1));
    c.scores = (oc_score*) malloc(sizeof(oc_score) * (size_t) c.gw * c.gh);
    c.threads = threads;
    c.r = r;
    if (p->special_row_interval > 0) {                      /* AbstractBlockAligner.cpp:418-439, as oracle_stage1 */
        c.interval_blocks = (p->special_row_interval + c.bh - 1) / c.bh;
        if (c.interval_blocks <= 0) c.interval_blocks = 1;
        const int rows = c.gh / c.interval_blocks;
        if (rows > 0) {
            r->special_rows = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) rows * (p->n + 1));
            r->special_row_ids = (int*) malloc(sizeof(int) * (size_t) rows);
            for (int k = 0; k < rows; k++) {
                const int i1 = (k + 1) * c.interval_blocks * c.bh;
                r->special_row_ids[k] = i1 > p->m ? p->m : i1;
            }
            r->n_special_rows = rows;
        }
    }
    if (p->want_last_row) r->last_row = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) (p->n + 1));
    if (p->want_last_col) r->last_col = (oc_cell*) malloc(sizeof(oc_cell) * (size_t) (p->m + 1));
    /* borders: corner + first row, and the whole first column split per block row */
    border_t frow, fcol;
    memset(&frow, 0, sizeof(frow)); memset(&fcol, 0, sizeof(fcol));
    frow.type = p->first_row_type; frow.position = p->row_start_offset; frow.custom = p->custom_first_row;
    fcol.type = p->first_col_type; fcol.position = p->col_start_offset; fcol.custom = p->custom_first_col;
    oc_cell dummy;
    border_read(&fcol, &dummy, 1);
    border_read(&frow, &dummy, 1);
    border_read(&frow, c.rowbuf, p->n);
    for (int by = 0; by < c.gh; by++) {
        int i0 = by * c.bh, i1 = (i0 + c.bh > p->m) ? p->m : i0 + c.bh;
        oc_cell* col = c.colbufs + (size_t) by * (c.bh + 1);
        col[0] = fcol.tail;
        border_read(&fcol, col + 1, i1 - i0);
    }
    pthread_barrier_init(&c.bar, NULL, (unsigned) threads);
    pthread_t* th = (pthread_t*) malloc(sizeof(pthread_t) * (size_t) threads);
    for (int t = 1; t < threads; t++) pthread_create(&th[t], NULL, mt_worker, &c);
    mt_worker(&c);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&c.bar);
    const int min_score = (p->best_mode == OC_BEST_ANYWHERE && p->recurrence == OC_SMITH_WATERMAN) ? 0 : -OC_INF;
    if (p->best_mode == OC_BEST_ANYWHERE) {
        for (int bx = 0; bx < c.gw; bx++)
            for (int by = 0; by < c.gh; by++) {
                oc_score s = c.scores[(size_t) bx * c.gh + by];
                if (s.score > -OC_INF) best_add(&r->best, min_score, s.i + 1, s.j + 1, s.score);
            }
    } else if (p->best_mode == OC_BEST_LAST_CELL) {
        best_add(&r->best, -OC_INF, p->m, p->n, c.rowbuf[p->n - 1].h);
    }
    r->blocks_total = (long long) c.gw * c.gh;
    free(th); free(c.rowbuf); free(c.colbufs); free(c.scores);
    return 0;
}
