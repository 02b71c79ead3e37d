"""oracle/aligner_double.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU stand-in for the engine behind the aligner interface the native drivers use (setSequences / alignPartition /
matchLastColumn / unsetSequences): MASA-Core's block aligner with the serial schedule, block by block over the
oracle's restatement of CPUBlockProcessor::processBlock (sw_oracle.c).  It follows, call for call, the aligner that
oracle/ref_driver.cpp links into the real MASA-Core to produce the fixtures under tests/golden/ -- so a native driver
(stage1.py, stage2.py, stage3.py) running on this double must reproduce the reference's crosspoint files BYTE FOR BYTE:
same grid, same special rows, same order of dispatches.

Follows M/libmasa/aligners/AbstractBlockAligner.cpp:276-327 (alignPartition: corner read, block loop, scores after the
grid, column-major), :418-449 (isSpecialRow / isSpecialColumn), M/libmasa/Grid.cpp:51-66, :133-171 (uniform blocks, the
last one shorter), M/libmasa/aligners/AbstractAligner.cpp:227-239 (the border tails), :158-163 (matchLastColumn).
No pruning: the traceback stages switch it off (sw_stage2.cpp:324, sw_stage3.cpp:307).

Only tests/ may import this module."""
import numpy as np

from . import binding as oracle

INF = oracle.INF


class SerialBlockAligner:
    def __init__(self, block_h, block_w):
        self.block_h, self.block_w = int(block_h), int(block_w)
        self.seq0 = self.seq1 = None
        self.cells = 0
        self.partitions = 0

    # -- IAligner ----------------------------------------------------------------------------------------------
    def setSequences(self, seq0, seq1):
        self.seq0 = np.ascontiguousarray(np.frombuffer(bytes(seq0), dtype=np.uint8) if isinstance(seq0, (bytes, bytearray)) else seq0, dtype=np.uint8)
        self.seq1 = np.ascontiguousarray(np.frombuffer(bytes(seq1), dtype=np.uint8) if isinstance(seq1, (bytes, bytearray)) else seq1, dtype=np.uint8)

    def unsetSequences(self):
        self.seq0 = self.seq1 = None

    def matchLastColumn(self, buffer, base, goal_score):
        rc, k, score, typ = oracle.match_column(buffer, base, goal_score)
        if rc == 1:
            return {"found": True, "k": k, "score": score, "type": typ}
        return {"found": False, "k": k if rc < 0 else -1, "score": 0, "type": rc if rc < 0 else 0}

    def stage4(self, crosspoints, max_partition_size=16):
        """the engine's mi355sw_stage4, on the oracle's restatement of MASA-Core's stage 4 (stage4_oracle.c)"""
        out, steps = oracle.stage4(self.seq0, self.seq1, crosspoints, max_partition_size)
        return out, {"steps": steps}

    def getStatistics(self):
        return {"strip_rows": self.block_h, "kernel_ms": 0.0, "pruned_cells": 0, "processed_cells": self.cells}

    def alignPartition(self, part, mgr):
        i0p, j0p, i1p, j1p = part.i0, part.j0, part.i1, part.j1
        bh, bw = max(self.block_h, 1), max(self.block_w, 1)
        gh = (i1p - i0p + bh - 1) // bh
        gw = (j1p - j0p + bw - 1) // bw
        rec = mgr.getRecurrenceType()
        self.partitions += 1

        def special_row(by):
            if mgr.mustDispatchLastRow() and by == gh - 1:
                return True
            if mgr.mustDispatchSpecialRows():
                interval = max((mgr.getSpecialRowInterval() + bh - 1) // bh, 1)
                return (by + 1) % interval == 0
            return False

        def special_col(bx):
            return mgr.mustDispatchLastColumn() and bx == gw - 1

        def bounds(bx, by):
            return (i0p + by * bh, j0p + bx * bw, min(i0p + (by + 1) * bh, i1p), min(j0p + (bx + 1) * bw, j1p))

        rows = [None] * gw
        scores = [[(-1, -1, -INF)] * gh for _ in range(gw)]
        tail = np.empty((1, 2), dtype=np.int32)
        mgr.receiveFirstColumn(tail, 1)                  # the corner, from both streams
        col_tail = tail[0].copy()
        mgr.receiveFirstRow(tail, 1)
        for by in range(gh):
            col = None
            for bx in range(gw):
                i0, j0, i1, j1 = bounds(bx, by)
                if by == 0:
                    rows[bx] = np.empty((j1 - j0, 2), dtype=np.int32)
                    mgr.receiveFirstRow(rows[bx], j1 - j0)
                if bx == 0:
                    col = np.empty((i1 - i0 + 1, 2), dtype=np.int32)
                    col[0] = col_tail
                    mgr.receiveFirstColumn(col[1:], i1 - i0)
                    col_tail = col[i1 - i0].copy()
                    if special_row(by):
                        c = col[i1 - i0:i1 - i0 + 1].copy()
                        c[0, 1] = -INF
                        mgr.dispatchRow(i1, c, 1)
                if by == 0 and special_col(bx):
                    c = rows[bx][j1 - j0 - 1:j1 - j0].copy()
                    c[0, 1] = -INF
                    mgr.dispatchColumn(j1, c, 1)
                scores[bx][by] = oracle.process_block(self.seq0, self.seq1, rows[bx], col, i0, j0, i1, j1, rec)
                self.cells += (i1 - i0) * (j1 - j0)
                if special_row(by):
                    mgr.dispatchRow(i1, rows[bx], j1 - j0)
                if special_col(bx):
                    mgr.dispatchColumn(j1, col[1:], i1 - i0)
        for bx in range(gw):
            for by in range(gh):
                mgr.dispatchScore(scores[bx][by], bx, by)
        if mgr.mustDispatchLastCell():
            mgr.dispatchScore((i1p - 1, j1p - 1, int(rows[gw - 1][-1, 0])), gw - 1, gh - 1)
