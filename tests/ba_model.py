"""An independent, cell-level model of the Rust block aligner (lib/mmseqs/lib/block-aligner/src/scan_block.rs), written from the crate's
source only -- it shares no code and no data structure with foldseek_amd/csrc/host/block_aligner.cpp, which restates the AVX2 lane
arithmetic vector by vector.  TEST INFRASTRUCTURE.

What it states (line numbers: scan_block.rs):
  * align_core (:120-630): the block TRAJECTORY -- start with a Grow to min_size, then per iteration: offsets (off = off_max of the previous
    step, everything stored relative to it around ZERO = 2^14), the region computed for a Right / Down shift by STEP = 8 or for a Grow (a down
    region then a right region), the block maximum, the checkpoint taken at every new best, X-drop termination after X_DROP_ITER = 2 bad
    steps in a row (:477-488), the forced directions at the sequence ends (:496-506), the grow rule (no new best for block_size / STEP
    iterations, or a Grow that brought no new best: back to the checkpoint with twice the size, :509-540), the shrink rule (the maxima of the
    last two entries of the bottom row / right column reach the block maximum: halve, :543-585) and the direction rule (Down only if the
    prefix maximum of the bottom row EXCEEDS that of the right column, :588-595);
  * place_block (:1302-1443 / :1140-1280): the recurrences cell by cell in the crate's relative int16 domain (saturating adds; MIN = 0 is
    the "nothing here" value of cells outside the block), the best cell of a region per vector LANE with "last one wins" on ties and the
    crate's choice between lanes (largest column, then largest row, :374-379), the trace bits;
  * Trace (:1726-2007): the stack of computed regions with checkpoint / restore, and the traceback through OP_LUT -- whose preference
    between the two gap states swaps with the orientation of the region a cell was computed in.
A region is `right` (vectors run down the query, one reference column per outer step) or `down` (transposed); the model computes both
with one routine on transposed views, as the crate does."""

L, STEP, X_DROP_ITER, SHRINK_SUFFIX_LEN = 16, 8, 2, 2
ZERO, MIN = 1 << 14, 0
NULL = None
GROW, RIGHT, DOWN = "grow", "right", "down"


def sat(x):
    return -32768 if x < -32768 else (32767 if x > 32767 else x)


class Trace:
    def __init__(self):
        self.blocks = []          # (i, j, height, width, right, {(i, j): (t, t2)})
        self.ckpt = 0

    def add_block(self, i, j, width, height, right):
        self.blocks.append((i, j, height, width, right, {}))

    def save_ckpt(self):
        self.ckpt = len(self.blocks)

    def restore_ckpt(self):
        del self.blocks[self.ckpt:]

    def rectangles(self):
        return [(b[0], b[1], b[2], b[3], int(b[4])) for b in self.blocks]

    def cigar(self, i, j):
        """:1844-2007, ops from the end cell backwards; returned in forward order as run-length text"""
        ops, table, bi = [], "D", len(self.blocks)
        while i > 0 or j > 0:
            while True:
                bi -= 1
                b_i, b_j, h, w, right, cells = self.blocks[bi]
                if i >= b_i and j >= b_j:
                    break
            while i >= b_i and j >= b_j and (i > 0 or j > 0):
                t, t2 = cells[(i, j)]
                if right:
                    if table == "C":
                        op, table = "D", ("D" if t2 & 1 else "C")
                    elif table == "R":
                        op, table = "I", ("D" if t2 & 2 else "R")
                    elif t == 0:
                        op = "M"
                    elif t & 1:
                        op, table = "D", ("D" if t2 & 1 else "C")
                    else:
                        op, table = "I", ("D" if t2 & 2 else "R")
                else:
                    if table == "R":
                        op, table = "I", ("D" if t2 & 1 else "R")
                    elif table == "C":
                        op, table = "D", ("D" if t2 & 2 else "C")
                    elif t == 0:
                        op = "M"
                    elif t & 1:
                        op, table = "I", ("D" if t2 & 1 else "R")
                    else:
                        op, table = "D", ("D" if t2 & 2 else "C")
                ops.append(op)
                if op == "M":
                    i -= 1; j -= 1
                elif op == "I":
                    i -= 1
                else:
                    j -= 1
        ops.reverse()
        out, k = "", 0
        while k < len(ops):
            e = k
            while e < len(ops) and ops[e] == ops[k]:
                e += 1
            out += f"{e - k}{ops[k]}"
            k = e
        return out


class BlockModel:
    def __init__(self, q, r, score, gap_open, gap_extend, min_size, max_size, x_drop=None, q_bias=None, r_bias=None, score2=None, q2=None, r2=None):
        """q, r: sequences of symbols; score(a, b) -> int8 score of two symbols.  x_drop None: the global variant (Block<TRACE, false>).
        The 3Di form (align_3di): score2 / q2 / r2 = the second matrix and second pair of strings, q_bias / r_bias the position biases."""
        assert gap_open < gap_extend < 0
        self.q, self.r, self.q2, self.r2 = list(q), list(r), (None if q2 is None else list(q2)), (None if r2 is None else list(r2))
        self.score, self.score2, self.qb, self.rb = score, score2, q_bias, r_bias
        self.open, self.ext = gap_open, gap_extend
        self.min_size, self.max_size = max(min_size, L), max(max_size, L)
        self.x_drop = x_drop
        self.trace = Trace()
        n = self.max_size + L
        self.D_col, self.C_col, self.D_row, self.R_row = [MIN] * n, [MIN] * n, [MIN] * n, [MIN] * n
        self.ck = None

    # padded, 1-indexed views (PaddedBytes :2149-2245): index 0 and everything past the end is NULL
    def _sym(self, s, k):
        return s[k - 1] if 1 <= k <= len(s) else NULL

    def _cell_score(self, i, j):
        a, b = self._sym(self.q, i), self._sym(self.r, j)
        s = -128 if a is NULL or b is NULL else self.score(a, b)
        if self.score2 is None:
            return s
        a2, b2 = self._sym(self.q2, i), self._sym(self.r2, j)
        s2 = -128 if a2 is NULL or b2 is NULL else self.score2(a2, b2)
        qb = self.qb[i - 1] if 1 <= i <= len(self.q) else 0
        rb = self.rb[j - 1] if 1 <= j <= len(self.r) else 0
        return sat(sat(s + s2) + sat(rb + qb))

    def _place(self, right, start_v, start_w, width, height, col_d, col_c, col_base, row_d, row_r, row_base, corner):
        """one region.  v = the coordinate the vectors run along (query rows for a right region, reference columns for a down region),
        w = the coordinate walked one per outer step.  col_d / col_c: the boundary next to the region on the low-w side, overwritten with
        the region's last w-line; row_d / row_r receive, per w, the D and R of the last v.  Returns per-lane (max, arg_v, arg_w)."""
        d_max, arg_v, arg_w = [MIN] * L, [0] * L, [0] * L
        cells = self.trace.blocks[-1][5]
        if width == 0 or height == 0:
            return d_max, arg_v, arg_w
        for w in range(width):
            old = col_d[col_base:col_base + height]
            r_incl, prev_r_is_open = MIN, False
            last_d = last_r = MIN
            for v in range(height):
                d10, c10 = old[v], col_c[col_base + v]
                d00 = corner if v == 0 else old[v - 1]
                gi, gj = (start_v + v, start_w + w) if right else (start_w + w, start_v + v)
                d11 = sat(d00 + self._cell_score(gi, gj))
                if gi == 0 and gj == 0:
                    d11 = ZERO
                c_open_v = sat(d10 + self.open)
                c11 = max(sat(c10 + self.ext), c_open_v)
                d11 = max(d11, c11)
                d_open = sat(d11 + sat(self.open - self.ext))
                # inclusive scan down the vector dimension; across vectors the crate carries the last lane of the previous vector (:1366)
                r11 = max(d_open, sat(r_incl + self.ext)) if v > 0 else max(d_open, sat(MIN + self.ext))
                d11 = max(d11, r11)
                t = (1 if d11 == c11 else 0) | (2 if d11 == r11 else 0)
                t2 = (1 if c11 == c_open_v else 0) | (2 if prev_r_is_open else 0)
                prev_r_is_open = r11 == d_open
                r_incl = r11
                cells[(gi, gj)] = (t, t2)
                lane = v % L
                if d11 >= d_max[lane]:                                        # :1412-1418: "last one wins" on ties
                    d_max[lane], arg_v[lane], arg_w[lane] = max(d_max[lane], d11), v - lane, w
                col_d[col_base + v], col_c[col_base + v] = d11, c11
                last_d, last_r = d11, r11
            corner = MIN
            row_d[row_base + w], row_r[row_base + w] = last_d, last_r
            if self.x_drop is None:                                            # :1427-1436 (global variant only): nothing left to compute
                len_v, len_w = (len(self.q), len(self.r)) if right else (len(self.r), len(self.q))
                if start_v + height > len_v and start_w + w >= len_w:
                    break
        return d_max, arg_v, arg_w

    def _prefix_max(self, buf):
        return max(buf[:STEP])

    def _suffix_max(self, buf, n):
        return max(buf[n - SHRINK_SUFFIX_LEN:n])

    def _shift_and_offset(self, bs, b1, b2, t1, t2, off_add):
        corner = sat(b1[STEP - 1] + off_add)
        for buf, tmp in ((b1, t1), (b2, t2)):
            new = [sat(buf[k + STEP] + off_add) for k in range(bs - STEP)] + list(tmp[:STEP])
            buf[:bs] = new
        return corner

    def align(self):
        qn, rn = len(self.q), len(self.r)
        best_max, best_i, best_j = 0, 0, 0
        prev_dir = direction = GROW
        prev_size, bs = 0, self.min_size
        off = off_max = 0
        y_drop_iter = x_drop_iter = 0
        si = sj = 0
        i_ck = j_ck = off_ck = 0
        corner = MIN
        tr = self.trace
        D_col, C_col, D_row, R_row = self.D_col, self.C_col, self.D_row, self.R_row
        self.steps = []
        # how often a decision of the trajectory sat exactly ON its boundary (the tests want such inputs): direction rule with equal maxima,
        # shrink rule with equality, X-drop threshold met exactly / missed by one, the grow counter at its limit, a second bad X-drop step
        self.ties = {"dir_equal": 0, "shrink_equal": 0, "xdrop_at_threshold": 0, "xdrop_one_below": 0, "grow_at_limit": 0, "xdrop_second_step": 0,
                     "best_equal": 0, "grow_twice": 0}
        while True:
            prev_off = off
            grow_max_l, grow_av, grow_aw = [MIN] * L, [0] * L, [0] * L
            if direction == RIGHT:
                off = off_max
                off_add = sat(prev_off - off)
                tr.add_block(si, sj + bs - STEP, STEP, bs, True)
                for k in range(bs):
                    D_col[k], C_col[k] = sat(D_col[k] + off_add), sat(C_col[k] + off_add)
                t1, t2 = [MIN] * L, [MIN] * L
                d_max, av, aw = self._place(True, si, sj + bs - STEP, STEP, bs, D_col, C_col, 0, t1, t2, 0,
                                            sat(corner + off_add) if prev_dir == DOWN else MIN)
                right_max = self._prefix_max(D_col)
                corner = self._shift_and_offset(bs, D_row, R_row, t1, t2, off_add)
                down_max = self._prefix_max(D_row)
            elif direction == DOWN:
                off = off_max
                off_add = sat(prev_off - off)
                tr.add_block(si + bs - STEP, sj, bs, STEP, False)
                for k in range(bs):
                    D_row[k], R_row[k] = sat(D_row[k] + off_add), sat(R_row[k] + off_add)
                t1, t2 = [MIN] * L, [MIN] * L
                d_max, av, aw = self._place(False, sj, si + bs - STEP, STEP, bs, D_row, R_row, 0, t1, t2, 0,
                                            sat(corner + off_add) if prev_dir == RIGHT else MIN)
                down_max = self._prefix_max(D_row)
                corner = self._shift_and_offset(bs, D_col, C_col, t1, t2, off_add)
                right_max = self._prefix_max(D_col)
            else:
                corner = MIN
                grow_step = bs - prev_size
                tr.add_block(si + prev_size, sj, prev_size, grow_step, False)
                grow_max_l, grow_av, grow_aw = self._place(False, sj, si + prev_size, grow_step, prev_size, D_row, R_row, 0, D_col, C_col, prev_size, MIN)
                tr.add_block(si, sj + prev_size, grow_step, bs, True)
                d_max, av, aw = self._place(True, si, sj + prev_size, grow_step, bs, D_col, C_col, 0, D_row, R_row, prev_size, MIN)
                right_max, down_max = self._prefix_max(D_col), self._prefix_max(D_row)
                self.ck = (D_col[:bs], C_col[:bs], D_row[:bs], R_row[:bs])
                tr.save_ckpt()
            self.steps.append((direction, si, sj, bs))
            prev_dir = direction
            d_max_max, grow_max = max(d_max), max(grow_max_l)
            mx = max(d_max_max, grow_max)
            off_max = off + mx - ZERO
            y_drop_iter += 1
            grow_no_max = direction == GROW
            if off_max == best_max and best_max > 0:
                self.ties["best_equal"] += 1
            if off_max > best_max:
                if self.x_drop is not None:
                    use_grow = direction == GROW and d_max_max < grow_max
                    cur_max, cm, ca_v, ca_w = (grow_max, grow_max_l, grow_av, grow_aw) if use_grow else (d_max_max, d_max, av, aw)
                    bi_, bj_ = 0, 0
                    for lane in range(L):
                        if cm[lane] != cur_max:
                            continue
                        idx_i, idx_j = ca_v[lane], ca_w[lane]
                        r_, c_ = idx_i + lane, (bs - STEP) + idx_j
                        if use_grow:
                            gi, gj = si + prev_size + idx_j, sj + idx_i + lane
                        elif direction == RIGHT:
                            gi, gj = si + r_, sj + c_
                        elif direction == DOWN:
                            gi, gj = si + c_, sj + r_
                        else:
                            gi, gj = si + idx_i + lane, sj + prev_size + idx_j
                        if (gj > bj_) if gj != bj_ else (gi > bi_):
                            bi_, bj_ = gi, gj
                    best_i, best_j = bi_, bj_
                if bs < self.max_size:
                    i_ck, j_ck, off_ck = si, sj, off
                    self.ck = (D_col[:bs], C_col[:bs], D_row[:bs], R_row[:bs])
                    tr.save_ckpt()
                    grow_no_max = False
                best_max = off_max
                y_drop_iter = 0
            if self.x_drop is not None:
                if off_max == best_max - self.x_drop:
                    self.ties["xdrop_at_threshold"] += 1
                if off_max == best_max - self.x_drop - 1:
                    self.ties["xdrop_one_below"] += 1
                if off_max < best_max - self.x_drop:
                    if x_drop_iter < X_DROP_ITER - 1:
                        x_drop_iter += 1
                    else:
                        self.ties["xdrop_second_step"] += 1
                        break
                else:
                    x_drop_iter = 0
            if si + bs > qn and sj + bs > rn:
                break
            if sj + bs > rn:
                si += STEP; direction = DOWN
                continue
            if si + bs > qn:
                sj += STEP; direction = RIGHT
                continue
            next_size = bs * 2
            if next_size <= self.max_size and y_drop_iter == (bs // STEP) - 1 and not grow_no_max:
                self.ties["grow_at_limit"] += 1
            if next_size <= self.max_size and grow_no_max and direction == GROW and prev_size:
                self.ties["grow_twice"] += 1
            if next_size <= self.max_size and (y_drop_iter > (bs // STEP) - 1 or grow_no_max):
                prev_size, bs, direction = bs, next_size, GROW
                si, sj, off = i_ck, j_ck, off_ck
                for buf, saved in zip((D_col, C_col, D_row, R_row), self.ck):
                    buf[:prev_size] = saved[:prev_size]
                tr.restore_ckpt()
                y_drop_iter = 0
                continue
            if bs > self.min_size and y_drop_iter == 0:
                shrink_max = max(self._suffix_max(D_row, bs), self._suffix_max(D_col, bs))
                if shrink_max == mx:
                    self.ties["shrink_equal"] += 1
                if shrink_max >= mx:
                    prev_dir = GROW
                    bs //= 2
                    for buf in (D_col, C_col, D_row, R_row):
                        buf[:bs] = buf[bs:2 * bs]
                    si += bs; sj += bs
                    i_ck, j_ck, off_ck = si, sj, off
                    self.ck = (D_col[:bs], C_col[:bs], D_row[:bs], R_row[:bs])
                    right_max, down_max = self._prefix_max(D_col), self._prefix_max(D_row)
                    tr.save_ckpt()
                    y_drop_iter = 0
            if down_max == right_max:
                self.ties["dir_equal"] += 1
            if down_max > right_max:
                si += STEP; direction = DOWN
            else:
                sj += STEP; direction = RIGHT
        if self.x_drop is not None:
            self.result = (best_max, best_i, best_j)
        else:
            if direction in (RIGHT, GROW):
                score = off + D_col[qn - si] - ZERO
            else:
                score = off + D_row[rn - sj] - ZERO
            self.result = (score, qn, rn)
        return self.result
