"""The host side of `mchap assemble` over a BLOCK of target loci at once: what application.ReadSource.reads / encode_reads
do locus by locus (reference application/baseclass.py:140-210: extract_read_variants, allele calls, de-duplicated rows with
counts, depths) as array operations over every (locus x read x SNV) cell of the block, ragged -- loci differ in their numbers
of SNVs and reads.  The per-locus functions stay the definition: tests/test_blockpath.py holds the two against each other
cell for cell, and application.assemble falls back to them for inputs this module does not take (BlockPathUnavailable).

Layout: the rows (reads) of all loci in one run, locus after locus (`row_start[l] .. row_start[l + 1]`); a row of locus l has
`M[l]` cells; `cell_of_row[g]` is the flat offset of row g's first cell.
"""
import numpy as np

_BIG = np.int64(1) << 40  # (locus index, position) keys: positions are below 2^40


class BlockPathUnavailable(Exception):
    """The block's inputs are outside what the array path takes (the per-locus path handles them)."""


class BlockPileup:
    """Character / quality matrices of every locus of a block for one sample (what extract_read_variants yields per locus)."""

    __slots__ = ("L", "M", "row_start", "row_locus", "cell_of_row", "cell_start", "chars", "quals", "snv_start")

    def matrices(self, l):
        """(chars uint8 [rows, M], quals int16 [rows, M]) of locus l."""
        a, b = int(self.cell_start[l]), int(self.cell_start[l + 1])
        n = int(self.row_start[l + 1] - self.row_start[l])
        return self.chars[a:b].reshape(n, int(self.M[l])), self.quals[a:b].reshape(n, int(self.M[l]))


def _ragged_arange(counts):
    """(owner index, position within the owner's run) of the concatenation of arange(c) for c in counts."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    owner = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    first = np.repeat(np.cumsum(counts) - counts, counts)
    return owner, np.arange(total, dtype=np.int64) - first


def locus_tables(loci):
    """(M [L], snv_start [L + 1], positions of all loci concatenated) -- positions must ascend within a locus."""
    M = np.fromiter((len(l.positions) for l in loci), dtype=np.int64, count=len(loci))
    snv_start = np.zeros(len(loci) + 1, dtype=np.int64)
    np.cumsum(M, out=snv_start[1:])
    pos = np.fromiter((p for l in loci for p in l.positions), dtype=np.int64, count=int(snv_start[-1]))
    if len(pos) > 1:
        inner = np.ones(len(pos), dtype=bool)
        inner[snv_start[:-1][M > 0]] = False  # (the first position of a locus has no predecessor within it)
        if not (np.diff(pos, prepend=pos[0])[inner] > 0).all():
            raise BlockPathUnavailable("SNV positions of a locus do not ascend")
    return M, snv_start, pos


def extract_block(loci, cols, sample, min_quality=20, skip_duplicates=True, skip_qcfail=True, skip_supplementary=True, tables=None):
    """io.extract_read_variants_columns for every locus of `loci` in one pass over AlignmentColumns `cols` (a coordinate-sorted
    file): the same matrices, rows in order of the first passing record of each query name, mates merged the same way."""
    from .io import _NIB

    L = len(loci)
    M, snv_start, P = tables if tables is not None else locus_tables(loci)
    if not cols.sorted and cols.n:
        raise BlockPathUnavailable("alignment file is not coordinate-sorted")
    out = BlockPileup()
    out.L, out.M, out.snv_start = L, M, snv_start
    names = {n: i for i, (n, _) in enumerate(cols.refs)}
    tid = np.fromiter((names.get(l.contig, -1) for l in loci), dtype=np.int64, count=L)
    start = np.fromiter((l.start for l in loci), dtype=np.int64, count=L)
    stop = np.fromiter((l.stop for l in loci), dtype=np.int64, count=L)
    skip = 0x4 | (0x400 if skip_duplicates else 0) | (0x200 if skip_qcfail else 0) | (0x800 if skip_supplementary else 0)
    if cols.n:
        base = np.where(tid >= 0, tid, 0) << 32
        lo = np.searchsorted(cols.sort_key, base | np.maximum(0, start - cols.max_span), side="left")
        hi = np.searchsorted(cols.sort_key, base | np.maximum(0, stop), side="left")
        hi = np.where(tid >= 0, np.maximum(hi, lo), lo)
    else:
        lo = hi = np.zeros(L, dtype=np.int64)
    pl, k = _ragged_arange(hi - lo)
    pr = lo[pl] + k
    want_rg = np.array([s == sample for s in cols.rg_samples] + [False])  # (index -1: no read group)
    ok = (cols.ref_id[pr] == tid[pl]) & (cols.pos[pr] < stop[pl]) & (cols.end[pr] > start[pl]) & ((cols.flag[pr] & skip) == 0) & \
        (cols.mapq[pr] >= min_quality) & want_rg[cols.rg[pr]]
    pl, pr = pl[ok], pr[ok]
    # rows: the distinct (locus, query name) pairs in order of their first passing record
    nq = int(cols.qname.max()) + 1 if cols.n else 1
    _, first_of, inv = np.unique(pl * nq + cols.qname[pr], return_index=True, return_inverse=True)
    order = np.argsort(first_of, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    pair_row = rank[inv.reshape(-1)]
    n_rows = len(order)
    row_locus = pl[first_of[order]]
    rows_of = np.bincount(row_locus, minlength=L).astype(np.int64)
    row_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(rows_of, out=row_start[1:])
    cell_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(rows_of * M, out=cell_start[1:])
    cell_of_row = cell_start[row_locus] + (np.arange(n_rows, dtype=np.int64) - row_start[row_locus]) * M[row_locus]
    out.row_start, out.row_locus, out.cell_of_row, out.cell_start = row_start, row_locus, cell_of_row, cell_start
    chars = np.full(int(cell_start[-1]), ord("-"), dtype=np.uint8)
    quals = np.zeros(int(cell_start[-1]), dtype=np.int64)
    if len(pr) and len(P):
        # the aligned segments (M = X) of the passing records, then the SNVs of the record's locus each one covers
        sp, k = _ragged_arange(cols.seg_first[pr + 1] - cols.seg_first[pr])
        seg = cols.seg_first[pr][sp] + k
        op = cols.c_op[seg]
        keep = (op == 0) | (op == 7) | (op == 8)
        sp, seg = sp[keep], seg[keep]
        sl = pl[sp]
        r0 = cols.c_ref0[seg]
        key = np.repeat(np.arange(L, dtype=np.int64), M) * _BIG + P
        a = np.searchsorted(key, sl * _BIG + r0, side="left")
        b = np.searchsorted(key, sl * _BIG + r0 + cols.c_len[seg], side="left")
        hs, k = _ragged_arange(b - a)
        snv = a[hs] + k                              # index into the concatenated positions
        s_ = seg[hs]
        rec = cols.c_rec[s_]
        ro = cols.c_read0[s_] + (P[snv] - cols.c_ref0[s_])
        byte = cols.buf[cols.seq_off[rec] + (ro >> 1)]
        base = _NIB[np.where(ro & 1, byte & 15, byte >> 4)]
        q = cols.buf[cols.qual_off[rec] + ro].astype(np.int64)
        cell = cell_of_row[pair_row[sp[hs]]] + (snv - snv_start[sl[hs]])
        # a cell hit more than once (overlapping mates): the hits applied in record order
        o = np.argsort(cell, kind="stable")
        sc = cell[o]
        if len(sc) < 2 or (sc[1:] != sc[:-1]).all():
            chars[cell] = base
            quals[cell] = q
        else:
            rank_c = np.zeros(len(cell), dtype=np.int64)
            startg = np.r_[True, sc[1:] != sc[:-1]]
            rank_c[o] = np.arange(len(sc)) - np.maximum.accumulate(np.where(startg, np.arange(len(sc)), 0))
            for kk in range(int(rank_c.max(initial=-1)) + 1):
                m = rank_c == kk
                c_, b_, q_ = cell[m], base[m], q[m]
                cur = chars[c_]
                empty = cur == ord("-")
                same = cur == b_
                chars[c_] = np.where(empty, b_, np.where(same, cur, ord("N")))
                quals[c_] = np.where(empty, q_, np.where(same, quals[c_] + q_, quals[c_]))
    out.chars, out.quals = chars, quals.astype(np.int16)
    return out


def pile_from_matrices(loci, matrices, tables=None):
    """A BlockPileup from character / quality matrices the caller has already (application.MatrixSource): matrices[l] =
    (chars [rows, M] as str or uint8 ASCII codes, quals [rows, M]) of locus l."""
    L = len(loci)
    M, snv_start, _ = tables if tables is not None else locus_tables(loci)
    out = BlockPileup()
    out.L, out.M, out.snv_start = L, M, snv_start
    cs, qs = [], []
    rows_of = np.zeros(L, dtype=np.int64)
    for l, (chars, quals) in enumerate(matrices):
        chars = np.asarray(chars)
        if chars.dtype != np.uint8:
            chars = chars.astype("S1").view(np.uint8).reshape(chars.shape) if chars.size else np.empty(chars.shape, dtype=np.uint8)
        assert chars.ndim == 2 and chars.shape[1] == M[l]
        rows_of[l] = chars.shape[0]
        cs.append(chars.reshape(-1))
        qs.append(np.asarray(quals, dtype=np.int16).reshape(-1))
    out.row_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(rows_of, out=out.row_start[1:])
    out.cell_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(rows_of * M, out=out.cell_start[1:])
    out.row_locus = np.repeat(np.arange(L, dtype=np.int64), rows_of)
    out.cell_of_row = out.cell_start[out.row_locus] + (np.arange(int(out.row_start[-1]), dtype=np.int64) - out.row_start[out.row_locus]) * M[out.row_locus]
    out.chars = np.concatenate(cs) if cs else np.zeros(0, dtype=np.uint8)
    out.quals = np.concatenate(qs) if qs else np.zeros(0, dtype=np.int16)
    return out


_MULT = (np.random.default_rng(0x5eed).integers(1, 1 << 62, size=256, dtype=np.int64).astype(np.uint64) << np.uint64(1)) | np.uint64(1)


class BlockEncoding:
    """encode_reads for every locus of a block: allele calls of all cells, depths, and per locus the distinct rows of calls in
    order of first appearance with their counts."""

    __slots__ = ("pile", "calls", "depth", "rcount", "rcalls", "dp", "urow_start", "ucell_start", "ucalls", "ucounts", "urow_locus",
                 "ucell_of_row")

    def per_locus(self, l):
        """dict(chars, calls, depth, ucalls, counts) of locus l, shaped like application.encode_reads's."""
        p = self.pile
        chars, _ = p.matrices(l)
        m = int(p.M[l])
        a, b = int(p.cell_start[l]), int(p.cell_start[l + 1])
        ua, ub = int(self.ucell_start[l]), int(self.ucell_start[l + 1])
        nu = int(self.urow_start[l + 1] - self.urow_start[l])
        return dict(chars=chars, calls=self.calls[a:b].reshape(chars.shape), depth=self.depth[int(p.snv_start[l]):int(p.snv_start[l + 1])],
                    ucalls=self.ucalls[ua:ub].reshape(nu, m), counts=self.ucounts[int(self.urow_start[l]):int(self.urow_start[l + 1])])


def allele_lut(loci, n_snv):
    """int8 [SNVs of all loci, 256]: allele index of every character at every SNV (the first allele that has it; -1: none)."""
    lut = np.full((max(n_snv, 1), 256), -1, dtype=np.int8)
    si, ai, ci = [], [], []
    j = 0
    for l in loci:
        for tup in l.alleles:
            for a, c in enumerate(tup):
                si.append(j)
                ai.append(a)
                ci.append(ord(c))
            j += 1
    if si:
        si, ai, ci = np.array(si, dtype=np.int64), np.array(ai, dtype=np.int8), np.array(ci, dtype=np.int64)
        _, first = np.unique(si * 256 + ci, return_index=True)  # (an allele character listed twice: its first index)
        lut[si[first], ci[first]] = ai[first]
    return lut


def encode_block(loci, pile, lut=None):
    """Allele index of every character (the first allele of the SNV that has it, else -1), depth per SNV (characters that are
    not '-'), reads and called cells per locus, DP (the rounded mean depth), distinct call rows with counts.  lut: allele_lut of
    the loci (shared by the samples of a block)."""
    L, M = pile.L, pile.M
    n_snv = int(pile.snv_start[-1])
    enc = BlockEncoding()
    enc.pile = pile
    if lut is None:
        lut = allele_lut(loci, n_snv)
    n_rows = len(pile.row_locus)
    row_M = M[pile.row_locus]
    row_of_cell, jj = _ragged_arange(row_M)
    cell_locus = pile.row_locus[row_of_cell]
    cell_snv = pile.snv_start[cell_locus] + jj
    calls = lut[cell_snv, pile.chars] if len(pile.chars) else np.zeros(0, dtype=np.int8)
    enc.calls = calls
    enc.depth = np.bincount(cell_snv, weights=(pile.chars != ord("-")), minlength=n_snv).astype(np.int64)[:n_snv]
    enc.rcount = (pile.row_start[1:] - pile.row_start[:-1]).astype(np.int64)
    enc.rcalls = np.bincount(cell_locus, weights=(calls >= 0), minlength=L).astype(np.int64)
    sums = np.bincount(np.repeat(np.arange(L), M), weights=enc.depth, minlength=L)
    with np.errstate(invalid="ignore", divide="ignore"):
        enc.dp = np.where(M > 0, np.round(sums / np.maximum(M, 1)), np.nan)
    # distinct rows per locus: rows sorted by (locus, a 64-bit hash of the row), equal neighbours grouped, every row then
    # compared cell by cell with the first row of its group (a hash collision would show there: none has been seen)
    rows = np.flatnonzero(row_M > 0)  # (rows of loci without SNVs have no cells; such loci are not sampled)
    if len(rows):
        h_cell = (calls.astype(np.int64) + 2).astype(np.uint64) * _MULT[jj % 256] + _MULT[(jj // 256) % 256] * (jj // 256).astype(np.uint64)
        h = np.add.reduceat(h_cell, pile.cell_of_row[rows])
        o = np.lexsort((h, pile.row_locus[rows]))
        hs, ls = h[o], pile.row_locus[rows][o]
        new = np.r_[True, (hs[1:] != hs[:-1]) | (ls[1:] != ls[:-1])]
        gid_sorted = np.cumsum(new) - 1
        first_sorted = np.flatnonzero(new)
        rep = rows[o[first_sorted]]                    # the first row (lowest index: the sort is stable) of each group
        gid = np.empty(len(rows), dtype=np.int64)
        gid[o] = gid_sorted
        counts = np.bincount(gid, minlength=len(rep)).astype(np.int64)
        rep_of_row = np.full(n_rows, -1, dtype=np.int64)
        rep_of_row[rows] = rep[gid]
        same = calls == calls[pile.cell_of_row[rep_of_row[row_of_cell]] + jj]
        if not same.all():
            raise BlockPathUnavailable("row hash collision")
        order = np.argsort(rep, kind="stable")         # in order of first appearance, locus after locus
        urows, ucounts = rep[order], counts[order]
    else:
        urows, ucounts = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    enc.urow_locus = pile.row_locus[urows]
    nu = np.bincount(enc.urow_locus, minlength=L).astype(np.int64)
    enc.urow_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(nu, out=enc.urow_start[1:])
    enc.ucell_start = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(nu * M, out=enc.ucell_start[1:])
    ur, uj = _ragged_arange(M[enc.urow_locus])
    enc.ucalls = calls[pile.cell_of_row[urows][ur] + uj] if len(urows) else np.zeros(0, dtype=np.int8)
    enc.ucell_of_row = enc.ucell_start[enc.urow_locus] + (np.arange(len(urows), dtype=np.int64) - enc.urow_start[enc.urow_locus]) * M[enc.urow_locus]
    enc.ucounts = ucounts
    return enc


def unit_inputs(enc, use):
    """The sampler's compact input for the loci `use` (indices, each with at least one SNV) of one sample's encoding: int8 calls
    of the distinct rows, unit after unit, their counts, and per unit (rows, first call element, first count or -1).  A locus
    without reads is one all-gap row without counts (assemble/mcmc.py:132-137)."""
    M = enc.pile.M[use]
    nu = (enc.urow_start[1:] - enc.urow_start[:-1])[use]
    R = np.maximum(nu, 1)
    cells = R * M
    off = np.cumsum(cells) - cells
    calls = np.full(int(cells.sum()), -1, dtype=np.int8)
    src_n = nu * M
    owner, k = _ragged_arange(src_n)
    calls[off[owner] + k] = enc.ucalls[enc.ucell_start[use][owner] + k]
    c_off = np.cumsum(nu) - nu
    owner, k = _ragged_arange(nu)
    counts = enc.ucounts[enc.urow_start[use][owner] + k]
    return calls, counts, R, off, np.where(nu > 0, c_off, -1)


def unpack_words(words, unit, fixed, fixed_off, M, bits):
    """Packed haplotype words -> allele rows, for words of many units at once: words uint64 [n], unit [n] (index of each
    word's unit), fixed int8 (the batch's fixed-allele templates, unit u at fixed_off[u] .. + M[u]), bits [U] bits per sampled
    position.  Returns (alleles int8 flat, first cell of every word): assemble.unpack_trace, ragged."""
    words = np.asarray(words, dtype=np.uint64)
    wide = words.ndim == 2   # [n, 2]: haplotypes of a batch of the general sampler
    n = len(words)
    Mw = M[unit]
    cell_of = np.cumsum(Mw) - Mw
    w, j = _ragged_arange(Mw)
    u = unit[w]
    fx = fixed[fixed_off[u] + j]
    het = fx < 0
    # a sampled position's field: the jj-th of the unit's mh sampled positions sits bits * (mh - 1 - jj) bits up
    csum = np.cumsum(het)
    before = csum - het                                   # sampled cells before this one (over the whole run)
    row_first = before[cell_of][w] if n else before       # ... before the word's first cell
    jj = before - row_first
    mh = (np.add.reduceat(het.astype(np.int64), cell_of[Mw > 0]) if (Mw > 0).any() else np.zeros(0, dtype=np.int64))
    mh_w = np.zeros(n, dtype=np.int64)
    mh_w[Mw > 0] = mh
    sh = (bits[u] * (mh_w[w] - 1 - jj)).clip(0).astype(np.uint64)
    mask = (np.uint64(1) << bits[u].astype(np.uint64)) - np.uint64(1)
    if wide:
        # two words per haplotype, the more significant first (assemble.unpack_trace(..., 2)): a field lies in the low word, in
        # the high one, or across both
        hi, lo = words[w, 0], words[w, 1]
        low = sh < 64
        s_lo = np.where(low, sh, 0).astype(np.uint64)
        across = low & (sh > 0)
        up = np.where(across, np.uint64(64) - s_lo, 0).astype(np.uint64)
        v_low = (lo >> s_lo) | np.where(across, hi << up, np.uint64(0))
        v_high = hi >> np.where(low, 0, sh - np.uint64(64)).astype(np.uint64)
        val = (np.where(low, v_low, v_high) & mask).astype(np.int8)
    else:
        val = ((words[w] >> sh) & mask).astype(np.int8)
    return np.where(het, val, fx).astype(np.int8), cell_of
