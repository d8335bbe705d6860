"""Synthetic 3Di/AA structure databases of the shape BASELINE.json's configs describe.

Lengths ~ gamma(mean 350) clipped to [30, 2000]; 3Di letters i.i.d. from the 3Di background, AA letters from
the BLOSUM62 background; a fraction of soft-masked residues; for every query a number of planted homologs
(point substitutions + short indels) so hit lists are non-trivial.  Pure numpy, fixed seeds.

The database is produced directly in the reference's *padded GPU layout* (makepaddedseqdb,
M/src/util/makepaddedseqdb.cpp:59-109): entries sorted by ascending length, codes 0..20 (+32 if masked),
each entry padded with code 20 to a multiple of 4, index length field = L + 2, keys renumbered 0..N-1.
"""
import numpy as np

BACK_3DI = np.array([0.0489372, 0.0306991, 0.101049, 0.0329671, 0.0276149, 0.0416262, 0.0452521, 0.030876,
                     0.0297251, 0.0607036, 0.0150238, 0.0215826, 0.0783843, 0.0512926, 0.0264886, 0.0610702,
                     0.0201311, 0.215998, 0.0310265, 0.0295417])
BACK_AA = np.array([0.07422, 0.02469, 0.05363, 0.05431, 0.04742, 0.07415, 0.02621, 0.06792, 0.05815, 0.09891,
                    0.02499, 0.04465, 0.03854, 0.03426, 0.05161, 0.05723, 0.05089, 0.07292, 0.01303, 0.03228])
ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"


class PaddedDB:
    """Target DB in padded layout: `data3di`/`dataaa` byte buffers, `offsets` (int64, n+1), `lengths` (int32)."""

    def __init__(self, data3di, dataaa, offsets, lengths):
        self.data3di, self.dataaa, self.offsets, self.lengths = data3di, dataaa, offsets, lengths
        self.n = len(lengths)
        self.residues = int(lengths.sum())

    def seq(self, i, which="3di", unmask=True):
        d = self.data3di if which == "3di" else self.dataaa
        s = d[self.offsets[i]:self.offsets[i] + self.lengths[i]]
        return np.where(s >= 32, s - 32, s).astype(np.uint8) if unmask else s


def _lengths(rng, n, mean=350.0, lo=30, hi=2000):
    shape = 2.2
    l = rng.gamma(shape, mean / shape, size=n)
    return np.clip(np.rint(l), lo, hi).astype(np.int32)


def _mutate(rng, s3, sa, sub_rate, indel_rate):
    """point substitutions + geometric indels applied jointly to the 3Di and AA strings"""
    L = len(s3)
    keep = rng.random(L) >= indel_rate / 2          # deletions
    s3, sa = s3[keep], sa[keep]
    L = len(s3)
    sub = rng.random(L) < sub_rate
    s3 = np.where(sub, rng.choice(20, size=L, p=BACK_3DI / BACK_3DI.sum()), s3).astype(np.uint8)
    sub2 = rng.random(L) < sub_rate
    sa = np.where(sub2, rng.choice(20, size=L, p=BACK_AA / BACK_AA.sum()), sa).astype(np.uint8)
    nins = rng.binomial(L, indel_rate / 6)
    if nins:
        pos = np.sort(rng.integers(0, L + 1, size=nins))
        lens = rng.geometric(1 / 3.0, size=nins)
        p3, pa, prev = [], [], 0
        for p, k in zip(pos, lens):
            p3 += [s3[prev:p], rng.choice(20, size=k, p=BACK_3DI / BACK_3DI.sum()).astype(np.uint8)]
            pa += [sa[prev:p], rng.choice(20, size=k, p=BACK_AA / BACK_AA.sum()).astype(np.uint8)]
            prev = p
        p3.append(s3[prev:]); pa.append(sa[prev:])
        s3, sa = np.concatenate(p3), np.concatenate(pa)
    return s3, sa


def make_queries(nq, seed=1, mean_len=350.0, lo=30, hi=2000):
    rng = np.random.default_rng(seed)
    lens = _lengths(rng, nq, mean_len, lo, hi)
    q3 = [rng.choice(20, size=l, p=BACK_3DI / BACK_3DI.sum()).astype(np.uint8) for l in lens]
    qa = [rng.choice(20, size=l, p=BACK_AA / BACK_AA.sum()).astype(np.uint8) for l in lens]
    return q3, qa


def sticky(rng, seq, stay):
    """Markov 'stay' chain over an i.i.d. string: with probability `stay` a residue repeats its predecessor -- runs and
    low-complexity stretches like real 3Di strings (helix = long V/L runs, strand = D runs)."""
    keep = rng.random(len(seq)) < stay
    keep[0] = False
    idx = np.where(~keep, np.arange(len(seq)), 0)
    np.maximum.accumulate(idx, out=idx)
    return seq[idx]


def make_db(n, queries=None, seed=20260923, homologs_per_query=50, mask_frac=0.01, mean_len=350.0, lo=30, hi=2000,
            x_frac=0.002, stay=0.0):
    """Returns PaddedDB. `queries` = (q3, qa) lists from make_queries to plant homologs of.
    stay > 0: 3Di letters repeat their predecessor with that probability (runs / low complexity)."""
    rng = np.random.default_rng(seed)
    lens = _lengths(rng, n, mean_len, lo, hi)
    total = int(lens.sum())
    flat3 = rng.choice(20, size=total, p=BACK_3DI / BACK_3DI.sum()).astype(np.uint8)
    if stay > 0:
        flat3 = sticky(rng, flat3, stay)
    flata = rng.choice(20, size=total, p=BACK_AA / BACK_AA.sum()).astype(np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    seqs3 = [flat3[s:s + l] for s, l in zip(starts, lens)]
    seqsa = [flata[s:s + l] for s, l in zip(starts, lens)]
    if queries is not None:
        q3, qa = queries
        slots = rng.permutation(n)
        k = 0
        for qi in range(len(q3)):
            for h in range(homologs_per_query):
                if k >= n:
                    break
                rate = 0.2 + 0.4 * (h / max(1, homologs_per_query - 1))
                s3, sa = _mutate(rng, q3[qi], qa[qi], rate, 0.05)
                if len(s3) < lo:
                    continue
                seqs3[slots[k]], seqsa[slots[k]] = s3[:hi], sa[:hi]
                k += 1
    lens = np.array([len(s) for s in seqs3], dtype=np.int32)
    order = np.argsort(lens, kind="stable")          # ascending length, ties by original id
    lens = lens[order]
    padded = (lens + 3) // 4 * 4
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(padded)
    d3 = np.full(int(offsets[-1]), 20, dtype=np.uint8)
    da = np.full(int(offsets[-1]), 20, dtype=np.uint8)
    for new, old in enumerate(order):
        o, l = offsets[new], lens[new]
        d3[o:o + l] = seqs3[old]
        da[o:o + l] = seqsa[old]
    # sprinkle X (20) and soft-masked (+32) residues, jointly in both strings like lower-case masking would
    pos = np.flatnonzero(d3 < 20)
    if x_frac > 0:
        xs = rng.choice(pos, size=int(len(pos) * x_frac), replace=False)
        d3[xs] = 20
    if mask_frac > 0:
        nrun = max(1, int(len(pos) * mask_frac / 8))
        st = rng.choice(pos, size=nrun, replace=False)
        for s in st:
            e = min(s + 8, len(d3))
            seg = slice(s, e)
            m = d3[seg] < 32
            # never mask padding bytes: padding is code 20 beyond the entry length; entries also hold real X=20,
            # so restrict to positions inside an entry
            idx = np.searchsorted(offsets, np.arange(s, e), side="right") - 1
            inside = (np.arange(s, e) - offsets[idx]) < lens[np.minimum(idx, n - 1)]
            mm = m & inside
            d3[seg] = np.where(mm, d3[seg] + 32, d3[seg])
    return PaddedDB(d3, da, offsets, lens)


# ---- vectorised generator for the full-size bench databases (1M targets: 350 M residues in a few seconds) ---------
def _draw(rng, n, back):
    """n i.i.d. letters from `back` through a 16-bit inverse-CDF table (one integer draw + one gather per letter)"""
    cdf = np.cumsum(back / back.sum())
    lut = np.minimum(np.searchsorted(cdf, (np.arange(65536) + 0.5) / 65536.0), len(back) - 1).astype(np.uint8)
    out = np.empty(n, np.uint8)
    step = 1 << 26
    for a in range(0, n, step):                       # chunked: the uint16 temporaries stay small
        m = min(step, n - a)
        out[a:a + m] = lut[rng.integers(0, 65536, size=m, dtype=np.uint16)]
    return out


def _homologs(rng, q3, qa, H, lo, hi, indel_rate=0.05):
    """H mutated copies of one query at once: substitution rate 20 % ... 60 % (drawn from the backgrounds), deletions
    (indel_rate / 2 per residue) and insertions (indel_rate / 6 per residue, geometric length, mean 3); same model as
    _mutate.  Returns (flat 3Di, flat AA, lengths)."""
    L = len(q3)
    rate = (0.2 + 0.4 * np.arange(H) / max(1, H - 1))[:, None]
    keep = rng.random((H, L)) >= indel_rate / 2
    s3 = np.where(rng.random((H, L)) < rate, _draw(rng, H * L, BACK_3DI).reshape(H, L), q3[None, :])
    sa = np.where(rng.random((H, L)) < rate, _draw(rng, H * L, BACK_AA).reshape(H, L), qa[None, :])
    lens = keep.sum(1)
    f3, fa = s3[keep], sa[keep]                        # row-major concatenation of the kept residues
    ins = rng.random(len(f3)) < indel_rate / 6
    cnt = np.ones(len(f3), np.int64)
    cnt[ins] += rng.geometric(1 / 3.0, size=int(ins.sum()))
    row = np.repeat(np.arange(H), lens)
    lens = np.bincount(row, weights=cnt, minlength=H).astype(np.int64)
    first = np.zeros(int(cnt.sum()), bool)
    first[np.concatenate([[0], np.cumsum(cnt)[:-1]])] = True
    f3, fa = np.repeat(f3, cnt), np.repeat(fa, cnt)
    nc = int((~first).sum())
    f3[~first] = _draw(rng, nc, BACK_3DI)              # the extra copies become the inserted letters
    fa[~first] = _draw(rng, nc, BACK_AA)
    # cut to [.., hi] and drop copies shorter than lo
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    out3, outa, outl = [], [], []
    for h in range(H):
        l = int(min(lens[h], hi))
        if l < lo:
            continue
        out3.append(f3[starts[h]:starts[h] + l]); outa.append(fa[starts[h]:starts[h] + l]); outl.append(l)
    return out3, outa, outl


def make_db_fast(n, queries=None, seed=20260923, homologs_per_query=50, mask_frac=0.01, mean_len=350.0, lo=30, hi=2000,
                 x_frac=0.002):
    """Same distributional model as make_db without per-entry Python work: lengths are drawn first and
    sorted, the padded buffers are filled with i.i.d. letters in one go, the planted homologs (homologs_per_query for
    EVERY query given) overwrite their slots.  Not bit-compatible with make_db (different draw order)."""
    rng = np.random.default_rng(seed)
    h3, ha, hl = [], [], []
    if queries is not None and homologs_per_query > 0:
        for q3, qa in zip(*queries):
            a, b, c = _homologs(rng, q3, qa, homologs_per_query, lo, hi)
            h3 += a; ha += b; hl += c
        if len(hl) > n:
            h3, ha, hl = h3[:n], ha[:n], hl[:n]
    nbg = n - len(hl)
    lens = np.concatenate([_lengths(rng, nbg, mean_len, lo, hi), np.array(hl, np.int32)]).astype(np.int32)
    order = np.argsort(lens, kind="stable")
    lens = lens[order]
    padded = (lens.astype(np.int64) + 3) // 4 * 4
    offsets = np.zeros(n + 1, np.int64)
    offsets[1:] = np.cumsum(padded)
    total = int(offsets[-1])
    d3, da = _draw(rng, total, BACK_3DI), _draw(rng, total, BACK_AA)
    slot = np.empty(n, np.int64)
    slot[order] = np.arange(n)                          # old entry number -> position in the length order
    for k in range(len(hl)):
        o = offsets[slot[nbg + k]]
        d3[o:o + hl[k]] = h3[k]; da[o:o + hl[k]] = ha[k]
    if x_frac > 0:
        xs = rng.integers(0, total, size=int(total * x_frac))
        d3[xs] = 20
    if mask_frac > 0:
        st = rng.integers(0, total, size=max(1, int(total * mask_frac / 8)))
        idx = np.minimum(st[:, None] + np.arange(8)[None, :], total - 1).ravel()
        e = np.searchsorted(offsets, idx, side="right") - 1
        idx = np.unique(idx[(idx - offsets[e]) < lens[e]])
        idx = idx[d3[idx] < 32]
        d3[idx] += 32
    for k in range(1, 4):                               # padding bytes (<= 3 per entry) are code 20 in both buffers
        p = offsets[:-1] + lens + (k - 1)
        p = p[p < offsets[1:]]
        d3[p] = 20; da[p] = 20
    return PaddedDB(d3, da, offsets, lens)
