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
