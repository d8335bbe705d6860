"""ctypes bindings of the TEST-ONLY k-mer prefilter checkers: oracle/libfso.so (fs_kmer_oracle.c, the C
restatement) and oracle/_ref/libfsref.so (ref_kmer_harness.cpp around the reference's own classes)."""
import ctypes as C
import os
import numpy as np

import oracle_lib

ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"
_A = np.frombuffer(ALPHABET.encode(), np.uint8)
HIT_DT = np.dtype([("id", np.uint32), ("score", np.int32), ("diag", np.uint16), ("pad", np.uint16)])


_COMMON_FIELDS = [(n, C.c_int32) for n in
                  "kmerSize spaced kmerThr maxResListLen compBias minDiagScoreThr maskLowerCase maskNrepeats".split()] + \
                 [("compBiasScale", C.c_float), ("bins", C.c_int32), ("maxDbMatches", C.c_int64),
                  ("foundDiagonalsSize", C.c_int64)]


class RefParams(C.Structure):
    # noDiagScore = 1: QueryMatcher built with diagonalScoring = false (--diag-score 0: the k-mer match count is the score)
    _fields_ = _COMMON_FIELDS + [("noDiagScore", C.c_int32), ("pad", C.c_int32)]


class OraParams(C.Structure):
    _fields_ = _COMMON_FIELDS + [("l2CacheSize", C.c_uint64), ("noDiagScore", C.c_int32), ("pad", C.c_int32)]


def default_params(cls=OraParams, **kw):
    d = dict(kmerSize=6, spaced=1, kmerThr=78, maxResListLen=1000, compBias=1, minDiagScoreThr=30, maskLowerCase=1,
             maskNrepeats=6, compBiasScale=0.15, bins=0, maxDbMatches=0, foundDiagonalsSize=0)
    if cls is OraParams:
        d["l2CacheSize"] = 2 * 1024 * 1024
    d.update(kw)
    if cls is not OraParams:
        d.pop("l2CacheSize", None)
    return cls(**d)


def to_ascii(codes):
    """numeric codes (0..20, +32 = soft-masked) -> ASCII bytes (lower case = masked)"""
    codes = np.asarray(codes, np.uint8)
    m = codes >= 32
    a = _A[np.where(m, codes - 32, codes)]
    return np.where(m, a + 32, a).astype(np.uint8)


def flatten(seqs):
    lens = np.array([len(s) for s in seqs], np.int32)
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    cat = np.concatenate(seqs).astype(np.uint8) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(cat), off, lens


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def load_ref():
    p = oracle_lib.ref_path()
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    if not hasattr(L, "ref_kpf_create"):
        return None
    L.ref_kpf_create.restype = C.c_void_p
    L.ref_kpf_run.restype = C.c_double
    L.ref_l2_cache_size.restype = C.c_uint64
    for f in ("ref_kpf_index_entries", "ref_kpf_index_list", "ref_kpf_index_offsets", "ref_kpf_kmer_list",
              "ref_kpf_scorematrix_row"):
        getattr(L, f).restype = C.c_int64
    return L


def load_ora():
    L = oracle_lib.load_oracle()
    L.fko_create.restype = C.c_void_p
    for f in ("fko_index_entries", "fko_index_list", "fko_index_offsets", "fko_kmer_list", "fko_scorematrix_row"):
        getattr(L, f).restype = C.c_int64
    return L


class RefKpf:
    """reference k-mer prefilter over a list of numeric target sequences"""

    def __init__(self, L, targets, threads=8, **kw):
        self.L, self.p = L, default_params(RefParams, **kw)
        cat, off, lens = flatten([to_ascii(t) for t in targets])
        self.n = len(targets)
        self.h = C.c_void_p(L.ref_kpf_create(C.byref(self.p), _vp(cat), _vp(off), _vp(lens), C.c_int64(self.n), threads))

    @classmethod
    def from_padded(cls, L, db, threads=8, **kw):
        """the same over a padded database (synth.PaddedDB) without per-target Python objects: the whole 3Di buffer becomes ASCII in one go, the harness
        reads the sequences through the database's own offsets / lengths (the padding between them is never looked at)"""
        self = cls.__new__(cls)
        self.L, self.p = L, default_params(RefParams, **kw)
        cat = np.ascontiguousarray(to_ascii(db.data3di))
        off = np.ascontiguousarray(db.offsets, np.int64)
        lens = np.ascontiguousarray(db.lengths, np.int32)
        self.n = int(db.n)
        self.h = C.c_void_p(L.ref_kpf_create(C.byref(self.p), _vp(cat), _vp(off), _vp(lens), C.c_int64(self.n), threads))
        return self

    def close(self):
        if self.h:
            self.L.ref_kpf_free(self.h)
            self.h = None

    def set(self, **kw):
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.L.ref_kpf_set_params(self.h, C.byref(self.p))

    def run(self, queries, identity=None, threads=1):
        nq = len(queries)
        cat, off, lens = flatten([to_ascii(np.asarray(q) % 32) for q in queries])
        ident = np.full(nq, -1, np.int64) if identity is None else np.asarray(identity, np.int64)
        cap = self.p.maxResListLen
        out = np.zeros((nq, cap), HIT_DT)
        cnt = np.zeros(nq, np.int32)
        stats = np.zeros((nq, 4))
        secs = self.L.ref_kpf_run(self.h, _vp(cat), _vp(off), _vp(lens), C.c_int64(nq), _vp(ident), threads, _vp(out), _vp(cnt), _vp(stats))
        return [out[q, :cnt[q]].copy() for q in range(nq)], stats, secs

    def kmer_list(self, kmer, thr, cap=1 << 23):
        out = np.zeros(cap, np.uint64)
        n = self.L.ref_kpf_kmer_list(self.h, _vp(np.ascontiguousarray(kmer, np.uint8)), int(thr), _vp(out), C.c_int64(cap))
        return out[:n].copy()

    def index_list(self, kmer, cap=1 << 20):
        s = np.zeros(cap, np.uint32); p = np.zeros(cap, np.uint16)
        n = self.L.ref_kpf_index_list(self.h, C.c_int64(int(kmer)), _vp(s), _vp(p), C.c_int64(cap))
        return s[:n].copy(), p[:n].copy()

    def offsets(self):
        ts = self.L.ref_kpf_index_offsets(self.h, None, C.c_int64(0))
        out = np.zeros(ts + 1, np.uint64)
        self.L.ref_kpf_index_offsets(self.h, _vp(out), C.c_int64(ts + 1))
        return out

    def masked(self, i, L):
        out = np.zeros(L, np.uint8)
        self.L.ref_kpf_masked(self.h, C.c_int64(i), _vp(out))
        return out

    def row(self, which, idx):
        size = 20 ** which
        s = np.zeros(size, np.int16); ix = np.zeros(size, np.uint32)
        self.L.ref_kpf_scorematrix_row(self.h, which, C.c_int64(idx), _vp(s), _vp(ix))
        return s, ix

    def submat(self, which):
        m = np.zeros((21, 21), np.int16)
        self.L.ref_kpf_submat(self.h, which, _vp(m))
        return m


class OraKpf:
    """C restatement (oracle) of the k-mer prefilter over a list of numeric target sequences"""

    def __init__(self, L, kmer_sub, pback, ung_sub, targets, **kw):
        self.L, self.p = L, default_params(OraParams, **kw)
        cat, off, lens = flatten([np.asarray(t, np.uint8) for t in targets])
        self.n = len(targets)
        ks = np.ascontiguousarray(kmer_sub, np.int16); us = np.ascontiguousarray(ung_sub, np.int16)
        pb = np.ascontiguousarray(pback, np.float64)
        self.h = C.c_void_p(L.fko_create(C.byref(self.p), _vp(ks), _vp(pb), _vp(us), _vp(cat), _vp(off), _vp(lens), C.c_int64(self.n)))

    def close(self):
        if self.h:
            self.L.fko_free(self.h)
            self.h = None

    def set(self, **kw):
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.L.fko_set_params(self.h, C.byref(self.p))

    def query(self, q, identity=-1):
        out = np.zeros(max(1, self.p.maxResListLen), HIT_DT)
        stats = np.zeros(4)
        qq = np.ascontiguousarray(q, np.uint8)
        n = self.L.fko_query(self.h, _vp(qq), len(qq), C.c_int64(identity), _vp(out), _vp(stats))
        self.last_rc = int(n)        # -1: the reference's std::sort branch (not modelled)
        return (out[:max(n, 0)].copy() if n >= 0 else None), stats

    def run(self, queries, identity=None):
        res, st = [], []
        for i, q in enumerate(queries):
            r, s = self.query(q, -1 if identity is None else int(identity[i]))
            res.append(r); st.append(s)
        return res, np.array(st)

    def kmer_list(self, kmer, thr, cap=1 << 23):
        out = np.zeros(cap, np.uint64)
        n = self.L.fko_kmer_list(self.h, _vp(np.ascontiguousarray(kmer, np.uint8)), int(thr), _vp(out), C.c_int64(cap))
        return out[:n].copy()

    def index_list(self, kmer, cap=1 << 20):
        s = np.zeros(cap, np.uint32); p = np.zeros(cap, np.uint16)
        n = self.L.fko_index_list(self.h, C.c_int64(int(kmer)), _vp(s), _vp(p), C.c_int64(cap))
        return s[:n].copy(), p[:n].copy()

    def offsets(self):
        ts = self.L.fko_index_offsets(self.h, None, C.c_int64(0))
        out = np.zeros(ts + 1, np.uint64)
        self.L.fko_index_offsets(self.h, _vp(out), C.c_int64(ts + 1))
        return out

    def index(self):
        off = self.offsets()
        ne = int(off[-1])
        s = np.zeros(ne, np.uint32); p = np.zeros(ne, np.uint16)
        self.L.fko_index_copy(self.h, _vp(off), _vp(s), _vp(p))
        return off, s, p

    def masked(self, i, L):
        out = np.zeros(L, np.uint8)
        self.L.fko_masked(self.h, C.c_int64(i), _vp(out))
        return out

    def row(self, which, idx):
        size = 20 ** which
        s = np.zeros(size, np.int16); ix = np.zeros(size, np.uint32)
        self.L.fko_scorematrix_row(self.h, which, C.c_int64(idx), _vp(s), _vp(ix))
        return s, ix
