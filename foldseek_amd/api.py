"""ctypes binding of libfsgpu.so (include/fsgpu.h + include/fshost.h) used by bench.py and the tests.

This is plumbing, not the product: every call goes through the C ABI of the shared library (HIP kernels + C++ host
code).  There is NO CPU fallback -- if the library or a GPU is missing the calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsgpu.so")

HIT_DT = np.dtype([("id", np.uint32), ("score", np.int32)])
SWRES_DT = np.dtype([("score", np.int32), ("qEnd", np.int32), ("dbEnd", np.int32), ("word", np.int32)])
RESULT_DT = np.dtype([("dbKey", np.uint32), ("score", np.int32), ("qcov", np.float32), ("dbcov", np.float32),
                      ("seqId", np.float32), ("_pad0", np.uint32), ("eval", np.float64), ("alnLength", np.uint32),
                      ("qStartPos", np.int32), ("qEndPos", np.int32), ("qLen", np.uint32), ("dbStartPos", np.int32),
                      ("dbEndPos", np.int32), ("dbLen", np.uint32), ("backtraceOff", np.uint32),
                      ("backtraceLen", np.uint32), ("_pad1", np.uint32)])


class Params(C.Structure):
    _fields_ = [("maxResListLen", C.c_int), ("minDiagScoreThr", C.c_int), ("compBiasCorrection", C.c_int),
                ("prefCompBiasScale", C.c_float), ("alignmentType", C.c_int), ("alnCompBiasScale", C.c_float),
                ("gapOpen", C.c_int), ("gapExtend", C.c_int), ("evalThr", C.c_double), ("covThr", C.c_float),
                ("covMode", C.c_int), ("addBacktrace", C.c_int), ("maxAccept", C.c_int), ("maxRejected", C.c_int),
                ("seqIdThr", C.c_float), ("alnLenThr", C.c_int), ("seqIdMode", C.c_int), ("altAlignment", C.c_int), ("skipUndefinedDiagonals", C.c_int)]


class KmerIndexParams(C.Structure):
    _fields_ = [("kmerSize", C.c_int32), ("spaced", C.c_int32), ("kmerThr", C.c_int32), ("maskLowerCase", C.c_int32),
                ("maskNrepeats", C.c_int32)]


class KmerSearchParams(C.Structure):
    _fields_ = [("maxResListLen", C.c_int32), ("minDiagScoreThr", C.c_int32), ("bins", C.c_int32), ("kmerScoreOnly", C.c_int32),
                ("maxDbMatches", C.c_int64), ("foundDiagonalsSize", C.c_int64), ("l2CacheSize", C.c_uint64)]


class KmerQuery(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("kmerThr", C.c_void_p), ("profile", C.c_void_p), ("L", C.c_int32),
                ("reserved", C.c_int32), ("identity", C.c_int64)]


KMER_HIT_DT = np.dtype([("id", np.uint32), ("score", np.int32), ("diag", np.uint16), ("pad", np.uint16)])


class FsgpuError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libfsgpu.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FsgpuError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
    sig = {
        "fsgpu_create": (i32, [i32, C.POINTER(vp)]),
        "fsgpu_destroy": (None, [vp]),
        "fsgpu_last_error": (C.c_char_p, [vp]),
        "fsgpu_clone": (i32, [vp, C.POINTER(vp)]),
        "fsgpu_device": (i32, [vp]),
        "fsgpu_device_count": (i32, []),
        "fsgpu_gapless_plan_items": (i64, [vp, C.c_uint32, i32, f64, vp, u64, vp]),
        "fsgpu_db_broadcast": (i32, [vp, C.POINTER(vp), i32, C.POINTER(i32)]),
        "fsgpu_rccl_selfcheck": (i32, [vp]),
        "fsgpu_stream": (vp, [vp]),
        "fsgpu_db_load": (i32, [vp, vp, vp, vp, vp, u64, u64]),
        "fsgpu_db_adopt_device": (i32, [vp, vp, vp, vp, vp, u64, u64]),
        "fsgpu_db_size": (u64, [vp]),
        "fsgpu_db_residues": (u64, [vp]),
        "fsgpu_gapless_scan": (i32, [vp, vp, i32, i32, i32, i64, i32, vp, C.POINTER(i32)]),
        "fsgpu_gapless_scores": (i32, [vp, vp]),
        "fsgpu_gapless_launch": (i32, [vp, vp, i32, i32, i32, i64, i32]),
        "fsgpu_gapless_finish": (i32, [vp, vp, C.POINTER(i32)]),
        "fsgpu_sw_batch": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, vp]),
        "fsgpu_sw_launch": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32]),
        "fsgpu_sw_finish": (i32, [vp, vp, vp]),
        "fsgpu_last_kernel_ms": (f64, [vp, i32]),
        "fsgpu_sw_last_passes": (None, [vp, vp]),
        "fsgpu_kmer_index_build": (i32, [vp, vp, vp]),
        "fsgpu_kmer_index_entries": (u64, [vp]),
        "fsgpu_kmer_search": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
        "fsgpu_kmer_index_copy": (i32, [vp, vp, vp, vp]),
        "fsgpu_kmer_row_copy": (i32, [vp, i32, vp, vp]),
        "fsgpu_kmer_last_counts": (None, [vp, vp]),
        "fsgpu_kmer_last_segments": (None, [vp, vp]),
        "fsgpu_kmer_batch_hint": (i32, [vp]),
        "fsgpu_kmer_plan_coarse": (i32, [vp, u64, C.c_uint32, vp, vp, C.c_uint32]),
        "fshost_kmer_query_prepare": (i32, [vp, vp, vp, i32, i32, f32, i32, i32, i32, vp, vp]),
        "fshost_kmer_threshold": (i32, [f32, i32]),
        "fshost_matrix_create": (vp, [i32, f32, f32]),
        "fshost_matrix_from_text": (vp, [C.c_char_p, f32, f32]),
        "fshost_matrix_from_scores": (vp, [vp, i32, vp]),
        "fshost_matrix_free": (None, [vp]),
        "fshost_matrix_size": (i32, [vp]),
        "fshost_matrix_scores": (C.POINTER(C.c_int16), [vp]),
        "fshost_matrix_background": (C.POINTER(C.c_double), [vp]),
        "fshost_matrix_encode": (None, [vp, C.c_char_p, i32, vp]),
        "fshost_matrix_letter": (C.c_char, [vp, i32]),
        "fshost_comp_bias": (None, [vp, vp, i32, f32, vp]),
        "fshost_round_bias": (None, [vp, i32, vp]),
        "fshost_prefilter_profile": (i32, [vp, vp, i32, i32, f32, vp, C.POINTER(i32)]),
        "fshost_align_profiles": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
        "fshost_evaluer_create": (vp, [C.c_char_p, u64]),
        "fshost_evaluer_free": (None, [vp]),
        "fshost_predict_mu_lambda": (None, [vp, vp, C.c_uint, i32, C.POINTER(f64), C.POINTER(f64)]),
        "fshost_evalue_corr": (f64, [vp, f64, f64, f64]),
        "fshost_params_default": (None, [C.POINTER(Params)]),
        "fshost_search_create": (vp, [vp, C.POINTER(Params), vp, C.c_char_p, vp, vp, vp, vp]),
        "fshost_search_free": (None, [vp]),
        "fshost_search_error": (C.c_char_p, [vp]),
        "fshost_search_prefilter": (i32, [vp, vp, i32, i64, vp]),
        "fshost_search_align": (i32, [vp, vp, vp, i32, i64, vp, i32, vp]),
        "fshost_search_align_batch": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "fsgpu_sw_multi": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "fsgpu_sw_multi_dir": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "fsgpu_sw_multi_dir_c": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "fsgpu_sw_multi_c": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        "fshost_search_backtrace": (C.c_char_p, [vp, vp]),
        "fshost_search_stats": (None, [vp, vp]),
        "fshost_search_backtrace_counts": (None, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "fshost_search_last_sw": (None, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "fshost_block_backtrace": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32),
                                         C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]),
        "fshost_format_prefilter_hit": (C.c_size_t, [C.c_char_p, C.c_uint32, i32, i32]),
        "fshost_format_result": (C.c_size_t, [C.c_char_p, vp, C.c_char_p, i32]),
        "fsgpu_gapless_scan_multi": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "fsgpu_gapless_last_batch": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "fsgpu_gapless_scores_multi": (i32, [vp, i32, vp]),
        "fsgpu_sw_batch_seqs": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        "fsgpu_diag_rescore": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp]),
        "fshost_search_prefilter_batch": (i32, [vp, i32, vp, vp, vp, vp, vp]),
        "fshost_search_kmer_batch": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "fshost_search_rescore_diagonal_batch": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "fshost_banded_backtrace": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]),
        "fshost_search_startpos_backtrace": (i32, [vp, vp, vp, i32, C.c_uint32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_uint), C.c_char_p, C.c_size_t]),
        "fshost_set_host_workers": (None, [i32]),
        "fshost_host_workers": (i32, []),
        "fshost_usable_cores": (i32, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def exported_symbols():
    return ["fsgpu_create", "fsgpu_destroy", "fsgpu_last_error", "fsgpu_device", "fsgpu_stream", "fsgpu_db_load",
            "fsgpu_db_adopt_device", "fsgpu_db_size", "fsgpu_db_residues", "fsgpu_gapless_scan", "fsgpu_gapless_scores",
            "fsgpu_gapless_launch", "fsgpu_gapless_finish", "fsgpu_sw_batch", "fsgpu_sw_multi", "fsgpu_sw_multi_dir", "fsgpu_sw_multi_dir_c", "fsgpu_sw_multi_c", "fsgpu_sw_launch", "fsgpu_sw_finish",
            "fsgpu_db_broadcast", "fsgpu_rccl_selfcheck", "fsgpu_device_count", "fsgpu_gapless_plan_items",
            "fsgpu_last_kernel_ms", "fsgpu_sw_last_passes", "fsgpu_kmer_index_build", "fsgpu_kmer_index_entries", "fsgpu_kmer_search",
            "fsgpu_kmer_index_copy", "fsgpu_kmer_row_copy", "fsgpu_kmer_last_counts", "fsgpu_kmer_last_segments", "fsgpu_kmer_plan_coarse", "fsgpu_kmer_batch_hint"]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Matrix:
    """fshost_matrix: bit-scaled substitution matrix (built-in 3Di / BLOSUM62 or user .out text)."""

    def __init__(self, which=None, bit_factor=2.0, score_bias=0.0, text=None):
        L = lib()
        self.h = L.fshost_matrix_from_text(text.encode(), bit_factor, score_bias) if text is not None else \
            L.fshost_matrix_create(which, bit_factor, score_bias)
        if not self.h:
            raise FsgpuError("matrix construction failed")
        self.n = L.fshost_matrix_size(self.h)

    def scores(self):
        return np.ctypeslib.as_array(lib().fshost_matrix_scores(self.h), (self.n * self.n,)).copy().reshape(self.n, self.n)

    def background(self):
        return np.ctypeslib.as_array(lib().fshost_matrix_background(self.h), (self.n,)).copy()

    def encode(self, s):
        out = np.zeros(len(s), np.uint8)
        lib().fshost_matrix_encode(self.h, s.encode(), len(s), _ptr(out))
        return out

    def comp_bias(self, seq, scale):
        seq = np.ascontiguousarray(seq, np.uint8)
        out = np.zeros(len(seq), np.float32)
        lib().fshost_comp_bias(self.h, _ptr(seq), len(seq), scale, _ptr(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().fshost_matrix_free(self.h)
            self.h = None


def prefilter_profile(m3di, q3di, comp_bias=True, scale=0.15):
    q = np.ascontiguousarray(q3di, np.uint8)
    pssm = np.zeros((m3di.n, len(q)), np.int8)
    cap = C.c_int(0)
    rc = lib().fshost_prefilter_profile(m3di.h, _ptr(q), len(q), int(comp_bias), scale, _ptr(pssm), C.byref(cap))
    if rc != 0:
        raise FsgpuError(f"fshost_prefilter_profile rc={rc}")
    return pssm, cap.value


def align_profiles(mAA, m3Di, qAA, q3Di, comp_bias=True, scale=0.5):
    qa = np.ascontiguousarray(qAA, np.uint8)
    q3 = np.ascontiguousarray(q3Di, np.uint8)
    n, L_ = m3Di.n, len(q3)
    pAA = np.zeros((n, L_), np.int16)
    p3 = np.zeros((n, L_), np.int16)
    cbA = np.zeros(L_, np.int8)
    cbS = np.zeros(L_, np.int8)
    rc = lib().fshost_align_profiles(mAA.h, m3Di.h, _ptr(qa), _ptr(q3), L_, int(comp_bias), scale, _ptr(pAA), _ptr(p3),
                                     _ptr(cbA), _ptr(cbS))
    if rc != 0:
        raise FsgpuError(f"fshost_align_profiles rc={rc}")
    return pAA, p3, cbA, cbS


def kmer_threshold(sensitivity=9.5, kmer_size=6):
    return lib().fshost_kmer_threshold(sensitivity, kmer_size)


def kmer_query_prepare(m_kmer, m_ungapped, q3di, comp_bias=True, scale=0.15, kmer_thr=78, kmer_size=6, spaced=1):
    """-> (seq, kmerThr[nPos], profile[L,21]) for Context.kmer_search"""
    q = np.ascontiguousarray(q3di, np.uint8)
    L_ = len(q)
    thr = np.zeros(max(L_, 1), np.int16)
    prof = np.zeros((max(L_, 1), 21), np.int8)
    n = lib().fshost_kmer_query_prepare(m_kmer.h, m_ungapped.h, _ptr(q), L_, int(comp_bias), scale, kmer_thr, kmer_size, spaced,
                                        _ptr(thr), _ptr(prof))
    if n < 0:
        raise FsgpuError("fshost_kmer_query_prepare failed")
    return q, thr[:n].copy(), prof[:L_].copy()


class Evaluer:
    def __init__(self, db_residues, nn_path=None):
        self.h = lib().fshost_evaluer_create(nn_path.encode() if nn_path else None, db_residues)
        if not self.h:
            raise FsgpuError("cannot load e-value network")

    def mu_lambda(self, q3di, alphabet=21):
        q = np.ascontiguousarray(q3di, np.uint8)
        lam, mu = C.c_double(), C.c_double()
        lib().fshost_predict_mu_lambda(self.h, _ptr(q), len(q), alphabet, C.byref(lam), C.byref(mu))
        return lam.value, mu.value

    def evalue_corr(self, score, lam, mu):
        return lib().fshost_evalue_corr(self.h, float(score), lam, mu)

    def __del__(self):
        if getattr(self, "h", None):
            lib().fshost_evaluer_free(self.h)
            self.h = None


class Context:
    """fsgpu_ctx: one MI355X, one HIP stream, one resident target DB."""

    def __init__(self, device=0, _clone_of=None):
        h = C.c_void_p()
        if _clone_of is not None:
            rc = lib().fsgpu_clone(_clone_of.h, C.byref(h))
        else:
            rc = lib().fsgpu_create(device, C.byref(h))
        if rc != 0:
            raise FsgpuError(f"fsgpu_create({device}) rc={rc}: {lib().fsgpu_last_error(None).decode()}")
        self.h = h
        self._keep = None if _clone_of is None else _clone_of._keep

    def clone(self):
        """second context (own stream + scratch) sharing this context's resident DB"""
        return Context(_clone_of=self)

    def broadcast_db_to(self, others):
        """replicate this context's resident DB into contexts created on other devices (fsgpu_db_broadcast); returns
        True when RCCL carried the broadcast, False for peer copies"""
        arr = (C.c_void_p * len(others))(*[o.h for o in others])
        used = C.c_int(0)
        self._chk(lib().fsgpu_db_broadcast(self.h, arr, len(others), C.byref(used)), "fsgpu_db_broadcast")
        for o in others:
            o._keep = self._keep
        return bool(used.value)

    def rccl_selfcheck(self):
        """librccl on this device alone (one-rank communicator, 1 MiB broadcast in place); raises when it does not run"""
        self._chk(lib().fsgpu_rccl_selfcheck(self.h), "fsgpu_rccl_selfcheck")

    def _chk(self, rc, what):
        if rc != 0:
            raise FsgpuError(f"{what} rc={rc}: {lib().fsgpu_last_error(self.h).decode()}")

    def load_db(self, db):
        """db: foldseek_amd.synth.PaddedDB-like (data3di, dataaa|None, offsets[n+1], lengths[n])."""
        d3 = np.ascontiguousarray(db.data3di, np.uint8)
        da = None if db.dataaa is None else np.ascontiguousarray(db.dataaa, np.uint8)
        off = np.ascontiguousarray(db.offsets, np.uint64)
        ln = np.ascontiguousarray(db.lengths, np.int32)
        self._keep = (d3, da, off, ln)
        self._chk(lib().fsgpu_db_load(self.h, _ptr(d3), _ptr(da), _ptr(off), _ptr(ln), len(ln), d3.size), "fsgpu_db_load")

    def adopt_device_db(self, d3_ptr, da_ptr, off_ptr, len_ptr, n, nbytes):
        self._chk(lib().fsgpu_db_adopt_device(self.h, d3_ptr, da_ptr, off_ptr, len_ptr, n, nbytes), "fsgpu_db_adopt_device")

    @property
    def n(self):
        return lib().fsgpu_db_size(self.h)

    @property
    def residues(self):
        return lib().fsgpu_db_residues(self.h)

    @property
    def stream(self):
        return lib().fsgpu_stream(self.h)

    def gapless_scan(self, pssm, cap, min_score=30, identity=-1, max_res=1000):
        pssm = np.ascontiguousarray(pssm, np.int8)
        out = np.zeros(max_res, HIT_DT)
        nout = C.c_int(0)
        self._chk(lib().fsgpu_gapless_scan(self.h, _ptr(pssm), pssm.shape[1], cap, min_score, identity, max_res, _ptr(out),
                                           C.byref(nout)), "fsgpu_gapless_scan")
        return out[:nout.value]

    # ---- k-mer prefilter ------------------------------------------------------------------------------------
    def kmer_index_build(self, kmer_matrix, kmer_thr=78, kmer_size=6, spaced=1, mask_lower_case=1, mask_n_repeats=6):
        """kmer_matrix: Matrix(FSHOST_MAT_3DI, 8.0, -0.2) -- the prefilter's seeding matrix"""
        p = KmerIndexParams(kmer_size, spaced, kmer_thr, mask_lower_case, mask_n_repeats)
        sub = np.ascontiguousarray(kmer_matrix.scores(), np.int16)
        self._kmer_ip = p
        self._chk(lib().fsgpu_kmer_index_build(self.h, C.byref(p), _ptr(sub)), "fsgpu_kmer_index_build")

    @property
    def kmer_index_entries(self):
        return lib().fsgpu_kmer_index_entries(self.h)

    def kmer_index_copy(self, nbytes_db):
        off = np.zeros(64000001, np.uint32)
        ent = np.zeros(max(1, self.kmer_index_entries), np.uint64)
        msk = np.zeros(max(1, nbytes_db), np.uint8)
        self._chk(lib().fsgpu_kmer_index_copy(self.h, _ptr(off), _ptr(ent), _ptr(msk)), "fsgpu_kmer_index_copy")
        return off, ent[:self.kmer_index_entries], msk[:nbytes_db]

    def kmer_index_reference_order(self, nbytes_db):
        """(offsets uint64[64e6+1], seqId uint32[], pos uint16[], masked) re-ordered to the reference's k-mer numbering
        (first3 + 8000*last3); the device keeps its table first-3-mer major (kmerDeviceIndex)."""
        off, ent, msk = self.kmer_index_copy(nbytes_db)
        k = np.arange(64000000, dtype=np.int64)
        kdev = (k % 8000) * 8000 + k // 8000
        size_dev = np.diff(off.astype(np.int64))
        size_ref = size_dev[kdev]
        roff = np.zeros(64000001, np.uint64)
        roff[1:] = np.cumsum(size_ref)
        nz = np.nonzero(size_ref)[0]
        starts = off.astype(np.int64)[kdev[nz]]
        lens = size_ref[nz]
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
        e = ent[idx]
        return roff, (e >> np.uint64(16)).astype(np.uint32), (e & np.uint64(0xffff)).astype(np.uint16), msk

    def kmer_row(self, row):
        s = np.zeros(8000, np.int16); ix = np.zeros(8000, np.uint16)
        self._chk(lib().fsgpu_kmer_row_copy(self.h, row, _ptr(s), _ptr(ix)), "fsgpu_kmer_row_copy")
        return s, ix

    def kmer_search(self, prepared, identity=None, max_res=1000, min_diag=30, bins=0, max_db_matches=0,
                    found_diagonals_size=0, l2_cache_size=0, want_stats=False, kmer_score_only=False):
        """prepared: list of (seq uint8[L], thr int16[nPos], profile int8[L,21]) from kmer_query_prepare."""
        nq = len(prepared)
        sp = KmerSearchParams(max_res, min_diag, bins, 1 if kmer_score_only else 0, max_db_matches, found_diagonals_size, l2_cache_size)
        qs = (KmerQuery * max(nq, 1))()
        keep = []
        for i, (seq, thr, prof) in enumerate(prepared):
            seq = np.ascontiguousarray(seq, np.uint8); thr = np.ascontiguousarray(thr, np.int16); prof = np.ascontiguousarray(prof, np.int8)
            keep.append((seq, thr, prof))
            qs[i].seq = seq.ctypes.data; qs[i].kmerThr = thr.ctypes.data; qs[i].profile = prof.ctypes.data
            qs[i].L = len(seq); qs[i].reserved = 0
            qs[i].identity = -1 if identity is None else int(identity[i])
        out = np.zeros((max(nq, 1), max_res), KMER_HIT_DT)
        nout = np.zeros(max(nq, 1), np.int32)
        status = np.zeros(max(nq, 1), np.int32)
        stats = np.zeros((max(nq, 1), 4))
        self._chk(lib().fsgpu_kmer_search(self.h, C.byref(sp), C.cast(qs, C.c_void_p), nq, _ptr(out), _ptr(nout), _ptr(status), _ptr(stats)),
                  "fsgpu_kmer_search")
        res = [out[q, :nout[q]].copy() for q in range(nq)]
        return (res, status[:nq], stats[:nq]) if want_stats else (res, status[:nq])

    def kmer_counts(self):
        """last batch: similar k-mers probed, index hits, double-diagonal candidates, elements handed to the host"""
        out = np.zeros(4, np.uint64)
        lib().fsgpu_kmer_last_counts(self.h, _ptr(out))
        return out

    def kmer_segments(self):
        """last batch's partition: [1] = [4] (query, chunk, key) runs of the duplicate stage, [3] tiles, [5] coarse keys, [6] ids of the widest key"""
        out = np.zeros(7, np.uint32)
        lib().fsgpu_kmer_last_segments(self.h, _ptr(out))
        return out

    def kmer_stage_ms(self):
        """ms of the last k-mer batch: device total, count, lists, emit, partition, double-diagonal detection, score, walk, select; [9] host tail; [10] k_kmer_lists kernel alone"""
        return [lib().fsgpu_last_kernel_ms(self.h, 2 + i) for i in range(11)]

    def gapless_scores_multi(self, query_index):
        out = np.zeros(self.n, np.uint8)
        self._chk(lib().fsgpu_gapless_scores_multi(self.h, int(query_index), _ptr(out)), "gapless_scores_multi")
        return out

    def gapless_last_batch(self):
        a, b = C.c_int(0), C.c_int(0)
        lib().fsgpu_gapless_last_batch(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def gapless_scores(self):
        s = np.zeros(self.n, np.uint8)
        self._chk(lib().fsgpu_gapless_scores(self.h, _ptr(s)), "fsgpu_gapless_scores")
        return s

    def sw_multi_dir(self, queries, direction, selections=None, gap_open=10, gap_extend=1):
        """queries: list of (pAAf|None, p3f, pAAr|None, p3r, L, target_ids); one direction of the structure SW for the
        selected pairs (fsgpu_sw_multi_dir).  Returns one SWRES array per query (unselected entries stay zero)."""
        class Q(C.Structure):
            _fields_ = [("pAA_fwd", C.c_void_p), ("p3Di_fwd", C.c_void_p), ("pAA_rev", C.c_void_p), ("p3Di_rev", C.c_void_p),
                        ("L", C.c_int32), ("n", C.c_int32), ("targetIds", C.c_void_p)]
        nq = len(queries)
        keep, arr = [], (Q * nq)()
        for i, (paf, p3f, par_, p3r, L, ids) in enumerate(queries):
            bufs = [None if x is None else np.ascontiguousarray(x, np.int16) for x in (paf, p3f, par_, p3r)]
            ids = np.ascontiguousarray(ids, np.uint32)
            keep.append((bufs, ids))
            arr[i].pAA_fwd, arr[i].p3Di_fwd, arr[i].pAA_rev, arr[i].p3Di_rev = [None if b is None else b.ctypes.data for b in bufs]
            arr[i].L, arr[i].n, arr[i].targetIds = int(L), len(ids), ids.ctypes.data
        total = sum(len(k[1]) for k in keep)
        out = np.zeros(max(1, total), SWRES_DT)
        selp = nselp = None
        if selections is not None:
            sels = [np.ascontiguousarray(x, np.int32) for x in selections]
            keep.append(sels)
            selp = (C.c_void_p * nq)(*[x.ctypes.data for x in sels])
            nselp = (C.c_int32 * nq)(*[len(x) for x in sels])
        self._chk(lib().fsgpu_sw_multi_dir(self.h, C.cast(arr, C.c_void_p), nq, gap_open, gap_extend, direction,
                                           None if selp is None else C.cast(selp, C.c_void_p), None if nselp is None else C.cast(nselp, C.c_void_p),
                                           out.ctypes.data), "fsgpu_sw_multi_dir")
        res, b = [], 0
        for k in keep[:nq]:
            res.append(out[b:b + len(k[1])].copy()); b += len(k[1])
        return res

    def sw_multi_dir_c(self, mat3di, matAA, queries, direction, selections=None, gap_open=10, gap_extend=1):
        """Compact form (fsgpu_sw_multi_dir_c): mat3di / matAA int8 [21, 21] (matAA None = 3Di only); queries: list of
        (qAA|None, q3Di, cbAA_fwd|None, cb3Di_fwd|None, cbAA_rev|None, cb3Di_rev|None, target_ids).  Returns one SWRES array per query."""
        class Q(C.Structure):
            _fields_ = [("qAA", C.c_void_p), ("q3Di", C.c_void_p), ("cbAA_fwd", C.c_void_p), ("cb3Di_fwd", C.c_void_p), ("cbAA_rev", C.c_void_p),
                        ("cb3Di_rev", C.c_void_p), ("L", C.c_int32), ("n", C.c_int32), ("targetIds", C.c_void_p)]
        nq = len(queries)
        m3 = np.ascontiguousarray(mat3di, np.int8)
        mA = None if matAA is None else np.ascontiguousarray(matAA, np.int8)
        keep, arr = [], (Q * nq)()
        for i, (qa, q3, cbaf, cb3f, cbar, cb3r, ids) in enumerate(queries):
            q3 = np.ascontiguousarray(q3, np.uint8)
            qa = None if qa is None else np.ascontiguousarray(qa, np.uint8)
            cbs = [None if x is None else np.ascontiguousarray(x, np.int8) for x in (cbaf, cb3f, cbar, cb3r)]
            ids = np.ascontiguousarray(ids, np.uint32)
            keep.append((ids, q3, qa, cbs))
            arr[i].qAA, arr[i].q3Di = (None if qa is None else qa.ctypes.data), q3.ctypes.data
            arr[i].cbAA_fwd, arr[i].cb3Di_fwd, arr[i].cbAA_rev, arr[i].cb3Di_rev = [None if b is None else b.ctypes.data for b in cbs]
            arr[i].L, arr[i].n, arr[i].targetIds = len(q3), len(ids), ids.ctypes.data
        total = sum(len(k[0]) for k in keep)
        out = np.zeros(max(1, total), SWRES_DT)
        selp = nselp = None
        if selections is not None:
            sels = [np.ascontiguousarray(x, np.int32) for x in selections]
            keep.append(sels)
            selp = (C.c_void_p * nq)(*[x.ctypes.data for x in sels])
            nselp = (C.c_int32 * nq)(*[len(x) for x in sels])
        self._chk(lib().fsgpu_sw_multi_dir_c(self.h, m3.ctypes.data, None if mA is None else mA.ctypes.data, C.cast(arr, C.c_void_p), nq, gap_open, gap_extend,
                                             direction, None if selp is None else C.cast(selp, C.c_void_p), None if nselp is None else C.cast(nselp, C.c_void_p),
                                             out.ctypes.data), "fsgpu_sw_multi_dir_c")
        res, b = [], 0
        for k in keep[:nq]:
            res.append(out[b:b + len(k[0])].copy()); b += len(k[0])
        return res

    def sw_multi_c(self, mat3di, matAA, queries, gap_open=10, gap_extend=1):
        """both directions in one submission (fsgpu_sw_multi_c); queries as in sw_multi_dir_c.  Returns (fwd arrays, rev arrays)."""
        class Q(C.Structure):
            _fields_ = [("qAA", C.c_void_p), ("q3Di", C.c_void_p), ("cbAA_fwd", C.c_void_p), ("cb3Di_fwd", C.c_void_p), ("cbAA_rev", C.c_void_p),
                        ("cb3Di_rev", C.c_void_p), ("L", C.c_int32), ("n", C.c_int32), ("targetIds", C.c_void_p)]
        nq = len(queries)
        m3 = np.ascontiguousarray(mat3di, np.int8)
        mA = None if matAA is None else np.ascontiguousarray(matAA, np.int8)
        keep, arr = [], (Q * nq)()
        for i, (qa, q3, cbaf, cb3f, cbar, cb3r, ids) in enumerate(queries):
            q3 = np.ascontiguousarray(q3, np.uint8)
            qa = None if qa is None else np.ascontiguousarray(qa, np.uint8)
            cbs = [None if x is None else np.ascontiguousarray(x, np.int8) for x in (cbaf, cb3f, cbar, cb3r)]
            ids = np.ascontiguousarray(ids, np.uint32)
            keep.append((ids, q3, qa, cbs))
            arr[i].qAA, arr[i].q3Di = (None if qa is None else qa.ctypes.data), q3.ctypes.data
            arr[i].cbAA_fwd, arr[i].cb3Di_fwd, arr[i].cbAA_rev, arr[i].cb3Di_rev = [None if b is None else b.ctypes.data for b in cbs]
            arr[i].L, arr[i].n, arr[i].targetIds = len(q3), len(ids), ids.ctypes.data
        total = sum(len(k[0]) for k in keep)
        fwd, rev = np.zeros(max(1, total), SWRES_DT), np.zeros(max(1, total), SWRES_DT)
        self._chk(lib().fsgpu_sw_multi_c(self.h, m3.ctypes.data, None if mA is None else mA.ctypes.data, C.cast(arr, C.c_void_p), nq, gap_open, gap_extend,
                                         fwd.ctypes.data, rev.ctypes.data), "fsgpu_sw_multi_c")
        rf, rr, b = [], [], 0
        for k in keep:
            rf.append(fwd[b:b + len(k[0])].copy()); rr.append(rev[b:b + len(k[0])].copy()); b += len(k[0])
        return rf, rr

    def sw_batch(self, pAAf, p3f, pAAr, p3r, target_ids, gap_open=10, gap_extend=1):
        p3f = np.ascontiguousarray(p3f, np.int16)
        p3r = np.ascontiguousarray(p3r, np.int16)
        pAAf = None if pAAf is None else np.ascontiguousarray(pAAf, np.int16)
        pAAr = None if pAAr is None else np.ascontiguousarray(pAAr, np.int16)
        t = np.ascontiguousarray(target_ids, np.uint32)
        fwd = np.zeros(len(t), SWRES_DT)
        rev = np.zeros(len(t), SWRES_DT)
        self._chk(lib().fsgpu_sw_batch(self.h, _ptr(pAAf), _ptr(p3f), _ptr(pAAr), _ptr(p3r), p3f.shape[1], _ptr(t), len(t),
                                       gap_open, gap_extend, _ptr(fwd), _ptr(rev)), "fsgpu_sw_batch")
        return fwd, rev

    def kernel_ms(self, which):
        return lib().fsgpu_last_kernel_ms(self.h, which)

    def sw_last_passes(self):
        """[[ms, cells, pairs, DP wave-instructions] forward, [...] reversed] of the last multi-query SW passes (k_sw2 launches only)"""
        out = np.zeros(8)
        lib().fsgpu_sw_last_passes(self.h, _ptr(out))
        return out.reshape(2, 4)

    def close(self):
        if getattr(self, "h", None):
            lib().fsgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Search:
    """fshost_search: one query at a time through prefilter + structurealign on a Context."""

    def __init__(self, ctx, params=None, keys=None, nn_path=None):
        self.ctx = ctx
        self.par = params if params is not None else default_params()
        d3, da, off, ln = ctx._keep
        self._keys = None if keys is None else np.ascontiguousarray(keys, np.uint32)
        self.h = lib().fshost_search_create(ctx.h, C.byref(self.par), _ptr(self._keys), nn_path.encode() if nn_path else None,
                                            _ptr(d3), _ptr(da), _ptr(off), _ptr(ln))
        err = lib().fshost_search_error(self.h).decode() if self.h else "null"
        if not self.h or err:
            raise FsgpuError(f"fshost_search_create: {err}")

    def prefilter(self, q3di, identity=-1):
        q = np.ascontiguousarray(q3di, np.uint8)
        hits = np.zeros(self.par.maxResListLen, HIT_DT)
        n = lib().fshost_search_prefilter(self.h, _ptr(q), len(q), identity, _ptr(hits))
        if n < 0:
            raise FsgpuError(f"prefilter rc={n}: {lib().fshost_search_error(self.h).decode()}")
        return hits[:n]

    def prefilter_batch(self, q3dis, identity=None):
        """several queries, as few scan launches as their lengths allow (fsgpu_gapless_scan_multi); list of hit arrays"""
        nq = len(q3dis)
        q3 = [np.ascontiguousarray(x, np.uint8) for x in q3dis]
        P = C.c_void_p * max(nq, 1)
        p3 = P(*[x.ctypes.data for x in q3])
        Ls = np.array([len(x) for x in q3], np.int32)
        ident = None if identity is None else np.ascontiguousarray(identity, np.int64)
        K = self.par.maxResListLen
        hits = np.zeros(max(1, nq * K), HIT_DT)
        nh = np.zeros(max(nq, 1), np.int32)
        rc = lib().fshost_search_prefilter_batch(self.h, nq, C.cast(p3, C.c_void_p), _ptr(Ls), _ptr(ident), _ptr(hits), _ptr(nh))
        if rc != 0:
            raise FsgpuError(f"prefilter_batch rc={rc}: {lib().fshost_search_error(self.h).decode()}")
        return [hits[i * K:i * K + nh[i]].copy() for i in range(nq)]

    def align(self, qAA, q3di, target_ids, identity=-1, with_backtrace=False):
        qa = np.ascontiguousarray(qAA, np.uint8)
        q3 = np.ascontiguousarray(q3di, np.uint8)
        t = np.ascontiguousarray(target_ids, np.uint32)
        res = np.zeros(max(1, len(t) * (1 + max(0, self.par.altAlignment))), RESULT_DT)
        n = lib().fshost_search_align(self.h, _ptr(qa), _ptr(q3), len(q3), identity, _ptr(t), len(t), _ptr(res))
        if n < 0:
            raise FsgpuError(f"align rc={n}: {lib().fshost_search_error(self.h).decode()}")
        res = res[:n]
        if with_backtrace:
            bts = [lib().fshost_search_backtrace(self.h, C.c_void_p(res[i:i + 1].ctypes.data)).decode() for i in range(n)]
            return res, bts
        return res

    def align_batch(self, qAAs, q3dis, target_id_lists, identity=None, with_backtrace=False):
        """several queries, one forward + one reversed device pass (fsgpu_sw_multi_dir); returns a list of result arrays (and backtrace lists)"""
        nq = len(q3dis)
        qa = [np.ascontiguousarray(x, np.uint8) for x in qAAs]
        q3 = [np.ascontiguousarray(x, np.uint8) for x in q3dis]
        ts = [np.ascontiguousarray(x, np.uint32) for x in target_id_lists]
        res = [np.zeros(max(1, len(t) * (1 + max(0, self.par.altAlignment))), RESULT_DT) for t in ts]
        P = C.c_void_p * max(nq, 1)
        pa, p3, pt, pr = P(*[x.ctypes.data for x in qa]), P(*[x.ctypes.data for x in q3]), P(*[x.ctypes.data for x in ts]), P(*[x.ctypes.data for x in res])
        Ls = np.array([len(x) for x in q3], np.int32)
        ns = np.array([len(x) for x in ts], np.int32)
        ident = None if identity is None else np.ascontiguousarray(identity, np.int64)
        nres = np.zeros(max(nq, 1), np.int32)
        rc = lib().fshost_search_align_batch(self.h, nq, C.cast(pa, C.c_void_p), C.cast(p3, C.c_void_p), _ptr(Ls), _ptr(ident), C.cast(pt, C.c_void_p),
                                             _ptr(ns), C.cast(pr, C.c_void_p), _ptr(nres))
        if rc != 0:
            raise FsgpuError(f"align_batch rc={rc}: {lib().fshost_search_error(self.h).decode()}")
        out = [res[i][:nres[i]] for i in range(nq)]
        if with_backtrace:
            bts = [[lib().fshost_search_backtrace(self.h, C.c_void_p(out[i][k:k + 1].ctypes.data)).decode() for k in range(len(out[i]))] for i in range(nq)]
            return out, bts
        return out

    def kmer_batch(self, m_kmer, m_ungapped, kmer_thr, spaced, qaa_ptrs, q3_ptrs, lens, pref_identity=None, aln_identity=None, max_res=200, min_diag=30,
                   kmer_score_only=False):
        """fshost_search_kmer_batch: k-mer prefilter + coverage pre-filter + structure alignment of a batch of queries in ONE library call (the per-batch
        body of the `search` module).  qaa_ptrs / q3_ptrs: uint64 arrays with the host addresses of the queries' code strings (0..20), lens: their lengths.
        Returns a dict: nhits, status, nkept, nres (int32[nq]), hits [nq, max_res], kept [nq, max_res], results [nq, max_res * (1 + alt)], seconds[4]."""
        nq = len(lens)
        pa = np.ascontiguousarray(qaa_ptrs, np.uint64); p3 = np.ascontiguousarray(q3_ptrs, np.uint64)
        Ls = np.ascontiguousarray(lens, np.int32)
        pi = None if pref_identity is None else np.ascontiguousarray(pref_identity, np.int64)
        ai = None if aln_identity is None else np.ascontiguousarray(aln_identity, np.int64)
        sp = KmerSearchParams(max_res, min_diag, 0, 1 if kmer_score_only else 0, 0, 0, 0)
        rcap = max_res * (1 + max(0, self.par.altAlignment))
        n1 = max(nq, 1)
        out = {"hits": np.zeros((n1, max_res), KMER_HIT_DT), "nhits": np.zeros(n1, np.int32), "status": np.zeros(n1, np.int32),
               "kept": np.zeros((n1, max_res), np.uint32), "nkept": np.zeros(n1, np.int32), "results": np.zeros((n1, rcap), RESULT_DT),
               "nres": np.zeros(n1, np.int32), "seconds": np.zeros(4)}
        rc = lib().fshost_search_kmer_batch(self.h, m_kmer.h, m_ungapped.h, C.cast(C.byref(sp), C.c_void_p), int(kmer_thr), int(spaced), nq, _ptr(pa), _ptr(p3), _ptr(Ls),
                                            _ptr(pi), _ptr(ai), _ptr(out["hits"]), _ptr(out["nhits"]), _ptr(out["status"]), _ptr(out["kept"]), _ptr(out["nkept"]),
                                            _ptr(out["results"]), _ptr(out["nres"]), _ptr(out["seconds"]))
        if rc != 0:
            raise FsgpuError(f"kmer_batch rc={rc}: {lib().fshost_search_error(self.h).decode()}")
        return out

    def startpos_backtrace(self, qAA, q3di, target_id, q_end, db_end, score):
        """alignStartPosBacktrace (SSW-style start + CIGAR: device reverse pass + host banded trace-back); (ok, qStart, dbStart, ids, cigar)"""
        qa, q3 = np.ascontiguousarray(qAA, np.uint8), np.ascontiguousarray(q3di, np.uint8)
        qs, ds, ident = C.c_int(), C.c_int(), C.c_uint()
        buf = C.create_string_buffer(len(q3) + 70000)
        rc = lib().fshost_search_startpos_backtrace(self.h, _ptr(qa), _ptr(q3), len(q3), int(target_id), int(q_end), int(db_end), int(score),
                                                    C.byref(qs), C.byref(ds), C.byref(ident), buf, len(buf))
        if rc < 0:
            raise FsgpuError(f"startpos_backtrace rc={rc}: {lib().fshost_search_error(self.h).decode()}")
        return rc == 1, qs.value, ds.value, ident.value, buf.value.decode()

    def backtrace_counts(self):
        """last align_batch: (accepted hits answered by the device block aligner, all accepted hits)"""
        a, b = C.c_int64(0), C.c_int64(0)
        lib().fshost_search_backtrace_counts(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def stats(self):
        out = np.zeros(8)
        lib().fshost_search_stats(self.h, _ptr(out))
        return out

    def last_sw(self, n):
        f, r = C.c_void_p(), C.c_void_p()
        lib().fshost_search_last_sw(self.h, C.byref(f), C.byref(r))
        fa = np.ctypeslib.as_array(C.cast(f, C.POINTER(C.c_int32)), (n * 4,)).copy().view(SWRES_DT)
        ra = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_int32)), (n * 4,)).copy().view(SWRES_DT)
        return fa, ra

    def format_result(self, r, backtrace=None, add_backtrace=False):
        buf = C.create_string_buffer(1024 + 65536 * 2)
        n = lib().fshost_format_result(buf, C.c_void_p(r.ctypes.data), backtrace.encode() if backtrace else None, int(add_backtrace))
        return buf.raw[:n].decode()

    def close(self):
        if getattr(self, "h", None):
            lib().fshost_search_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def block_backtrace(mAA, m3Di, qAA, q3Di, cbAA, cbSS, tAA, t3Di, q_end, db_end, score, gap_open=10, gap_extend=1):
    """host start position + backtrace of one hit; returns (ok, qStart, dbStart, identicalAA, backtrace)"""
    qa, q3 = np.ascontiguousarray(qAA, np.uint8), np.ascontiguousarray(q3Di, np.uint8)
    ta, t3 = np.ascontiguousarray(tAA, np.uint8), np.ascontiguousarray(t3Di, np.uint8)
    ca, cs = np.ascontiguousarray(cbAA, np.int8), np.ascontiguousarray(cbSS, np.int8)
    qs, ds, ident = C.c_int(), C.c_int(), C.c_uint()
    buf = C.create_string_buffer(len(qa) + len(ta) + 8)
    ok = lib().fshost_block_backtrace(mAA.h, m3Di.h, _ptr(qa), _ptr(q3), _ptr(ca), _ptr(cs), len(qa), _ptr(ta), _ptr(t3), len(ta),
                                      int(q_end), int(db_end), int(score), gap_open, gap_extend, C.byref(qs), C.byref(ds), C.byref(ident),
                                      buf, len(buf))
    return bool(ok), qs.value, ds.value, ident.value, buf.value.decode()


def banded_backtrace(mAA, m3Di, qAA, q3Di, cbAA, cbSS, tAA, t3Di, q_start, q_end, db_start, db_end, score, gap_open=10, gap_extend=1):
    """host banded_sw + computerBacktrace of a pair with known start and end cells; returns (ok, identicalAA, backtrace)"""
    qa, q3 = np.ascontiguousarray(qAA, np.uint8), np.ascontiguousarray(q3Di, np.uint8)
    ta, t3 = np.ascontiguousarray(tAA, np.uint8), np.ascontiguousarray(t3Di, np.uint8)
    ca, cs = np.ascontiguousarray(cbAA, np.int8), np.ascontiguousarray(cbSS, np.int8)
    ident = C.c_uint()
    buf = C.create_string_buffer(len(qa) + len(ta) + 8)
    ok = lib().fshost_banded_backtrace(mAA.h, m3Di.h, _ptr(qa), _ptr(q3), _ptr(ca), _ptr(cs), _ptr(ta), _ptr(t3), int(q_start), int(q_end), int(db_start),
                                       int(db_end), int(score), gap_open, gap_extend, C.byref(ident), buf, len(buf))
    return ok == 1, ident.value, buf.value.decode()


def set_host_workers(n):
    """size of the process-wide host pool that computes the per-hit backtraces (0: the calling threads do it themselves)"""
    lib().fshost_set_host_workers(int(n))


def host_workers():
    return int(lib().fshost_host_workers())


def default_params():
    p = Params()
    lib().fshost_params_default(C.byref(p))
    return p


def format_prefilter_hit(key, score, diagonal=0):
    buf = C.create_string_buffer(64)
    n = lib().fshost_format_prefilter_hit(buf, key, score, diagonal)
    return buf.raw[:n].decode()
