"""ctypes bindings for the TEST-ONLY checkers: oracle/libfso.so (C restatement) and, when present,
oracle/_ref/libfsref.so (the reference's own sources compiled in-container).  Test infrastructure only."""
import ctypes as C
import os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

SW_DT = np.dtype([("score", np.int32), ("qEnd", np.int32), ("dbEnd", np.int32), ("word", np.int32)])
REFSW_DT = np.dtype([("score", np.int32), ("qEnd", np.int32), ("dbEnd", np.int32), ("word", np.int32),
                     ("qCov", np.float32), ("tCov", np.float32)])
REFALN_DT = np.dtype([("fwdScore", np.int32), ("revScore", np.int32), ("score", np.int32),
                      ("qStart", np.int32), ("qEnd", np.int32), ("dbStart", np.int32), ("dbEnd", np.int32),
                      ("status", np.int32), ("alnLen", np.int32), ("identicalAA", np.int32),
                      ("qCov", np.float32), ("tCov", np.float32), ("seqId", np.float32), ("_pad", np.int32),
                      ("evalue", np.float64)])
HIT_DT = np.dtype([("key", np.uint32), ("score", np.int32)])


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libfso.so"])


def load_oracle():
    path = os.path.join(ORACLE_DIR, "libfso.so")
    if not os.path.exists(path):
        build_oracle()
    L = C.CDLL(path)
    L.fso_submat_build.argtypes = [f64p, f64p, C.c_double, C.c_int, C.c_float, C.c_float, i16p, f64p]
    L.fso_comp_bias.argtypes = [i16p, f64p, C.c_int, u8p, C.c_int, C.c_float, f32p]
    L.fso_round_bias.argtypes = [f32p, C.c_int, i8p]
    L.fso_ungapped_bias.argtypes = [i8p, C.c_int, i8p, C.c_int]
    L.fso_ungapped_bias.restype = C.c_int
    L.fso_ungapped_score.argtypes = [u8p, C.c_int, i8p, C.c_int, i8p, u8p, C.c_int]
    L.fso_ungapped_score.restype = C.c_int
    L.fso_sw_profiles.argtypes = [u8p, u8p, C.c_int, i8p, i8p, C.c_int, i8p, i8p, i16p, i16p]
    L.fso_sw_score_endpos.argtypes = [i16p, i16p, C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fso_sw_pass.argtypes = [i16p, i16p, C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fso_sw_rowmajor.argtypes = [i16p, i16p, C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fso_compute_cov.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    L.fso_compute_cov.restype = C.c_float
    L.fso_predict_mu_lambda.argtypes = [u8p, u8p, C.c_uint, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.fso_evalue_corr.argtypes = [C.c_double] * 4
    L.fso_evalue_corr.restype = C.c_double
    L.fso_prefilter_select.argtypes = [i32p, u32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fso_prefilter_select.restype = C.c_int
    return L


def ref_path():
    return os.path.join(ORACLE_DIR, "_ref", "libfsref.so")


def load_ref():
    """Returns the reference driver or None when oracle/_ref was not built (no /root/reference at build time)."""
    p = ref_path()
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.ref_submat.argtypes = [C.c_int, C.c_float, C.c_float, i16p, f64p]
    L.ref_submat.restype = C.c_int
    L.ref_comp_bias.argtypes = [C.c_int, C.c_float, C.c_float, u8p, C.c_int, C.c_float, f32p]
    L.ref_ungapped.argtypes = [u8p, C.c_int, C.c_int, C.c_float, u8p, i64p, i32p, C.c_int64, C.c_int, i32p]
    L.ref_ungapped.restype = C.c_double
    L.ref_structure_align.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                      u8p, u8p, i64p, i32p, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.ref_structure_align.restype = C.c_double
    if hasattr(L, "ref_structure_startpos"):
        L.ref_structure_startpos.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, u8p, u8p, i64p, i32p, C.c_int64,
                                             i32p, C.c_void_p, C.c_int64]
        L.ref_structure_startpos.restype = None
    L.ref_mu_lambda.argtypes = [u8p, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ref_evalue_corr.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int64]
    L.ref_evalue_corr.restype = C.c_double
    L.ref_num_threads.restype = C.c_int
    return L
