import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = C.CDLL("/root/repo/oracle/_ref/libfsref.so")
class P(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "kmerSize spaced kmerThr maxResListLen compBias minDiagScoreThr maskLowerCase maskNrepeats".split()] + \
               [("compBiasScale", C.c_float), ("bins", C.c_int32), ("maxDbMatches", C.c_int64), ("foundDiagonalsSize", C.c_int64)]
L.ref_kpf_create.restype = C.c_void_p
L.ref_kpf_run.restype = C.c_double
q3, qa = synth.make_queries(NQ, seed=1)
db = synth.make_db(N, (q3, qa))
A = np.frombuffer(synth.ALPHABET.encode(), np.uint8)
def ascii_of(codes):
    m = codes >= 32
    a = A[np.where(m, codes - 32, codes)]
    return np.where(m, a + 32, a).astype(np.uint8)
tl = db.lengths.astype(np.int32)
toff = np.zeros(db.n + 1, np.int64); toff[1:] = np.cumsum(tl)
tcat = np.concatenate([ascii_of(db.data3di[db.offsets[i]:db.offsets[i] + tl[i]]) for i in range(db.n)])
ql = np.array([len(x) for x in q3], np.int32)
qoff = np.zeros(NQ + 1, np.int64); qoff[1:] = np.cumsum(ql)
qcat = np.concatenate([A[x] for x in q3]).astype(np.uint8)
p = P(6, 1, 78, 1000, 1, 30, 1, 6, 0.15, 0, 0, 0)
t = time.time()
h = L.ref_kpf_create(C.byref(p), tcat.ctypes.data_as(C.c_void_p), toff.ctypes.data_as(C.c_void_p), tl.ctypes.data_as(C.c_void_p), C.c_int64(db.n), 8)
print("index build %.2fs entries=%d" % (time.time() - t, L.ref_kpf_index_entries(C.c_void_p(h))))
HIT = np.dtype([("id", np.uint32), ("score", np.int32), ("diag", np.uint16), ("pad", np.uint16)])
out = np.zeros((NQ, 1000), HIT); cnt = np.zeros(NQ, np.int32); stats = np.zeros((NQ, 4))
ident = np.full(NQ, -1, np.int64)
secs = L.ref_kpf_run(C.c_void_p(h), qcat.ctypes.data_as(C.c_void_p), qoff.ctypes.data_as(C.c_void_p), ql.ctypes.data_as(C.c_void_p), C.c_int64(NQ),
                     ident.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p))
print("run %.3fs (1 thread) -> %.2f ms/query" % (secs, secs / NQ * 1e3))
for q in range(NQ):
    print(q, ql[q], cnt[q], stats[q], out[q, :5].tolist())
