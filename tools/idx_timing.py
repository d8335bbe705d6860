"""f4 measurement: what persisting the k-mer table would buy the device path.  For an N-target synthetic DB on disk: wall time of
`fsgpu-modules indexdb` (device build + renumbering + 0.9-3.4 GB written), size of the table entries inside the index, time to read them back
(warm page cache, then to the device), and the device rebuild (`fsgpu_kmer_index_build`) they would replace.  usage: idx_timing.py [N=100000]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from foldseek_amd import api, dbio, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = tempfile.mkdtemp(prefix="fs_idx_")
try:
    db = synth.make_db_fast(n, None, seed=20260923, homologs_per_query=0)
    dbio.write_seq_db_from_padded(os.path.join(w, "t_ss"), db, "3di")
    with open(os.path.join(w, "t_ss_h"), "wb") as f, open(os.path.join(w, "t_ss_h.index"), "w") as fi:
        off = 0
        for k in range(n):
            b = b"s%d\n\0" % k
            f.write(b); fi.write("%d\t%d\t%d\n" % (k, off, len(b))); off += len(b)
    np.array([12], np.int32).tofile(os.path.join(w, "t_ss_h.dbtype"))
    par = ["--seed-sub-mat", "aa:3di.out,nucl:3di.out", "-k", "0", "--alph-size", "aa:21,nucl:5", "--mask", "0", "--mask-lower-case", "1", "--mask-n-repeat", "6",
           "--spaced-kmer-mode", "1", "-s", "9.5", "--k-score", "seq:2147483647,prof:2147483647", "--index-subset", "5", "--index-dbsuffix", "_ss", "--threads", "1"]
    t0 = time.time()
    subprocess.check_call([os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules"), "indexdb", os.path.join(w, "t_ss"), os.path.join(w, "t_ss")] + par)
    t_indexdb = time.time() - t0
    ents = {int(l.split()[0]): (int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(w, "t_ss.idx.index"))}
    table_bytes = ents[9][1] + ents[10][1]
    t0 = time.time()
    with open(os.path.join(w, "t_ss.idx"), "rb") as f:
        f.seek(ents[9][0]); e = np.frombuffer(f.read(ents[9][1]), np.uint8)
        f.seek(ents[10][0]); o = np.frombuffer(f.read(ents[10][1]), np.uint8)
    t_read = time.time() - t0
    import torch
    t0 = time.time()
    de = torch.from_numpy(e.copy()).cuda(); do = torch.from_numpy(o.copy()).cuda(); torch.cuda.synchronize()
    t_h2d = time.time() - t0
    ctx = api.Context(0); ctx.load_db(db)
    m8 = api.Matrix(0, 8.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)            # first call: kernel load
    t0 = time.time(); ctx.kmer_index_build(m8, kmer_thr=78); t_build = time.time() - t0
    print(json.dumps({"targets": n, "indexdb_wall_s": t_indexdb, "idx_file_bytes": os.path.getsize(os.path.join(w, "t_ss.idx")), "kmer_table_bytes": table_bytes,
                      "read_table_from_page_cache_s": t_read, "pageable_host_to_device_s": t_h2d, "device_rebuild_s": t_build,
                      "note": "reading the persisted table (warm page cache) + moving it to the device vs rebuilding it from the resident sequences; a persisted table "
                              "would additionally need the renumbering to the device's first-3-mer-major order"}))
finally:
    shutil.rmtree(w, ignore_errors=True)
