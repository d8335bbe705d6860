"""GPU test (-m gpu) of the drop-in modules: ASCII / padded MMseqs databases on disk -> fsgpu-modules ungappedprefilter ->
structurealign -> result databases, compared with the library pipeline that test_golden.py pins to the reference."""
import os
import subprocess
import numpy as np
import pytest

from foldseek_amd import api, synth, dbio
import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")


def _expected(db, q3, qa, keys_t, atype, max_seqs):
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = atype
    par.maxResListLen = max_seqs
    par.addBacktrace = 1
    s = api.Search(ctx, par, keys=keys_t)
    pref, aln = [], []
    for i in range(len(q3)):
        hits = s.prefilter(q3[i])
        pref.append("".join(api.format_prefilter_hit(int(keys_t[h["id"]]), int(h["score"]), 0) for h in hits))
        res, bts = s.align(qa[i], q3[i], hits["id"], with_backtrace=True)
        aln.append("".join(s.format_result(res[k:k + 1], bts[k], True) for k in range(len(res))))
    s.close()
    ctx.close()
    return pref, aln


@pytest.mark.parametrize("padded", [False, True])
def test_modules_end_to_end(tmp_path, padded):
    rng = np.random.default_rng(31)
    q3, qa = synth.make_queries(5, seed=41, mean_len=150, lo=60, hi=300)
    db = synth.make_db(700, (q3, qa), seed=42, homologs_per_query=25, mean_len=180, lo=30, hi=600, mask_frac=0.02)
    qkeys = [100 + 7 * i for i in range(len(q3))]
    qdb = str(tmp_path / "query")
    dbio.write_seq_db(qdb, qa, qkeys)
    dbio.write_seq_db(qdb + "_ss", q3, qkeys)
    tdb = str(tmp_path / "target")
    if padded:
        dbio.write_padded_db(tdb, db, "aa")
        dbio.write_padded_db(tdb + "_ss", db, "3di")
        keys_t = np.arange(db.n, dtype=np.uint32)
    else:
        keys_t = (np.arange(db.n) * 3 + 5).astype(np.uint32)
        seqs3, masks, seqsa = [], [], []
        for i in range(db.n):
            raw = db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]]
            seqs3.append(np.where(raw >= 32, raw - 32, raw).astype(np.uint8))
            masks.append(raw >= 32)
            seqsa.append(db.dataaa[db.offsets[i]:db.offsets[i] + db.lengths[i]])
        dbio.write_seq_db(tdb, seqsa, keys_t)
        dbio.write_seq_db(tdb + "_ss", seqs3, keys_t, masks)
    pref, aln = str(tmp_path / "pref"), str(tmp_path / "aln")
    for atype in (2, 0):
        for f in (pref, aln):
            for ext in ("", ".index", ".dbtype"):
                if os.path.exists(f + ext):
                    os.remove(f + ext)
        subprocess.check_call([BIN, "ungappedprefilter", qdb + "_ss", tdb + "_ss", pref, "--max-seqs", "150", "--threads", "2"])
        subprocess.check_call([BIN, "structurealign", qdb, tdb, pref, aln, "--alignment-type", str(atype), "-a", "--threads", "2", "-e", "10"])
        tp, dp = dbio.read_db(pref)
        ta, da = dbio.read_db(aln)
        assert tp & 0xffff == 7 and ta & 0xffff == 5
        epref, ealn = _expected(db, q3, qa, keys_t, atype, 150)
        assert sorted(dp.keys()) == sorted(qkeys) == sorted(da.keys())
        for i, k in enumerate(qkeys):
            assert dp[k].decode() == epref[i]
            assert da[k].decode() == ealn[i]
        assert sum(len(v) for v in da.values()) > 0
