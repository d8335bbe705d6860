"""Host-only modules against the reference BINARY, no GPU needed (runs wherever oracle/_ref_full/bin/foldseek exists: this container and the
GPU box): makepaddedseqdb on a database full of edge cases -- empty entries, one-residue entries, X, soft-masked residues, lengths around the
4-byte padding, headers with database prefixes -- must write the same seven files byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from foldseek_amd import dbio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")
pytestmark = pytest.mark.skipif(not os.path.exists(FS), reason="oracle/_ref_full/bin/foldseek not built")


def _sorted_index(path):
    lines = sorted(open(path).read().splitlines(), key=lambda l: int(l.split()[0]))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("write_lookup", ["1", "0"])
def test_makepaddedseqdb_edge_cases_equal_the_reference_binary(tmp_path, write_lookup):
    rng = np.random.default_rng(3)
    lens = [0, 1, 2, 3, 4, 5, 17, 300, 1, 0, 64, 65, 2000, 33, 7, 8]
    seqs = [rng.integers(0, 21, L).astype(np.uint8) for L in lens]          # code 20 = X
    masks = [rng.random(L) < 0.2 for L in lens]
    keys = [5, 9, 2, 100, 7, 8, 1, 50, 51, 52, 3, 4, 6, 77, 1000000, 12]
    w = str(tmp_path)
    dbio.write_seq_db(os.path.join(w, "t_ss"), seqs, keys, masks)
    heads = ["sp|P%05d|NAME_%d some text" % (k, k) if k % 3 else ("gi|%d|ref|XP_%d.1| hypothetical" % (k, k) if k % 2 else "plain_%d words" % k) for k in keys]
    with open(os.path.join(w, "t_ss_h"), "wb") as f, open(os.path.join(w, "t_ss_h.index"), "w") as fi:
        off = 0
        for k, h in zip(keys, heads):
            b = h.encode() + b"\n\0"
            f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
    np.array([12], np.int32).tofile(os.path.join(w, "t_ss_h.dbtype"))
    _sorted_index(os.path.join(w, "t_ss.index")); _sorted_index(os.path.join(w, "t_ss_h.index"))
    r = subprocess.run([FS, "base:makepaddedseqdb", "t_ss", "ref_pad", "--write-lookup", write_lookup, "--threads", "1", "-v", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    m = subprocess.run([BIN, "makepaddedseqdb", "t_ss", "mine_pad", "--write-lookup", write_lookup, "--threads", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert m.returncode == 0, m.stdout[-2000:]
    for ext in ("", ".index", ".dbtype", ".lookup", "_h", "_h.index", "_h.dbtype"):
        a, b = os.path.join(w, "ref_pad" + ext), os.path.join(w, "mine_pad" + ext)
        assert os.path.exists(a) == os.path.exists(b), ext
        if os.path.exists(a):
            assert open(a, "rb").read() == open(b, "rb").read(), ext
    assert os.path.exists(os.path.join(w, "ref_pad.lookup")) == (write_lookup == "1")


def test_convertalis_hand_written_records_equal_the_reference_binary(tmp_path):
    """alignment records no module run produces but the parser must take like the reference: an empty entry, a record without backtrace next to
    ones with it (start -1), e-values 0 / 1e-301 / 1e+10, a count-less "M", an UNCOMPRESSED backtrace, interleaved single-column gaps; default
    columns in the three BLAST-tab modes, then every alignment / sequence / set column"""
    import shutil
    gold = os.path.join(ROOT, "tests", "golden", "scop_v1")
    w = str(tmp_path)
    for f in ("db", "db.index", "db.dbtype", "db_h", "db_h.index", "db_h.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype", "db.lookup", "db.source"):
        shutil.copy(os.path.join(gold, f), os.path.join(w, f))
    lens = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(os.path.join(w, "db.index"))}
    keys = sorted(lens)

    def rec(t, score, sid, ev, qs, qe, ql, ts, te, tl, bt=None):
        return f"{t}\t{score}\t{sid}\t{ev}\t{qs}\t{qe}\t{ql}\t{ts}\t{te}\t{tl}" + (f"\t{bt}" if bt else "") + "\n"

    def write_aln(with_plain_record):
        q0, q1, q2 = keys[0], keys[1], keys[5]
        e = {q0: rec(keys[0], 2131, "1.00", "1.318E-55", 0, lens[q0] - 1, lens[q0], 0, lens[keys[0]] - 1, lens[keys[0]], f"{lens[q0]}M") +
                 rec(keys[2], 0, "0.000", "0.000E+00", 3, 10, lens[q0], 5, 14, lens[keys[2]], "3M2I3M2D2M") +
                 (rec(keys[3], 17, "0.125", "1.5E+01", -1, 40, lens[q0], -1, 33, lens[keys[3]]) if with_plain_record else ""),
             q1: "",
             q2: rec(keys[4], 55, "0.333", "9.999E-301", 10, 20, lens[q2], 12, 20, lens[keys[4]], "1M1I1M1I1M1I1M1I1M1I1M") +
                 rec(keys[6], 99, "0.999", "1.000E+10", 0, 0, lens[q2], 0, 0, lens[keys[6]], "M") +
                 rec(keys[7], 30000, "0.5", "1e-5", 5, 24, lens[q2], 7, 20, lens[keys[7]], "MMMIIIDDD10M2I1D")}
        blob, idx, off = b"", [], 0
        for k in sorted(e):
            b = e[k].encode() + b"\0"
            idx.append(f"{k}\t{off}\t{len(b)}\n"); blob += b; off += len(b)
        open(os.path.join(w, "aln"), "wb").write(blob)
        open(os.path.join(w, "aln.index"), "w").write("".join(idx))
        np.array([5], np.int32).tofile(os.path.join(w, "aln.dbtype"))

    default = "query,target,fident,alnlen,mismatch,gapopen,qstart,qend,tstart,tend,evalue,bits"
    full = "query,target,pident,nident,qcov,tcov,cigar,qaln,taln,q3dialn,t3dialn,qlen,tlen,qkey,tkey,qheader,theader,qset,tsetid,alnlen,mismatch,gapopen,qseq,t3di"
    for with_plain, cols, modes in ((True, default, ("0", "2", "4")), (False, full, ("0", "4"))):
        write_aln(with_plain)
        for fm in modes:
            par = ["--format-mode", fm, "--format-output", cols, "--threads", "1", "-v", "1"]
            r = subprocess.run([FS, "convertalis", "db", "db", "aln", "ref.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            m = subprocess.run([BIN, "convertalis", "db", "db", "aln", "mine.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert r.returncode == 0 and m.returncode == 0, (r.stdout[-500:], m.stdout[-500:])
            a, b = open(os.path.join(w, "ref.m8"), "rb").read(), open(os.path.join(w, "mine.m8"), "rb").read()
            assert a == b and a.count(b"\n") >= 5, (with_plain, fm)
    # a column list that Util::split leaves empty ("" or ","): the reference falls back to its fixed 12-column line (structureconvertalis.cpp:776-800)
    write_aln(True)
    for empty in ("", ","):
        for fm in ("0", "4"):
            par = ["--format-mode", fm, "--format-output", empty, "--threads", "1", "-v", "1"]
            r = subprocess.run([FS, "convertalis", "db", "db", "aln", "ref.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            m = subprocess.run([BIN, "convertalis", "db", "db", "aln", "mine.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert r.returncode == m.returncode, (r.stdout[-500:], m.stdout[-500:])
            if r.returncode == 0:
                a, b = open(os.path.join(w, "ref.m8"), "rb").read(), open(os.path.join(w, "mine.m8"), "rb").read()
                assert a == b and a.count(b"\n") >= 5, (empty, fm)
    # a record without backtrace + an alignment column: both refuse
    write_aln(True)
    par = ["--format-output", "query,target,qaln", "--threads", "1", "-v", "1"]
    assert subprocess.run([FS, "convertalis", "db", "db", "aln", "ref.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode != 0
    assert subprocess.run([BIN, "convertalis", "db", "db", "aln", "mine.m8"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode != 0


def test_indexdb_without_kmer_table_equals_the_reference_binary(tmp_path):
    """`indexdb <db> <db> --index-subset 2` (the first call of F/data/structureindex.sh: sequence + header DBs and the masked sequence lookup, no
    k-mer table -> no device needed) on a database with empty / one-residue entries, X, lower-case stretches and homopolymer runs around the
    --mask-n-repeat limit: entry by entry the reference binary's index (tests/idx_compare.py says which bytes the reference leaves undefined)"""
    from idx_compare import compare
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 6, 7, 8, 17, 300, 1, 0, 64, 65, 1500, 33, 7, 8]
    seqs = [rng.integers(0, 21, L).astype(np.uint8) for L in lens]
    seqs[7][10:16] = 3; seqs[7][30:37] = 4; seqs[7][100:120] = 20; seqs[12][0:9] = 1; seqs[12][-7:] = 2     # runs of 6 (kept), 7+ (masked), X, at both ends
    masks = [rng.random(L) < 0.15 for L in lens]
    keys = [5, 9, 2, 100, 7, 8, 1, 50, 51, 52, 3, 4, 6, 77, 1000000, 12]
    par = ["--seed-sub-mat", "aa:3di.out,nucl:3di.out", "-k", "0", "--alph-size", "aa:21,nucl:5", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "1",
           "--max-seq-len", "65535", "--max-seqs", "1000", "--mask", "0", "--mask-prob", "0.999995", "--mask-lower-case", "1", "--mask-n-repeat", "6",
           "--spaced-kmer-mode", "1", "-s", "9.5", "--k-score", "seq:2147483647,prof:2147483647", "--check-compatible", "0", "--search-type", "0",
           "--split", "0", "--split-memory-limit", "0", "-v", "1", "--threads", "1", "--index-subset", "2"]
    for side, exe in (("ref", FS), ("mine", BIN)):
        w = os.path.join(str(tmp_path), side)
        os.makedirs(w)
        dbio.write_seq_db(os.path.join(w, "t"), seqs, keys, masks)
        with open(os.path.join(w, "t_h"), "wb") as f, open(os.path.join(w, "t_h.index"), "w") as fi:
            off = 0
            for k in sorted(keys):
                b = f"entry {k}".encode() + b"\n\0"
                f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
        np.array([12], np.int32).tofile(os.path.join(w, "t_h.dbtype"))
        r = subprocess.run([exe, "indexdb", "t", "t"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
    assert compare(os.path.join(str(tmp_path), "ref", "t.idx"), os.path.join(str(tmp_path), "mine", "t.idx")) == []
    # without the header database both stop (the reference: "needs header information")
    for exe in (FS, BIN):
        w = os.path.join(str(tmp_path), "nohdr_" + os.path.basename(exe))
        os.makedirs(w)
        dbio.write_seq_db(os.path.join(w, "t"), seqs, keys, masks)
        assert subprocess.run([exe, "indexdb", "t", "t"] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode != 0


def test_convertalis_reads_sequences_and_headers_out_of_the_reference_written_index(tmp_path):
    """the reference's `createindex` writes <db>.idx / <db>_ss.idx (sequences under DBR1*, headers under HDR1*); with the plain databases moved
    away both binaries must print the same text from inside the index (DbReader::openInsideIndex / openHeaders against the reference's
    IndexReader), equal to what the plain databases gave"""
    import shutil
    gold = os.path.join(ROOT, "tests", "golden", "scop_v1")
    w = str(tmp_path)
    for f in ("db", "db.index", "db.dbtype", "db_h", "db_h.index", "db_h.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype", "db.lookup", "db.source"):
        shutil.copy(os.path.join(gold, f), os.path.join(w, f))
    for e in ("", ".index", ".dbtype"):
        shutil.copy(os.path.join(gold, "db_h" + e), os.path.join(w, "db_ss_h" + e))
    keys = sorted(int(l.split()[0]) for l in open(os.path.join(w, "db.index")))
    lens = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(os.path.join(w, "db.index"))}
    blob, idx, off = b"", "", 0
    for q, t in ((keys[0], keys[1]), (keys[3], keys[3]), (keys[7], keys[2])):
        n = min(lens[q], lens[t]) - 2
        b = f"{t}\t100\t0.5\t1e-5\t1\t{n}\t{lens[q]}\t2\t{n + 1}\t{lens[t]}\t{n}M\n".encode() + b"\0"
        idx += f"{q}\t{off}\t{len(b)}\n"; blob += b; off += len(b)
    open(os.path.join(w, "aln"), "wb").write(blob); open(os.path.join(w, "aln.index"), "w").write(idx)
    np.array([5], np.int32).tofile(os.path.join(w, "aln.dbtype"))
    cols = "query,target,qseq,tseq,qheader,theader,qaln,taln,q3di,t3di,q3dialn,t3dialn,qlen,tlen,cigar"
    subprocess.run([FS, "createindex", "db", "tmp", "--threads", "1", "-v", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)   # no db_ca: stops after both indexes
    assert os.path.exists(os.path.join(w, "db.idx.index")) and os.path.exists(os.path.join(w, "db_ss.idx.index"))
    r = subprocess.run([FS, "convertalis", "db", "db", "aln", "ref_plain.m8", "--threads", "1", "-v", "1", "--format-output", cols], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    os.makedirs(os.path.join(w, "gone"))
    for f in ("db", "db.index", "db.dbtype", "db_h", "db_h.index", "db_h.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype", "db_ss_h", "db_ss_h.index", "db_ss_h.dbtype"):
        shutil.move(os.path.join(w, f), os.path.join(w, "gone", f))
    r = subprocess.run([FS, "convertalis", "db.idx", "db.idx", "aln", "ref_idx.m8", "--threads", "1", "-v", "1", "--format-output", cols], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    m = subprocess.run([BIN, "convertalis", "db.idx", "db.idx", "aln", "mine_idx.m8", "--threads", "1", "--format-output", cols], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert m.returncode == 0, m.stdout[-2000:]
    ref = open(os.path.join(w, "ref_idx.m8"), "rb").read()
    assert ref == open(os.path.join(w, "ref_plain.m8"), "rb").read() and ref.count(b"\n") == 3
    assert open(os.path.join(w, "mine_idx.m8"), "rb").read() == ref
    for f in ("db.idx", "db_ss.idx"):                      # the reference's k-mer tables: 0.4 GB each, not worth keeping in pytest's tmp
        os.remove(os.path.join(w, f))
    shutil.rmtree(os.path.join(w, "tmp"), ignore_errors=True)


@pytest.mark.parametrize("module,pos,flag,value", [
    ("structurealign", ["db", "db", "pref", "out"], "--threads", "0"), ("structurealign", ["db", "db", "pref", "out"], "--threads", "abc"),
    ("structurealign", ["db", "db", "pref", "out"], "-c", "1.5"), ("structurealign", ["db", "db", "pref", "out"], "-c", "0.5"),
    ("structurealign", ["db", "db", "pref", "out"], "--cov-mode", "6"), ("structurealign", ["db", "db", "pref", "out"], "--alignment-type", "4"),
    ("structurealign", ["db", "db", "pref", "out"], "--min-seq-id", "2"), ("structurealign", ["db", "db", "pref", "out"], "-e", "nan"),
    ("structurealign", ["db", "db", "pref", "out"], "--gap-open", "12abc"), ("structurealign", ["db", "db", "pref", "out"], "--gap-open", "aa:3,nucl:1x"),
    ("structurealign", ["db", "db", "pref", "out"], "--gap-open", "aa:3,prot:4"), ("structurealign", ["db", "db", "pref", "out"], "--gap-open", "aa:3"),
    ("structurealign", ["db", "db", "pref", "out"], "--gap-open", "xyz"), ("structurealign", ["db", "db", "pref", "out"], "--gap-extend", "99999999999999999999"),
    ("structurealign", ["db", "db", "pref", "out"], "--alt-ali", "-1"), ("structurealign", ["db", "db", "pref", "out"], "--comp-bias-corr", "2"),
    ("prefilter", ["db_ss", "db_ss", "out"], "--max-seqs", "0"), ("prefilter", ["db_ss", "db_ss", "out"], "--max-seqs", "1.5"),
    ("prefilter", ["db_ss", "db_ss", "out"], "-s", "-1"), ("prefilter", ["db_ss", "db_ss", "out"], "-k", "x"),
    ("prefilter", ["db_ss", "db_ss", "out"], "--split-memory-limit", "12X"), ("prefilter", ["db_ss", "db_ss", "out"], "--mask-n-repeat", "-2"),
    ("convertalis", ["db", "db", "aln", "out.m8"], "--format-mode", "7"), ("convertalis", ["db", "db", "aln", "out.m8"], "--translation-table", "0"),
    ("makepaddedseqdb", ["db_ss", "pad"], "--write-lookup", "2"), ("makepaddedseqdb", ["db_ss", "pad"], "--threads", "1x"),
])
def test_argument_domain_errors_equal_the_reference_binary(tmp_path, module, pos, flag, value):
    """a value outside (or inside) a parameter's pattern: both binaries say "Error in argument <flag>" / "Error in value parsing <flag>" or
    neither does (what happens after a parse that passed -- missing prefilter DB, no device -- is not compared here)"""
    import shutil
    gold = os.path.join(ROOT, "tests", "golden", "scop_v1")
    w = str(tmp_path)
    for f in ("db", "db.index", "db.dbtype", "db_h", "db_h.index", "db_h.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype"):
        shutil.copy(os.path.join(gold, f), os.path.join(w, f))
    ref = subprocess.run([FS, module] + pos + [flag, value, "-v", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    mine = subprocess.run([BIN, module] + pos + [flag, value], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    for msg in (f"Error in argument {flag}", f"Error in value parsing {flag}", f"Error in argument regex {flag}"):
        assert (msg in ref.stdout) == (msg in mine.stdout), (msg, ref.stdout[-300:], mine.stdout[-300:])
    if "Error in" in ref.stdout:
        assert ref.returncode != 0 and mine.returncode != 0
