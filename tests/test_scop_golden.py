"""Drop-in tests on REAL data and REFERENCE-WRITTEN databases (tests/golden/scop_v1, generator make_scop_golden.py):

the reference's own `createdb` / `makepaddedseqdb` wrote the sequence DBs from F/example (25 SCOP domains + 1tim/8tim),
its `prefilter` / `ungappedprefilter` / `structurealign` wrote the result DBs -- here `fsgpu-modules` runs with the SAME
positional arguments and the SAME complete parameter strings (what F/data/structuresearch.sh hands the modules) on those
DBs and every entry of every result DB must be byte-identical.  Nothing in these tests is written by foldseek_amd/dbio.py.

WHAT "byte-identical to the reference" covers here.  The reference binary these result DBs come from (oracle/_ref_full, and oracle/_ref for
the library-level tests) cannot link the upstream Rust block-aligner (no cargo in this image): oracle/patch_ref_full.py links this
repository's own foldseek_amd/csrc/host/block_aligner.cpp for the `block_*` symbols.  So in structurealign records the
BACKTRACE-DERIVED columns -- qStart, dbStart, the CIGAR / backtrace string, seqId, alnLen -- compare this repository's restatement of
the block aligner with ITSELF (run once by the reference's caller, once by ours): they pin the caller-side logic (which rectangle is
aligned, block sizes, the acceptance rule, reversal and offsets), not the crate's tie-breaking.  The prefilter DBs and the SW columns
(score, qEnd, dbEnd, e-value, coverage gates, result order) are independent of this repository.  What pins the aligner itself:
tests/test_block_aligner.py (the crate's own known answers, an independent model of its tie rules, optimality by re-scoring) and
oracle/ba_kat (vectors to run against the crate wherever cargo exists).
"""
import json
import os
import shutil
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")
GOLD = os.path.join(ROOT, "tests", "golden", "scop_v1")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))

# runs whose parameters ask for something the device path refuses (with a message) instead of computing it
NOT_IMPLEMENTED = {}


def read_db(path):
    """(dbtype int32, {key: entry bytes without the terminator}) straight from the files, no project code involved"""
    t = int.from_bytes(open(path + ".dbtype", "rb").read(4), "little", signed=True)
    if os.path.exists(path):
        data = open(path, "rb").read()
    else:      # DBWriter with several threads leaves the data in <db>.0, <db>.1, ... (offsets run through their concatenation)
        data, k = b"", 0
        while os.path.exists(f"{path}.{k}"):
            data += open(f"{path}.{k}", "rb").read()
            k += 1
    out = {}
    for line in open(path + ".index"):
        k, off, ln = line.split()
        out[int(k)] = data[int(off):int(off) + int(ln) - 1]
    return t, out


@pytest.fixture()
def scop(tmp_path):
    """a private copy of the frozen reference DBs with the links makepaddedseqdb makes (AA / header data -> source DB)"""
    work = tmp_path / "scop"
    shutil.copytree(GOLD, work)
    for link, target in MANIFEST["links"].items():
        os.symlink(str(work / target), str(work / link))
    return work


def test_makepaddedseqdb_equals_reference_written_db(scop):
    """host-only module: base:makepaddedseqdb of the reference on db_ss (+ headers) == ours, all seven files byte for byte
    (M/src/util/makepaddedseqdb.cpp:14-154; the workflow links <db>_h next to the 3Di DB first, F/data/makepaddeddb.sh:10-19)"""
    for ext in ("", ".index", ".dbtype"):
        if not os.path.exists(scop / ("db_ss_h" + ext)):
            shutil.copy(scop / ("db_h" + ext), scop / ("db_ss_h" + ext))
    subprocess.check_call([BIN, "makepaddedseqdb", str(scop / "db_ss"), str(scop / "mine_ss"), "--threads", "1", "-v", "1"])
    for ext in ("", ".index", ".lookup", ".dbtype", "_h", "_h.index", "_h.dbtype"):
        assert open(scop / ("mine_ss" + ext), "rb").read() == open(scop / ("db_pad_ss" + ext), "rb").read(), ext


@pytest.mark.parametrize("module,args,needle", [
    ("prefilter", ["--no-such-flag", "1"], 'Unrecognized parameter "--no-such-flag"'),
    ("prefilter", ["--mask", "1"], "--mask 1 is not implemented"),
    ("prefilter", ["-k", "7"], "-k 7 is not implemented"),
    ("prefilter", ["--sub-mat", "aa:blosum62.out,nucl:nucleotide.out"], "--sub-mat"),
    ("prefilter", ["--exact-kmer-matching", "1"], "not implemented"),
    ("ungappedprefilter", ["--prefilter-mode", "2"], "not implemented"),
    ("ungappedprefilter", ["--max-seqs"], "Missing argument --max-seqs"),
    ("structurealign", ["--alignment-type", "1"], "--alignment-type 1 is not implemented"),
    ("structurealign", ["--tmscore-threshold", "0.5"], "not implemented"),
    ("structurealign", ["--realign", "1"], "not implemented"),
    ("structurealign", ["-a", "maybe"], "Invalid boolean string maybe"),
    ("structurealign", ["--compressed", "1"], "--compressed 1 is not implemented"),
])
def test_modules_refuse_what_they_do_not_implement(scop, module, args, needle):
    """a drop-in must not swallow flags: unknown ones get the reference's message (Parameters.cpp:2087), known ones whose
    value selects an unimplemented feature are refused -- before any device is touched, so this runs without a GPU"""
    pos = [str(scop / "db_ss"), str(scop / "db_ss"), str(scop / "out")] if module != "structurealign" else \
          [str(scop / "db"), str(scop / "db"), str(scop / "pref_kmer"), str(scop / "out")]
    r = subprocess.run([BIN, module] + pos + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    assert needle in r.stderr
    assert not os.path.exists(scop / "out.index")


# every frozen run with the host block aligner; the runs of the modules that compute backtraces once more with the device block aligner
_RUNS = [(n, 0) for n in sorted(MANIFEST["runs"])] + [(n, 1) for n in sorted(MANIFEST["runs"]) if MANIFEST["runs"][n]["module"] in ("structurealign", "search")]


@pytest.mark.gpu
@pytest.mark.parametrize("name,dev", _RUNS)
def test_module_equals_reference_result_db(scop, name, dev):
    """dev: FSGPU_DEVICE_BACKTRACE -- with 1 the CIGARs of the frozen reference-binary outputs are met by the device block aligner (k_block_backtrace), with 0
    by the host restatement; runs without backtraces (prefilter modules, rescorediagonal) are run once"""
    run = MANIFEST["runs"][name]
    out = str(scop / ("mine_%d_" % dev + name))
    cmd = [BIN, run["module"]] + [str(scop / p) for p in run["positional"]] + [out] + run["parameters"]
    env = dict(os.environ, FSGPU_DEVICE_BACKTRACE=str(dev), FSGPU_BT_PASS2="1", FSGPU_MODULE_TIMING="1")      # PASS2: the device's 512-row pass runs however few hits reach it
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    if name in NOT_IMPLEMENTED:
        assert r.returncode == 1 and NOT_IMPLEMENTED[name] in r.stderr
        return
    assert r.returncode == 0, r.stderr
    want_t, want = read_db(str(scop / name))
    got_t, got = read_db(out)
    assert got_t == want_t
    assert sorted(got) == sorted(want)
    for k in sorted(want):
        assert got[k] == want[k], f"{name}: entry {k}\nwant {want[k][:300]!r}\ngot  {got[k][:300]!r}"
    assert sum(len(v) for v in want.values()) > 0
    if dev == 1 and run["module"] == "structurealign" and "-a" in run["parameters"] and "--alt-ali" not in run["parameters"]:
        m = re.search(r"backtrace [0-9.]+ \((\d+) of (\d+) on the device\)", r.stderr)
        assert m and int(m.group(2)) > 0 and int(m.group(1)) >= 0.95 * int(m.group(2)), r.stderr[-400:]


@pytest.mark.gpu
def test_fused_search_equals_reference_two_step(scop):
    """`search` (prefilter + structurealign in one process) on the reference-written DBs == the reference's two modules"""
    for mode, pref, aln in ((0, "pref_kmer", "aln_t2_a"), (1, "pref_ung", "aln_t2_a_ung")):
        out, outp = str(scop / f"mine_search{mode}"), str(scop / f"mine_search{mode}_pref")
        cmd = [BIN, "search", str(scop / "db"), str(scop / "db"), out, outp, "--prefilter-mode", str(mode), "-a", "1",
               "--alignment-type", "2", "--sort-by-structure-bits", "0", "--threads", "2", "-s", "9.5", "--max-seqs", "1000", "-e", "10"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert read_db(outp) == read_db(str(scop / pref))
        assert read_db(out) == read_db(str(scop / aln))


@pytest.mark.gpu
def test_search_then_convertalis_equals_reference_m8(scop):
    """easy-search without the structure parsing: `search` + `convertalis` on the reference-written sequence DBs == the .m8 the
    reference's prefilter + structurealign + convertalis produce (the search runs with 3 host threads, so the entries of its alignment DB
    are written in key order whatever the thread interleaving was)"""
    aln, m8 = str(scop / "mine_aln"), str(scop / "mine.m8")
    r = subprocess.run([BIN, "search", str(scop / "db"), str(scop / "db"), aln, "--prefilter-mode", "0", "-a", "1", "--alignment-type", "2",
                        "--sort-by-structure-bits", "0", "--threads", "3", "-s", "9.5", "--max-seqs", "1000", "-e", "10"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([BIN, "convertalis", str(scop / "db"), str(scop / "db"), aln, m8] + MANIFEST["convert_runs"]["conv_default.m8"]["parameters"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert open(m8, "rb").read() == open(scop / "conv_default.m8", "rb").read()


@pytest.mark.gpu
def test_rescorediagonal_undefined_pairs_are_refused_or_skipped(scop):
    """pref_kmer holds 297 (query, target, diagonal) lines whose reference result is undefined (negative diagonal, target longer
    than the query: the reference's reverse pass reads past the query, structurerescorediagonal.cpp:96-99).  Default: the module
    says so and fails; --undefined-diagonals skip drops exactly those lines == the reference run on the defined lines."""
    run = MANIFEST["runs"]["resc_t2_a"]
    pos = [str(scop / "db"), str(scop / "db"), str(scop / "pref_kmer")]
    r = subprocess.run([BIN, "structurerescorediagonal"] + pos + [str(scop / "mine_u1")] + run["parameters"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "undefined" in r.stderr and "structurerescorediagonal.cpp:96-99" in r.stderr
    out = str(scop / "mine_u2")
    r = subprocess.run([BIN, "structurerescorediagonal"] + pos + [out] + run["parameters"] + ["--undefined-diagonals", "skip"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert read_db(out) == read_db(str(scop / "resc_t2_a"))


FS_REF = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FS_REF), reason="oracle/_ref_full/bin/foldseek not built (oracle/build_ref_full.sh cpu)")
def test_modules_take_precomputed_indexes_as_targets(scop):
    """SURVEY 8f rank 4.  Once `createindex` has run, the search workflow hands the modules the INDEX as target
    (`prefilter db_ss db_ss.idx`, `structurealign db db.idx`, F/data/structuresearch.sh "${TARGET_PREFILTER}${INDEXEXT}").  The indexes
    are written here by the reference binary (`indexdb`, 900 MB each -- not frozen); our modules must produce the result DBs the
    reference produced with them (note: with an index target the query and target DB names differ, so the self hit is no longer
    forced to the top -- pref_kmer_idx differs from pref_kmer in exactly that).  Second pass: the plain target DBs are DELETED,
    the sequences then come out of the index files themselves (DBR1INDEX / DBR1DATA entries)."""
    # query side under its own name, so that the target's plain files can be removed later
    for f in os.listdir(scop):
        if f.startswith("db") and not f.startswith("db_pad") and not os.path.islink(scop / f):
            shutil.copy(scop / f, scop / ("q" + f))
    for cmd in MANIFEST["indexdb"]:
        r = subprocess.run([FS_REF] + cmd, cwd=scop, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
    runs = MANIFEST["runs_with_index"]
    for attempt in ("plain DB next to the index", "index files only"):
        for name in ("pref_kmer_idx", "aln_t2_a_idx"):
            run = runs[name]
            pos = ["q" + run["positional"][0]] + run["positional"][1:]
            out = str(scop / f"mine_{name}_{attempt[:5].strip()}")
            cmd = [BIN, run["module"]] + [str(scop / p) for p in pos] + [out] + run["parameters"]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, (attempt, name, r.stderr)
            assert read_db(out) == read_db(str(scop / name)), (attempt, name)
        for f in ("db", "db.index", "db.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype"):     # second pass: sequences from inside the indexes
            if os.path.exists(scop / f):
                os.remove(scop / f)


def _frozen_text(scop, name):
    import gzip
    p = scop / name
    return open(p, "rb").read() if os.path.exists(p) else gzip.open(str(p) + ".gz", "rb").read()


@pytest.mark.parametrize("name", sorted(MANIFEST["convert_runs"]))
def test_convertalis_equals_reference_output(scop, name):
    """`convertalis` (host-only text formatting, runs without a GPU) on the reference-written alignment DBs with the reference's
    positional arguments and parameter strings: the output FILE must be byte-identical (BLAST-tab, with lengths, with column
    headers and every supported --format-output column; alignments without backtrace; padded target; --db-output 1)"""
    run = MANIFEST["convert_runs"][name]
    out = str(scop / ("mine_" + name))
    cmd = [BIN, run["module"]] + [str(scop / p) for p in run["positional"]] + [out] + run["parameters"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    if "--db-output" in run["parameters"] and run["parameters"][run["parameters"].index("--db-output") + 1] == "1":
        want_t, want = read_db(str(scop / name))
        got_t, got = read_db(out)
        assert (got_t, sorted(got)) == (want_t, sorted(want))
        assert all(got[k] == want[k] for k in want)
        return
    want, got = _frozen_text(scop, name), open(out, "rb").read()
    assert len(want) > 1000
    if got != want:
        for i, (a, b) in enumerate(zip(want.split(b"\n"), got.split(b"\n"))):
            assert a == b, f"{name}: line {i + 1}\nwant {a[:400]!r}\ngot  {b[:400]!r}"
    assert got == want
    assert not os.path.exists(out + ".index") and not os.path.exists(out + ".dbtype")      # a plain file, like the reference leaves it


@pytest.mark.parametrize("args,needle", [
    (["--format-output", "query,target,lddt"], "column lddt is not implemented on this path"),
    (["--format-output", "query,target,alntmscore"], "column alntmscore is not implemented"),
    (["--format-output", "query,nosuchcolumn"], "Format code nosuchcolumn does not exist."),
    (["--format-mode", "3"], "--format-mode 3 is not implemented on the device path"),
    (["--format-mode", "1"], "--format-mode 1 is not implemented"),
    (["--no-such-flag", "1"], 'Unrecognized parameter "--no-such-flag"'),
])
def test_convertalis_refuses_what_it_does_not_implement(scop, args, needle):
    r = subprocess.run([BIN, "convertalis", str(scop / "db"), str(scop / "db"), str(scop / "aln_t2_a"), str(scop / "out.m8")] + args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and needle in r.stderr, r.stderr
    assert not os.path.exists(scop / "out.m8")


def test_convertalis_needs_the_backtrace_for_alignment_columns(scop):
    r = subprocess.run([BIN, "convertalis", str(scop / "db"), str(scop / "db"), str(scop / "aln_t2"), str(scop / "out.m8"), "--format-output", "query,target,qaln,taln"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "Backtrace cigar is missing in the alignment result" in r.stderr


def test_convertalis_prob_column(scop):
    """`prob` = CalcProbTP::calculate(bits) (F/src/commons/CalcProbTP.h:8-32): 0 up to 10 bits, 1 from 100 bits, the fitted gamma
    mixture ratio in between -- checked against a double-precision restatement (the reference computes it in float: 3 printed decimals agree
    to +-0.001).  The reference opens the _ca DB for this column although the value only depends on the score; this module does not need it."""
    import math
    out = str(scop / "prob.m8")
    r = subprocess.run([BIN, "convertalis", str(scop / "db"), str(scop / "db"), str(scop / "aln_t2_a"), out, "--format-output", "bits,prob"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr

    def gamma_pdf(alpha, beta, x):
        return math.exp(alpha * math.log(beta) + (alpha - 1) * math.log(x) - beta * x - math.lgamma(alpha))

    def want(score):
        if score <= 10:
            return 0.0
        if score >= 100:
            return 1.0
        tp = (0.8279 * gamma_pdf(1.8123, 1 / 46.0042, score) + 0.1721 * gamma_pdf(1.0057, 1 / 563.5014, score)) * 0.1023
        fp = (0.34 * gamma_pdf(4.9259, 1 / 4.745, score) + 0.66 * gamma_pdf(9.4834, 1 / 1.3136, score)) * 0.8977
        return 1 / (1 + fp / tp)
    rows = [l.split("\t") for l in open(out).read().splitlines()]
    assert len(rows) > 500
    seen_mid = 0
    for bits, prob in rows:
        w = want(int(bits))
        assert abs(float(prob) - w) <= 0.0011, (bits, prob, w)
        seen_mid += 10 < int(bits) < 100
    assert seen_mid > 100
