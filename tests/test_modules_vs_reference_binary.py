"""Module-level cross-check AT SCALE against the reference BINARY (oracle/_ref_full/bin/foldseek, the whole reference built by
oracle/build_ref_full.sh; a built file that travels to the GPU box): both read the SAME on-disk databases (3000 synthetic targets
up to 1200 residues with soft-masked stretches, 48 queries with planted homologs), the reference runs its CPU modules with the
workflow's complete parameter strings, `fsgpu-modules` gets the same argument vectors, and every entry of every result DB must be
byte-identical: k-mer prefilter, ungapped prefilter (plain and reference-padded target), structurealign on both (3Di+AA and 3Di
only, with backtraces), structurerescorediagonal on the defined lines.  The padded target is written by the REFERENCE's
makepaddedseqdb from the ASCII DB.  scop_v1 covers real structures at 30 entries; this covers length classes, masking, row tiles
(queries > 512 residues), score ties at the --max-seqs cut and thousands of alignments.
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
import subprocess

import numpy as np
import pytest

from foldseek_amd import dbio, synth
from test_scop_golden import BIN, GOLD, read_db

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(FS), reason="oracle/_ref_full/bin/foldseek not built")]


def _run(cmd, cwd, env=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0, " ".join(cmd[:4]) + "\n" + r.stdout[-3000:]
    return r.stdout


def _par(name, threads, **over):
    par = list(MANIFEST["runs"][name]["parameters"])
    par[par.index("--threads") + 1] = str(threads)
    for k, v in over.items():
        par[par.index(k) + 1] = str(v)
    return par


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    w = tmp_path_factory.mktemp("refbin")
    q3, qa = synth.make_queries(48, seed=4242, mean_len=260, lo=40, hi=900)      # a few queries beyond 512 residues (row tiles)
    db = synth.make_db(3000, (q3, qa), seed=4243, homologs_per_query=25, mean_len=240, lo=20, hi=1200, mask_frac=0.03)
    tkeys = (np.arange(db.n) * 2 + 11).astype(np.uint32)
    seqs3 = [db.seq(i, "3di", unmask=True) for i in range(db.n)]
    masks = [db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]] >= 32 for i in range(db.n)]
    seqsa = [db.seq(i, "aa") for i in range(db.n)]
    dbio.write_seq_db(str(w / "t"), seqsa, tkeys)
    dbio.write_seq_db(str(w / "t_ss"), seqs3, tkeys, masks)
    qkeys = [7 + 3 * i for i in range(len(q3))]
    dbio.write_seq_db(str(w / "q"), qa, qkeys)
    dbio.write_seq_db(str(w / "q_ss"), q3, qkeys)
    # headers (makepaddedseqdb wants them) and the reference-written padded 3Di target
    hdr = [np.frombuffer(f"s{k}".encode(), np.uint8) for k in tkeys]
    with open(w / "t_ss_h", "wb") as f, open(w / "t_ss_h.index", "w") as fi:
        off = 0
        for k, h in zip(tkeys, hdr):
            b = h.tobytes() + b"\n\0"
            f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
    np.array([12], np.int32).tofile(str(w / "t_ss_h.dbtype"))
    _run([FS, "base:makepaddedseqdb", "t_ss", "tp_ss", "--threads", "1", "-v", "1"], w)
    return w


def _same(w, a, b):
    ta, da = read_db(str(w / a))
    tb, dbb = read_db(str(w / b))
    assert ta == tb and sorted(da) == sorted(dbb), (a, b)
    for k in sorted(da):
        assert da[k] == dbb[k], f"{a} vs {b}: entry {k}\n{da[k][:400]!r}\n{dbb[k][:400]!r}"
    return sum(v.count(b"\n") for v in da.values())


def test_prefilters_and_alignments_equal_the_reference_binary(world):
    w = world
    lines = {}
    # ---- prefilters ----
    for name, mod, tgt, par in (("kmer", "prefilter", "t_ss", _par("pref_kmer", 8, **{"--max-seqs": "300"})),
                                ("ung", "ungappedprefilter", "t_ss", _par("pref_ung", 8, **{"--max-seqs": "300"})),
                                ("ungp", "ungappedprefilter", "tp_ss", _par("pref_ung", 8, **{"--max-seqs": "300"}))):
        _run([FS, mod, "q_ss", tgt, "ref_" + name] + par, w)
        _run([BIN, mod, "q_ss", tgt, "mine_" + name] + par, w)
        lines[name] = _same(w, "ref_" + name, "mine_" + name)
    assert lines["kmer"] > 3000 and lines["ung"] > 10000 and lines["ungp"] == lines["ung"]
    # ---- structurealign on the reference's prefilter output ----
    for name, pref, atype in (("aln2", "ref_kmer", 2), ("aln0", "ref_ung", 0)):
        par = _par("aln_t2_a", 8, **{"--alignment-type": atype})
        _run([FS, "structurealign", "q", "t", pref, "ref_" + name] + par, w)
        _run([BIN, "structurealign", "q", "t", pref, "mine_" + name] + par, w)
        lines[name] = _same(w, "ref_" + name, "mine_" + name)
    assert lines["aln2"] > 800 and lines["aln0"] > 800
    # ---- fused search == the reference's two steps ----
    _run([BIN, "search", "q", "t", "mine_fused", "mine_fused_pref", "--prefilter-mode", "0", "-a", "1", "--alignment-type", "2", "--sort-by-structure-bits", "0",
          "--max-seqs", "300", "-s", "9.5", "-e", "10", "--threads", "3"], w)
    _same(w, "ref_kmer", "mine_fused_pref")
    _same(w, "ref_aln2", "mine_fused")


def test_rescorediagonal_equals_the_reference_binary_on_defined_lines(world):
    """the k-mer prefilter's (target, diagonal) lines restricted to the pairs the reference defines (fsgpu_diag.hip): both binaries"""
    w = world
    if not os.path.exists(w / "ref_kmer"):
        _run([FS, "prefilter", "q_ss", "t_ss", "ref_kmer"] + _par("pref_kmer", 8, **{"--max-seqs": "300"}), w)
    qlen = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(w / "q_ss.index")}
    tlen = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(w / "t_ss.index")}
    _, pref = read_db(str(w / "ref_kmer"))
    blob, index, off, kept = b"", [], 0, 0
    for q in sorted(pref):
        keep = []
        for ln in pref[q].decode().splitlines():
            t, _, d = ln.split("\t")
            d, lq, lt = int(d), qlen[q], tlen[int(t)]
            if (d >= 0 and d < lq) or (d < 0 and -d < lt and lt <= lq):
                keep.append(ln)
        kept += len(keep)
        body = ("\n".join(keep) + "\n" if keep else "").encode() + b"\0"
        index.append(f"{q}\t{off}\t{len(body)}\n"); blob += body; off += len(body)
    open(w / "pref_def", "wb").write(blob); open(w / "pref_def.index", "w").write("".join(index))
    np.array([7], np.int32).tofile(str(w / "pref_def.dbtype"))
    assert kept > 1000
    par = _par("resc_t2_a", 8)
    _run([FS, "structurerescorediagonal", "q", "t", "pref_def", "ref_resc"] + par, w)
    _run([BIN, "structurerescorediagonal", "q", "t", "pref_def", "mine_resc"] + par, w)
    assert _same(w, "ref_resc", "mine_resc") > 300


@pytest.mark.parametrize("run", ["aln_t2_a_cov1", "aln_t2_a_cov3", "aln_t2_a_cov5", "aln_t2_a_sid2", "aln_t2_a_minlen", "aln_t0_a_gap", "aln_t2_a_altali",
                                 "aln_t2_a_cbs", "aln_t2_a_e001_c08", "aln_t2_a_maxacc", "pref_kmer_c07m5", "pref_kmer_nospace", "pref_kmer_s75",
                                 "pref_kmer_minung45", "pref_ung_max7", "pref_ung_nocb"])
def test_parameter_variants_equal_the_reference_binary_at_scale(world, run):
    """the parameter variants frozen for the SCOP example set (tests/golden/make_scop_golden.py), here on the 3000-target / 48-query
    databases with both binaries side by side: acceptance criteria, coverage and sequence-identity modes, gap costs, alternative
    alignments, k-mer options"""
    w = world
    spec = MANIFEST["runs"][run]
    par = _par(run, 8)
    if spec["module"] == "structurealign":
        if not os.path.exists(w / "ref_kmer"):
            _run([FS, "prefilter", "q_ss", "t_ss", "ref_kmer"] + _par("pref_kmer", 8, **{"--max-seqs": "300"}), w)
        pos = ["q", "t", "ref_kmer"]
    else:
        pos = ["q_ss", "t_ss"]
    _run([FS, spec["module"]] + pos + ["ref_" + run] + par, w)
    _run([BIN, spec["module"]] + pos + ["mine_" + run] + par, w)
    assert _same(w, "ref_" + run, "mine_" + run) > 50


EDGE_CASES = [
    ("structurealign", "aln_t2_a", {"-e": "0"}), ("structurealign", "aln_t2_a", {"-e": "1e-300"}), ("structurealign", "aln_t2_a", {"-e": "1e+30"}),
    ("structurealign", "aln_t2_a", {"--max-accept": "1"}), ("structurealign", "aln_t2_a", {"--max-rejected": "1"}),
    ("structurealign", "aln_t2_a", {"-c": "1.0"}), ("structurealign", "aln_t2_a", {"--min-seq-id": "1.0"}),
    ("structurealign", "aln_t2_a", {"--min-aln-len": "100000"}), ("structurealign", "aln_t2_a", {"--alt-ali": "5"}),
    ("structurealign", "aln_t2_a", {"--add-self-matches": "1"}), ("structurealign", "aln_t2_a", {"--gap-open": "aa:2,nucl:2", "--gap-extend": "aa:1,nucl:1"}),
    ("prefilter", "pref_kmer", {"--max-seqs": "1"}), ("prefilter", "pref_kmer", {"--max-seqs": "100000"}), ("prefilter", "pref_kmer", {"-s": "1"}),
    ("prefilter", "pref_kmer", {"--k-score": "seq:200,prof:200"}), ("prefilter", "pref_kmer", {"--min-ungapped-score": "0"}),
    ("prefilter", "pref_kmer", {"--min-ungapped-score": "255"}), ("prefilter", "pref_kmer", {"--mask-n-repeat": "2"}),
    ("ungappedprefilter", "pref_ung", {"--max-seqs": "1"}), ("ungappedprefilter", "pref_ung", {"--max-seqs": "100000"}),
    ("ungappedprefilter", "pref_ung", {"--min-ungapped-score": "0"}), ("ungappedprefilter", "pref_ung", {"--min-ungapped-score": "254"}),
    ("ungappedprefilter", "pref_ung", {"--comp-bias-corr-scale": "1.0"}),
]


@pytest.mark.parametrize("module,base,over", EDGE_CASES, ids=[f"{m}:{' '.join(f'{k}={v}' for k, v in o.items())}" for m, _, o in EDGE_CASES])
def test_edge_values_behave_like_the_reference_binary(world, module, base, over):
    """extreme but legal parameter values: where the reference produces a result DB ours must be byte-identical, where the reference
    refuses or dies ours must not succeed either"""
    w = world
    tag = module[:4] + "_" + "_".join(f"{k.strip('-')}{v}".replace(":", "").replace(",", "").replace("+", "p") for k, v in over.items())
    par = _par(base, 8, **over)
    if module == "structurealign":
        if not os.path.exists(w / "ref_kmer"):
            _run([FS, "prefilter", "q_ss", "t_ss", "ref_kmer"] + _par("pref_kmer", 8, **{"--max-seqs": "300"}), w)
        pos = ["q", "t", "ref_kmer"]
    else:
        pos = ["q_ss", "t_ss"]
    ref = subprocess.run([FS, module] + pos + ["ref_" + tag] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    mine = subprocess.run([BIN, module] + pos + ["mine_" + tag] + par, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if ref.returncode != 0:
        assert mine.returncode != 0, f"reference failed ({ref.stdout[-300:]}) but this module succeeded"
        return
    # (prefilter --min-ungapped-score 0 was refused until round 4; the score-0 elements a cut of 0 lets through are emitted now:
    # tests/test_kmer_gpu.py::test_cut_zero_with_diagonal_scores_equals_the_compiled_reference)
    assert mine.returncode == 0, mine.stdout[-1500:]
    _same(w, "ref_" + tag, "mine_" + tag)


def test_convertalis_equals_the_reference_binary_at_scale(world):
    """convertalis of a few thousand alignments (with backtraces): default columns, every sequence / alignment column, format modes 2 and 4"""
    w = world
    if not os.path.exists(w / "ref_aln2"):
        if not os.path.exists(w / "ref_kmer"):
            _run([FS, "prefilter", "q_ss", "t_ss", "ref_kmer"] + _par("pref_kmer", 8, **{"--max-seqs": "300"}), w)
        _run([FS, "structurealign", "q", "t", "ref_kmer", "ref_aln2"] + _par("aln_t2_a", 8), w)
    for name, keys in (("q", [7 + 3 * i for i in range(48)]), ("t", [int(l.split()[0]) for l in open(w / "t.index")])):
        if os.path.exists(w / f"{name}_h"):
            continue
        with open(w / f"{name}_h", "wb") as f, open(w / f"{name}_h.index", "w") as fi:
            off = 0
            for k in keys:
                b = f"{name}{k} some description {k % 7}".encode() + b"\n\0"
                f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
        np.array([12], np.int32).tofile(str(w / f"{name}_h.dbtype"))
    conv = MANIFEST["convert_runs"]
    for tag, par in (("default", conv["conv_default.m8"]["parameters"]), ("all", conv["conv_fmt4_all.m8"]["parameters"]), ("fmt2", conv["conv_fmt2.m8"]["parameters"])):
        _run([FS, "convertalis", "q", "t", "ref_aln2", f"ref_{tag}.m8"] + par, w)
        _run([BIN, "convertalis", "q", "t", "ref_aln2", f"mine_{tag}.m8"] + par, w)
        a, b = open(w / f"ref_{tag}.m8", "rb").read(), open(w / f"mine_{tag}.m8", "rb").read()
        assert len(a) > 50000 and a == b, tag


CASCADE = [("-s", "1", "--max-seqs", "100", "--diag-score", "0", "--min-ungapped-score", "0"),
           ("-s", "4.5", "--max-seqs", "200", "--diag-score", "1", "--min-ungapped-score", "30"),
           ("-s", "8", "--max-seqs", "1000", "--diag-score", "1", "--min-ungapped-score", "30")]


@pytest.mark.parametrize("step", [0, 1, 2])
def test_cluster_cascade_prefilter_steps_equal_the_reference_binary(world, step):
    """the three prefilter calls of the cluster workflow's cascade (F/data/structurecluster.sh, parameter strings as `foldseek cluster -v 3` prints
    them): all-vs-all of the 3000-target 3Di database at -s 1 (--diag-score 0: k-mer match counts as scores, cut 0), -s 4.5 and -s 8, -c 0.8,
    no composition bias correction, self matches added -- both binaries, every entry byte-identical"""
    w = world
    par = ["--sub-mat", "aa:3di.out,nucl:3di.out", "--seed-sub-mat", "aa:3di.out,nucl:3di.out", "-k", "0", "--target-search-mode", "0",
           "--k-score", "seq:2147483647,prof:2147483647", "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535", "--split", "0", "--split-mode", "2",
           "--split-memory-limit", "0", "-c", "0.8", "--cov-mode", "0", "--comp-bias-corr", "0", "--comp-bias-corr-scale", "1", "--exact-kmer-matching", "0",
           "--mask", "0", "--mask-prob", "0.999995", "--mask-lower-case", "1", "--mask-n-repeat", "6", "--add-self-matches", "1", "--spaced-kmer-mode", "1",
           "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800", "--threads", "8", "--compressed", "0",
           "-v", "1"] + list(CASCADE[step])
    _run([FS, "prefilter", "t_ss", "t_ss", f"ref_casc{step}"] + par, w)
    _run([BIN, "prefilter", "t_ss", "t_ss", f"mine_casc{step}"] + par, w)
    assert _same(w, f"ref_casc{step}", f"mine_casc{step}") >= 3000          # at least every self match
