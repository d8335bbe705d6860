"""The INTEGRATION.md 2 / 2b adapters compiled INTO the reference (oracle/patch_ref_full.py -> oracle/_ref_full/adapter_hunks.diff,
foldseek_amd/csrc/host/adapters/*.inc): the reference binary `foldseek-fsgpu` keeps its own Parameters / DBReader / DBWriter /
workflows and hands the per-query work of `structurealign --gpu 1` and `prefilter --gpu 1` to libfsgpu.so.

Checked on the MI355X: (1) the two modules with the frozen parameter strings of tests/golden/scop_v1 + `--gpu 1` against the result
DBs the reference's CPU modules wrote; (2) `easy-search` FROM STRUCTURE FILES (createdb, makepaddedseqdb, ungappedprefilter --gpu 1
through our Marv, structurealign --gpu 1 through the adapter incl. the reference's own TM-score / LDDT re-ranking, convertalis):
the .m8 must equal the one the CPU reference binary writes for the same command without --gpu, byte for byte; (3) the same with the
k-mer prefilter on the device (`--prefilter-mode 0 --gpu 1`)."""
import os
import subprocess

import pytest

from test_scop_golden import MANIFEST, read_db, scop  # noqa: F401  (scop is a fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS_GPU = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek-fsgpu")
FS_CPU = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
EXAMPLES = os.path.join(ROOT, "tests", "golden", "example_structures")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (os.path.exists(FS_GPU) and os.path.exists(FS_CPU)),
                                                  reason="oracle/_ref_full/bin/foldseek{,-fsgpu} not built (oracle/build_ref_full.sh all)")]


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "foldseek_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    return env


def _run(cmd, cwd=None, timeout=900):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=_env(), cwd=cwd, timeout=timeout)
    assert r.returncode == 0, (" ".join(cmd[:6]), r.stdout[-3000:])
    return r.stdout


def _params(spec):
    """the frozen parameter string of a run with --gpu 1 and the verbosity raised to 3 (the device path announces itself at INFO level)"""
    par = list(spec["parameters"])
    if "-v" in par:
        par[par.index("-v") + 1] = "3"
    else:
        par += ["-v", "3"]
    return par + ["--gpu", "1"]


@pytest.fixture()
def padded_target(tmp_path):
    """target databases written by the REFERENCE from the structure files: createdb + makepaddedseqdb (CPU binary).  The reference's
    makepaddedseqdb workflow replaces a symlinked header DB through FileUtil::move, which takes "st_dev of file != st_dev of directory" for
    "another file system" and then COPIES through the symlink, truncating the source headers (M/src/commons/FileUtil.cpp:364-398);
    overlayfs (the container's /tmp) reports such pairs, so `easy-search <dir> --gpu 1` loses its target headers there whatever runs
    the kernels.  The header data file is therefore saved and restored around that one step."""
    import shutil
    w = str(tmp_path)
    _run([FS_CPU, "createdb", EXAMPLES, "target", "--threads", "2", "-v", "1"], cwd=w)
    shutil.copy(os.path.join(w, "target_h"), os.path.join(w, "target_h.saved"))
    _run([FS_CPU, "makepaddedseqdb", "target", "target_pad", "--threads", "2", "-v", "1"], cwd=w)
    shutil.copy(os.path.join(w, "target_h.saved"), os.path.join(w, "target_h"))
    return w


ALN_RUNS = [k for k, v in MANIFEST["runs"].items() if v["module"] == "structurealign"]
KMER_RUNS = [k for k, v in MANIFEST["runs"].items() if v["module"] == "prefilter"]


@pytest.mark.parametrize("run", ALN_RUNS)
def test_reference_structurealign_gpu1_equals_its_cpu_result(scop, run):
    """every frozen structurealign parameter set (alignment types, -a, gap costs, coverage modes, e-value / coverage / identity / length
    thresholds, --max-accept / --max-rejected, --alt-ali, padded targets): the reference module with --gpu 1 == the DB its CPU path wrote"""
    spec = MANIFEST["runs"][run]
    out = str(scop / ("adapter_" + run))
    _run([FS_GPU, "structurealign"] + [str(scop / x) for x in spec["positional"]] + [out] + _params(spec))
    assert read_db(out) == read_db(str(scop / run))


@pytest.mark.parametrize("run", KMER_RUNS)
def test_reference_prefilter_gpu1_equals_its_cpu_result(scop, run):
    """every frozen k-mer prefilter parameter set (-s, --max-seqs, -c / --cov-mode, composition bias, masking, spaced k-mers,
    --min-ungapped-score, padded targets): `prefilter --gpu 1` (index table + matchQuery on the device) == the DB the CPU path wrote"""
    spec = MANIFEST["runs"][run]
    out = str(scop / ("adapter_" + run))
    log = _run([FS_GPU, "prefilter"] + [str(scop / x) for x in spec["positional"]] + [out] + _params(spec))
    assert "Index table (device)" in log, log[-1500:]          # the device path really ran (no silent CPU fallback)
    assert read_db(out) == read_db(str(scop / run))


RESC_RUNS = [k for k, v in MANIFEST["runs"].items() if v["module"] == "structurerescorediagonal"]


@pytest.mark.parametrize("run", RESC_RUNS)
def test_reference_structurerescorediagonal_gpu1_equals_its_cpu_result(scop, run):
    """the frozen single-diagonal rescoring runs (linclust hand-off of easy-cluster; 3Di and 3Di+AA, with and without backtrace, the cluster
    workflow's -e 0.01 -c 0.8): the reference module with --gpu 1 (k_diag_rescore behind the adapter) == the DB its CPU path wrote"""
    spec = MANIFEST["runs"][run]
    out = str(scop / ("adapter_" + run))
    log = _run([FS_GPU, "structurerescorediagonal"] + [str(scop / x) for x in spec["positional"]] + [out] + _params(spec))
    assert "Diagonal rescoring (device)" in log, log[-1500:]
    assert read_db(out) == read_db(str(scop / run))


def test_reference_prefilter_gpu1_cluster_cascade_step0(scop):
    """the first prefilter call of the cluster workflow's cascade (`-s 1 --diag-score 0 --min-ungapped-score 0 --max-seqs 100 -c 0.8`, all-vs-all):
    k-mer match counts as scores.  The reference module with --gpu 1 takes the device path for it too (no silent CPU fallback) and writes
    what its own CPU path writes."""
    spec = MANIFEST["runs"]["pref_kmer"]
    par = _params(spec)
    for k, v in (("-s", "1"), ("--max-seqs", "100"), ("--diag-score", "0"), ("--min-ungapped-score", "0"), ("-c", "0.8"), ("--comp-bias-corr", "0"), ("--add-self-matches", "1")):
        par[par.index(k) + 1] = v
    cpu = [x for x in par if x != "--gpu"]
    cpu = cpu[:-1] if cpu[-1] == "1" and par[-2:] == ["--gpu", "1"] else cpu
    ref, mine = str(scop / "casc0_cpu"), str(scop / "casc0_gpu")
    _run([FS_CPU, "prefilter", str(scop / "db_ss"), str(scop / "db_ss"), ref] + cpu)
    log = _run([FS_GPU, "prefilter", str(scop / "db_ss"), str(scop / "db_ss"), mine] + par)
    assert "Index table (device)" in log, log[-1500:]
    a, b = read_db(ref), read_db(mine)
    assert a == b and sum(len(v) for v in a[1].values()) > 100


@pytest.mark.parametrize("extra", [["--prefilter-mode", "1"], ["--prefilter-mode", "1", "--alignment-type", "0"], ["--prefilter-mode", "0"],
                                   ["--prefilter-mode", "0", "-s", "7.5", "--max-seqs", "5"], ["--prefilter-mode", "1", "--sort-by-structure-bits", "0", "-e", "0.001"]])
def test_easy_search_from_a_structure_file_gpu1_equals_cpu_binary(padded_target, extra):
    """`foldseek easy-search <query structure file> <targetDB> out.m8 tmp`: the CPU reference binary vs `foldseek-fsgpu ... --gpu 1` (Marv
    + adapters on the MI355X), same prefilter mode on both sides (1: ungapped -- what --gpu 1 selects by default --, 0: k-mer).  Default
    parameters include --sort-by-structure-bits 1: the C-alpha DBs exist, so the adapter's records go through the reference's TMaligner /
    LDDTCalculator before they are ranked.  From createdb of the query to convertalis everything but the two device stages is the
    reference's own workflow."""
    w = padded_target
    q = os.path.join(EXAMPLES, "d1asha_")
    _run([FS_CPU, "easy-search", q, "target_pad", "cpu.m8", "tmp_cpu", "--threads", "2", "-v", "1"] + extra, cwd=w)
    log = _run([FS_GPU, "easy-search", q, "target_pad", "gpu.m8", "tmp_gpu", "--threads", "2", "-v", "3", "--gpu", "1"] + extra, cwd=w)
    assert "structurealign" in log and "--gpu 1" in log
    if extra[1] == "0":
        assert "Index table (device)" in log
    a, b = open(os.path.join(w, "cpu.m8"), "rb").read(), open(os.path.join(w, "gpu.m8"), "rb").read()
    assert a == b and a.count(b"\n") >= (2 if "--max-seqs" in extra else 4), (a[:600], b[:600])


def test_easy_search_all_against_all_gpu1_equals_cpu_binary(padded_target):
    """every example structure as query (12 x 12), alignments with backtraces in the output columns; one host thread on both sides: with
    more, the order of the per-query blocks in the result DB (hence in the .m8) follows the OpenMP schedule in the reference as well"""
    w = padded_target
    cols = ["--prefilter-mode", "1", "--format-output", "query,target,fident,alnlen,mismatch,gapopen,qstart,qend,tstart,tend,evalue,bits,cigar,qaln,taln"]
    _run([FS_CPU, "easy-search", EXAMPLES, "target_pad", "cpu.m8", "tmp_cpu", "--threads", "1", "-v", "1"] + cols, cwd=w)
    _run([FS_GPU, "easy-search", EXAMPLES, "target_pad", "gpu.m8", "tmp_gpu", "--threads", "1", "-v", "1", "--gpu", "1"] + cols, cwd=w)
    a, b = open(os.path.join(w, "cpu.m8"), "rb").read(), open(os.path.join(w, "gpu.m8"), "rb").read()
    assert a == b and a.count(b"\n") >= 60


def test_easy_cluster_gpu1_equals_cpu_binary(tmp_path):
    """`foldseek easy-cluster <structure files> out tmp` (BASELINE configs[4]'s workflow): linclust's kmermatcher + structurerescorediagonal + clust,
    then the cascade of three prefilter / structurealign / clust steps.  With `--gpu 1` the patched reference runs all three prefilter calls
    (the first with --diag-score 0) and its structurealign calls on this library; the cluster assignment must equal the CPU binary's."""
    w = str(tmp_path)
    _run([FS_CPU, "easy-cluster", EXAMPLES, "cpu", "tmp_cpu", "--threads", "1", "-v", "1"], cwd=w)
    log = _run([FS_GPU, "easy-cluster", EXAMPLES, "gpu", "tmp_gpu", "--threads", "1", "-v", "3", "--gpu", "1"], cwd=w)
    assert log.count("Index table (device)") == 3, log[-2000:]
    assert "Diagonal rescoring (device)" in log                       # linclust's kmermatcher hits rescored on the device
    assert sum(1 for l in log.splitlines() if l.startswith("structurealign ") and "--gpu 1" in l) >= 2
    a, b = open(os.path.join(w, "cpu_cluster.tsv"), "rb").read(), open(os.path.join(w, "gpu_cluster.tsv"), "rb").read()
    assert a == b and a.count(b"\n") == 12
    for f in ("_rep_seq.fasta", "_all_seqs.fasta"):
        assert open(os.path.join(w, "cpu" + f), "rb").read() == open(os.path.join(w, "gpu" + f), "rb").read()
