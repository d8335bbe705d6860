"""configs[4] at its stated size against the reference BINARY: the cluster workflow's cascade (F/data/structurecluster.sh: three k-mer
prefilter calls -- -s 1 with k-mer match counts as scores, -s 4.5, -s 8, all -c 0.8, no composition bias, self matches -- each followed by
structurealign -e 0.01 -c 0.8) on a 200 000-structure database with the shape of a clustering input (20 000 families of 10: a seed
structure and mutated relatives).  1024 of its entries, spread over the whole length range (30 .. 2000 residues: paired and row-tiled
scans do not occur here, but every SW shape does: 32 lanes per target pair, 64 lanes for queries of 513 .. 1024 residues, the row-tiled
kernel beyond), are the queries of BOTH binaries -- the reference's CPU modules with 16 threads and `fsgpu-modules`:
  * prefilter: every entry byte-identical (ids, scores / match counts, diagonals, order, the --max-seqs cut);
  * structurealign on the reference's prefilter output: every record byte-identical (scores, e-values, coverage, start / end, CIGAR);
  * the fused `search` module with ONE feeder thread, i.e. device batches of up to 1024 queries as the all-vs-all run uses them (k-mer
    batches of 1024 queries with the 32-bit (query << 18 | target) candidate keys, one SW submission per batch over its few thousand
    pairs): prefilter DB and alignment DB byte-identical to the reference's two steps.
(What "byte-identical" covers for the backtrace-derived columns: see tests/test_modules_vs_reference_binary.py.)"""
import os
import sys

import numpy as np
import pytest

from foldseek_amd import dbio
from test_modules_vs_reference_binary import BIN, FS, MANIFEST, _run, _same

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(FS), reason="oracle/_ref_full/bin/foldseek not built")]

N, FAMILIES, NQ = 200000, 20000, 1024
CASCADE = [("-s", "1", "--max-seqs", "100", "--diag-score", "0", "--min-ungapped-score", "0"),
           ("-s", "4.5", "--max-seqs", "200", "--diag-score", "1", "--min-ungapped-score", "30"),
           ("-s", "8", "--max-seqs", "1000", "--diag-score", "1", "--min-ungapped-score", "30")]
PREF = ["--sub-mat", "aa:3di.out,nucl:3di.out", "--seed-sub-mat", "aa:3di.out,nucl:3di.out", "-k", "0", "--target-search-mode", "0",
        "--k-score", "seq:2147483647,prof:2147483647", "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535", "--split", "0", "--split-mode", "2",
        "--split-memory-limit", "0", "-c", "0.8", "--cov-mode", "0", "--comp-bias-corr", "0", "--comp-bias-corr-scale", "1", "--exact-kmer-matching", "0",
        "--mask", "0", "--mask-prob", "0.999995", "--mask-lower-case", "1", "--mask-n-repeat", "6", "--add-self-matches", "1", "--spaced-kmer-mode", "1",
        "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800", "--threads", "16", "--compressed", "0",
        "-v", "1"]


def _aln_par():
    par = list(MANIFEST["runs"]["aln_t2_a_e001_c08"]["parameters"])
    for k, v in (("--threads", "16"), ("--comp-bias-corr", "0"), ("--add-self-matches", "1"), ("-e", "0.01")):
        par[par.index(k) + 1] = v
    assert float(par[par.index("-c") + 1]) == 0.8
    return par


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    sys.path.insert(0, ROOT)
    import bench
    from foldseek_amd import synth
    w = tmp_path_factory.mktemp("c5")
    db = bench.allvsall_db(synth, N, FAMILIES)
    dbio.write_seq_db_from_padded(str(w / "t_ss"), db, "3di")
    dbio.write_seq_db_from_padded(str(w / "t"), db, "aa")
    pick = np.unique(np.linspace(0, db.n - 1, NQ).astype(np.int64))          # the DB is length sorted: the whole range, 30 .. 2000 residues
    sub = synth.PaddedDB(db.data3di, db.dataaa, np.append(db.offsets[:-1][pick], db.offsets[-1]), db.lengths[pick])
    # (offsets of a sub-selection are not contiguous: the writer only reads offsets[:-1] and lengths)
    dbio.write_seq_db_from_padded(str(w / "q_ss"), sub, "3di", keys=pick)
    dbio.write_seq_db_from_padded(str(w / "q"), sub, "aa", keys=pick)
    lens = db.lengths[pick]
    assert (lens <= 512).sum() > 500 and ((lens > 512) & (lens <= 1024)).sum() > 50 and (lens > 1024).sum() > 5
    return w


@pytest.mark.parametrize("step", [0, 1, 2])
def test_cascade_step_equals_the_reference_binary_at_200k(world, step):
    w = world
    par = PREF + list(CASCADE[step])
    _run([FS, "prefilter", "q_ss", "t_ss", f"ref_p{step}"] + par, w)
    _run([BIN, "prefilter", "q_ss", "t_ss", f"mine_p{step}"] + par, w)
    lines = _same(w, f"ref_p{step}", f"mine_p{step}")
    assert lines >= NQ + (NQ if step else 0), lines                           # the self match; from -s 4.5 on family members too
    apar = _aln_par()
    _run([FS, "structurealign", "q", "t", f"ref_p{step}", f"ref_a{step}"] + apar, w)
    # the accepted hits' backtraces on the device aligner in steps 0 and 2 (FSGPU_DEVICE_BACKTRACE=1: the reference binary's CIGARs are met by
    # k_block_backtrace itself), on the host restatement in step 1
    dev = "0" if step == 1 else "1"
    out = _run([BIN, "structurealign", "q", "t", f"ref_p{step}", f"mine_a{step}"] + apar, w, env={"FSGPU_DEVICE_BACKTRACE": dev, "FSGPU_BT_PASS2": "1", "FSGPU_MODULE_TIMING": "1"})      # PASS2: the 512-row pass takes whatever the first hands back
    alines = _same(w, f"ref_a{step}", f"mine_a{step}")
    assert alines >= NQ, alines
    import re
    m = re.search(r"backtrace [0-9.]+ \((\d+) of (\d+) on the device\)", out)
    assert m and int(m.group(2)) >= NQ and (int(m.group(1)) >= 0.95 * int(m.group(2)) if dev == "1" else int(m.group(1)) == 0), out[-300:]
    # the fused module, one feeder thread: device batches of up to 1024 queries (what the all-vs-all run submits; round 5: the limit of a device batch)
    s, maxseqs = CASCADE[step][1], CASCADE[step][3]
    fused = [BIN, "search", "q", "t", f"fused_a{step}", f"fused_p{step}", "--prefilter-mode", "0", "-s", s, "--max-seqs", maxseqs, "--diag-score", CASCADE[step][5],
             "--min-ungapped-score", CASCADE[step][7], "-c", "0.8", "--cov-mode", "0", "-e", "0.01", "--alignment-type", "2", "-a", "1", "--comp-bias-corr", "0",
             "--sort-by-structure-bits", "0", "--add-self-matches", "1", "--threads", "2"]
    env = dict(os.environ, FSGPU_FEEDERS="1", FSGPU_KMER_TRACE="1")
    import subprocess
    r = subprocess.run(fused, cwd=w, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    sizes = [int(l.split("nq=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("kmer batch nq=")]
    assert sizes and max(sizes) >= (900 if step == 0 else 256), sizes         # the batches the all-vs-all module runs with: at -s 1 a first batch of 32 (no history on the context), then the other 992 at once
    _same(w, f"ref_p{step}", f"fused_p{step}")
    _same(w, f"ref_a{step}", f"fused_a{step}")
