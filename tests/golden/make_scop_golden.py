#!/usr/bin/env python3
"""tests/golden/make_scop_golden.py -- generator of tests/golden/scop_v1 (TEST INFRASTRUCTURE).

Runs the WHOLE reference binary (oracle/_ref_full/bin/foldseek, built from a patched out-of-tree copy of
/root/reference by oracle/build_ref_full.sh) on the reference's own example structures (F/example: 25 SCOP domains +
1tim/8tim) and freezes

  * the databases the reference's `createdb` and `makepaddedseqdb` WRITE (db, db_ss, db_h, db_pad*, with .index /
    .dbtype / .lookup) -- the modules under test read these, never a DB written by this repository's own dbio.py;
  * the result DBs of the reference's `prefilter` (k-mer), `ungappedprefilter` (CPU) and `structurealign` run
    all-vs-all with the parameter strings F/data/structuresearch.sh passes (captured with `easy-search -v 3`), plus
    the variants listed in RUNS below.

Only runs where /root/reference and the built binary exist (this container); the frozen files travel to the GPU box.
`--sort-by-structure-bits 0`: TM-score/LDDT rescoring needs the _ca DB and is out of scope.
"""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FS = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
EXAMPLE = "/root/reference/example"
OUT = os.path.join(HERE, "scop_v1")

SUBMAT = "aa:3di.out,nucl:3di.out"
# F/data/structuresearch.sh:41-53 ${PREFILTER_PAR} as easy-search composes it (v 3 log of the reference binary)
PREFILTER_PAR = ["--sub-mat", SUBMAT, "--seed-sub-mat", SUBMAT, "-s", "9.5", "-k", "6", "--target-search-mode", "0",
                 "--k-score", "seq:2147483647,prof:2147483647", "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535",
                 "--max-seqs", "1000", "--split", "0", "--split-mode", "2", "--split-memory-limit", "0", "-c", "0",
                 "--cov-mode", "0", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "0.15", "--diag-score", "1",
                 "--exact-kmer-matching", "0", "--mask", "0", "--mask-prob", "0.999995", "--mask-lower-case", "1",
                 "--mask-n-repeat", "6", "--min-ungapped-score", "30", "--add-self-matches", "0", "--spaced-kmer-mode", "1",
                 "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800",
                 "--threads", "1", "--compressed", "0", "-v", "1"]
UNGAPPED_PAR = ["--sub-mat", SUBMAT, "-c", "0", "-e", "1.79769e+308", "--cov-mode", "0", "--comp-bias-corr", "1",
                "--comp-bias-corr-scale", "0.15", "--min-ungapped-score", "30", "--max-seqs", "1000", "--db-load-mode", "0",
                "--gpu", "0", "--gpu-server", "0", "--gpu-server-wait-timeout", "600", "--prefilter-mode", "1",
                "--threads", "1", "--compressed", "0", "-v", "1"]


def align_par(atype, a, extra=()):
    # F/data/structuresearch.sh:116-143 ${ALIGNMENT_PAR}; --sort-by-structure-bits 0 (no _ca rescoring, see docstring)
    return ["--tmscore-threshold", "0", "--tmscore-threshold-mode", "0", "--lddt-threshold", "0", "--sort-by-structure-bits", "0",
            "--alignment-type", str(atype), "--exact-tmscore", "0", "--sub-mat", SUBMAT, "-a", str(a), "--alignment-mode", "3",
            "--alignment-output-mode", "0", "--wrapped-scoring", "0", "-e", "10", "--min-seq-id", "0", "--min-aln-len", "0",
            "--seq-id-mode", "0", "--alt-ali", "0", "-c", "0", "--cov-mode", "0", "--max-seq-len", "65535", "--comp-bias-corr", "1",
            "--comp-bias-corr-scale", "0.5", "--max-rejected", "2147483647", "--max-accept", "2147483647", "--add-self-matches", "0",
            "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800",
            "--score-bias", "0", "--realign", "0", "--realign-score-bias", "-0.2", "--realign-max-seqs", "2147483647",
            "--corr-score-weight", "0", "--gap-open", "aa:10,nucl:10", "--gap-extend", "aa:1,nucl:1", "--zdrop", "40",
            "--threads", "1", "--compressed", "0", "-v", "1"] + list(extra)


def override(par, **kw):
    """replace the value of given flags in a parameter list (flag names with '-' written as '_' and a leading 'p_')"""
    par = list(par)
    for k, v in kw.items():
        flag = k
        i = par.index(flag)
        par[i + 1] = str(v)
    return par


# name -> (module, positional args (relative to the work dir), parameter list)
RUNS = {
    # ---- prefilters ----
    "pref_kmer": ("prefilter", ["db_ss", "db_ss"], PREFILTER_PAR),
    "pref_kmer_s75": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"-s": "7.5"})),
    "pref_kmer_max5_self": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"--max-seqs": "5", "--add-self-matches": "1"})),
    "pref_kmer_pad": ("prefilter", ["db_ss", "db_pad_ss"], PREFILTER_PAR),
    "pref_ung": ("ungappedprefilter", ["db_ss", "db_ss"], UNGAPPED_PAR),
    "pref_ung_pad": ("ungappedprefilter", ["db_ss", "db_pad_ss"], UNGAPPED_PAR),
    "pref_ung_max7": ("ungappedprefilter", ["db_ss", "db_ss"], override(UNGAPPED_PAR, **{"--max-seqs": "7", "--min-ungapped-score": "40"})),
    "pref_ung_nocb": ("ungappedprefilter", ["db_ss", "db_ss"], override(UNGAPPED_PAR, **{"--comp-bias-corr": "0"})),
    # ---- structurealign (3Di+AA and 3Di only), on the prefilter results above ----
    "aln_t2_a": ("structurealign", ["db", "db", "pref_kmer"], align_par(2, 1)),
    "aln_t0_a": ("structurealign", ["db", "db", "pref_kmer"], align_par(0, 1)),
    "aln_t2": ("structurealign", ["db", "db", "pref_kmer"], align_par(2, 0)),
    "aln_t2_a_ung": ("structurealign", ["db", "db", "pref_ung"], align_par(2, 1)),
    "aln_t2_a_pad": ("structurealign", ["db", "db_pad", "pref_ung_pad"], align_par(2, 1)),
    "aln_t0_a_pad": ("structurealign", ["db", "db_pad", "pref_kmer_pad"], align_par(0, 1)),
    "aln_t2_a_e001_c08": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-e": "0.001", "-c": "0.8"})),
    "aln_t2_a_cov2": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-c": "0.9", "--cov-mode": "2"})),
    "aln_t2_a_maxacc": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--max-accept": "3", "--max-rejected": "2"})),
    "aln_t2_a_altali": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--alt-ali": "2"})),
    "aln_t2_a_nocb": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--comp-bias-corr": "0"})),
    # ---- acceptance criteria other than the workflow's (Alignment::checkCriteria, Util::hasCoverage / canBeCovered, seq-id modes) ----
    "aln_t2_a_sid1": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--min-seq-id": "0.3", "--seq-id-mode": "1"})),
    "aln_t2_a_sid2": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--min-seq-id": "0.2", "--seq-id-mode": "2"})),
    "aln_t2_a_minlen": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--min-aln-len": "60"})),
    "aln_t2_a_cov1": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-c": "0.7", "--cov-mode": "1"})),
    "aln_t2_a_cov3": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-c": "0.6", "--cov-mode": "3"})),
    "aln_t2_a_cov4": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-c": "0.6", "--cov-mode": "4"})),
    "aln_t2_a_cov5": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"-c": "0.6", "--cov-mode": "5"})),
    "aln_t0_a_gap": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(0, 1), **{"--gap-open": "aa:8,nucl:8", "--gap-extend": "aa:2,nucl:2"})),
    "aln_t2_a_cbs": ("structurealign", ["db", "db", "pref_kmer"], override(align_par(2, 1), **{"--comp-bias-corr-scale": "0.25"})),
    "pref_kmer_c08": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"-c": "0.8", "--cov-mode": "0"})),
    "pref_kmer_c07m2": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"-c": "0.7", "--cov-mode": "2"})),
    "pref_kmer_c07m5": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"-c": "0.7", "--cov-mode": "5"})),
    "pref_kmer_nospace": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"--spaced-kmer-mode": "0"})),
    "pref_kmer_nomasklc": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"--mask-lower-case": "0"})),
    "pref_kmer_cbs0": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"--comp-bias-corr": "0"})),
    "pref_kmer_minung45": ("prefilter", ["db_ss", "db_ss"], override(PREFILTER_PAR, **{"--min-ungapped-score": "45"})),
}


def rescore_par(atype, a, e, c, extra=()):
    # F/data/structurecluster.sh via easy-cluster -v 3: ${STRUCTURERESCOREDIAGONAL_PAR}
    return ["--exact-tmscore", "0", "--tmscore-threshold", "0", "--tmscore-threshold-mode", "0", "--lddt-threshold", "0",
            "--alignment-type", str(atype), "--sub-mat", SUBMAT, "-a", str(a), "--alignment-mode", "3", "--alignment-output-mode", "0",
            "--wrapped-scoring", "0", "-e", str(e), "--min-seq-id", "0", "--min-aln-len", "0", "--seq-id-mode", "0", "--alt-ali", "0",
            "-c", str(c), "--cov-mode", "0", "--max-seq-len", "65535", "--comp-bias-corr", "0", "--comp-bias-corr-scale", "1",
            "--max-rejected", "2147483647", "--max-accept", "2147483647", "--add-self-matches", "1", "--db-load-mode", "0",
            "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800", "--score-bias", "0", "--realign", "0",
            "--realign-score-bias", "-0.2", "--realign-max-seqs", "2147483647", "--corr-score-weight", "0", "--gap-open", "aa:10,nucl:10",
            "--gap-extend", "aa:1,nucl:1", "--zdrop", "40", "--threads", "1", "--compressed", "0", "-v", "1"] + list(extra)


RESCORE_RUNS = {
    # structurerescorediagonal on the k-mer prefilter's (target, diagonal) lists restricted to the pairs whose reference result
    # is DEFINED (pref_kmer_defined, written by write_defined_pref below: diagonal >= 0, or < 0 with a target not longer than
    # the query -- see foldseek_amd/csrc/fsgpu_diag.hip); the first line is the parameter set of the cluster workflow
    "resc_t2_clu": ("structurerescorediagonal", ["db", "db", "pref_kmer_defined"], rescore_par(2, 0, "0.01", "0.8")),
    "resc_t2_a": ("structurerescorediagonal", ["db", "db", "pref_kmer_defined"], rescore_par(2, 1, "10", "0")),
    "resc_t0_a_sid1": ("structurerescorediagonal", ["db", "db", "pref_kmer_defined"], override(rescore_par(0, 1, "1000", "0.3"), **{"--seq-id-mode": "1", "--cov-mode": "2"})),
}


# precomputed indexes (F/data/structureindex.sh: `createindex` = two `indexdb` calls; the C-alpha append step is skipped, no _ca DB here).
# The .idx files (900 MB each: 20^6 offsets + 3-mer matrices) are NOT frozen: the tests recreate them with the reference binary.
INDEXDB_COMMON = ["--seed-sub-mat", SUBMAT, "-k", "0", "--alph-size", "aa:21,nucl:5", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "1",
                  "--max-seq-len", "65535", "--max-seqs", "1000", "--mask", "0", "--mask-prob", "0.999995", "--mask-lower-case", "1",
                  "--mask-n-repeat", "6", "--spaced-kmer-mode", "1", "-s", "9.5", "--k-score", "seq:2147483647,prof:2147483647",
                  "--check-compatible", "0", "--search-type", "0", "--split", "0", "--split-memory-limit", "0", "-v", "1", "--threads", "1"]
INDEXDB = [["indexdb", "db", "db"] + INDEXDB_COMMON + ["--index-subset", "2"],
           ["indexdb", "db_ss", "db_ss"] + INDEXDB_COMMON + ["--index-dbsuffix", "_ss", "--index-subset", "5"]]
INDEX_RUNS = {
    # what the search workflow runs once indexes exist: the TARGET is the index (structuresearch.sh "${TARGET_PREFILTER}${INDEXEXT}")
    "pref_kmer_idx": ("prefilter", ["db_ss", "db_ss.idx"], override(PREFILTER_PAR, **{"-k": "0"})),
    "aln_t2_a_idx": ("structurealign", ["db", "db.idx", "pref_kmer_idx"], align_par(2, 1)),
}


# convertalis (F/data/easystructuresearch.sh) -- first the complete parameter string `easy-search -v 3` prints for it
CONVERT_PAR = ["--sub-mat", SUBMAT, "--format-mode", "0", "--format-output", "query,target,fident,alnlen,mismatch,gapopen,qstart,qend,tstart,tend,evalue,bits",
               "--translation-table", "1", "--gap-open", "aa:10,nucl:10", "--gap-extend", "aa:1,nucl:1", "--db-output", "0", "--db-load-mode", "0",
               "--search-type", "0", "--threads", "1", "--compressed", "0", "-v", "1", "--exact-tmscore", "0"]
ALL_COLUMNS = ("query,target,qkey,tkey,evalue,gapopen,pident,fident,nident,qstart,qend,qlen,tstart,tend,tlen,alnlen,bits,cigar,qseq,tseq,q3di,t3di,"
               "qheader,theader,qaln,taln,q3dialn,t3dialn,mismatch,qcov,tcov,empty")
CONVERT_RUNS = {
    "conv_default.m8": ("convertalis", ["db", "db", "aln_t2_a"], CONVERT_PAR),
    "conv_nobt.m8": ("convertalis", ["db", "db", "aln_t2"], CONVERT_PAR),                      # no backtrace: estimated mismatches, gap opens 0
    "conv_t0.m8": ("convertalis", ["db", "db", "aln_t0_a"], ["--threads", "1", "-v", "1"]),      # module defaults
    "conv_fmt2.m8": ("convertalis", ["db", "db", "aln_t2_a_altali"], override(CONVERT_PAR, **{"--format-mode": "2"})),
    "conv_fmt4_all.m8": ("convertalis", ["db", "db", "aln_t2_a"], override(CONVERT_PAR, **{"--format-mode": "4", "--format-output": ALL_COLUMNS})),
    "conv_pad.m8": ("convertalis", ["db", "db_pad", "aln_t2_a_pad"], override(CONVERT_PAR, **{"--format-output": "query,target,tkey,theader,taln,t3dialn,bits"})),
    "conv_resc.m8": ("convertalis", ["db", "db", "resc_t2_a"], override(CONVERT_PAR, **{"--format-output": "query,target,fident,nident,alnlen,mismatch,gapopen,cigar,qcov,tcov,evalue,bits"})),
    "conv_dbout": ("convertalis", ["db", "db", "aln_t2_a"], override(CONVERT_PAR, **{"--db-output": "1"})),
    # set columns (<db>.lookup / <db>.source); `tset` on its own prints the EMPTY source name in the reference (it asks for the lookup only)
    "conv_sets.m8": ("convertalis", ["db", "db", "aln_t2_a"], override(CONVERT_PAR, **{"--format-output": "query,target,qset,qsetid,tset,tsetid,bits"})),
    "conv_tset_alone.m8": ("convertalis", ["db", "db_pad", "aln_t2_a_pad"], override(CONVERT_PAR, **{"--format-output": "query,target,tset,tkey"})),
}


def read_db(path):
    data = open(path, "rb").read()
    out = {}
    for line in open(path + ".index"):
        k, off, ln = line.split()
        out[int(k)] = data[int(off):int(off) + int(ln) - 1]
    return out


def write_defined_pref(work):
    """pref_kmer restricted to pairs with a defined reference result; an INPUT of the reference run, same on-disk format"""
    lens = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(os.path.join(work, "db_ss.index"))}
    pref = read_db(os.path.join(work, "pref_kmer"))
    blob, index, off, kept, dropped = b"", [], 0, 0, 0
    for q in sorted(pref):
        lines = []
        for ln in pref[q].decode().splitlines():
            t, _, d = ln.split("\t")
            d, lq, lt = int(d), lens[q], lens[int(t)]
            ok = (d >= 0 and d < lq) or (d < 0 and -d < lt and lt <= lq)
            kept += ok; dropped += not ok
            if ok:
                lines.append(ln)
        body = ("\n".join(lines) + "\n" if lines else "").encode() + b"\0"
        index.append(f"{q}\t{off}\t{len(body)}\n")
        blob += body; off += len(body)
    open(os.path.join(work, "pref_kmer_defined"), "wb").write(blob)
    open(os.path.join(work, "pref_kmer_defined.index"), "w").write("".join(index))
    shutil.copy(os.path.join(work, "pref_kmer.dbtype"), os.path.join(work, "pref_kmer_defined.dbtype"))
    return kept, dropped


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise SystemExit(f"FAILED: {' '.join(cmd)}")
    return r.stdout


def main():
    if not os.path.exists(FS):
        raise SystemExit("build the reference first: bash oracle/build_ref_full.sh cpu")
    work = os.path.join(ROOT, "oracle", "_ref_full", "scop_work")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    run([FS, "createdb", EXAMPLE, "db", "--threads", "1", "-v", "1"], work)
    for f in os.listdir(work):          # no C-alpha DB: TM/LDDT rescoring is out of scope and turns itself off without it
        if f.startswith("db_ca"):
            os.remove(os.path.join(work, f))
    run([FS, "makepaddedseqdb", "db", "db_pad", "--threads", "1", "-v", "1"], work)
    manifest = {"reference": "steineggerlab/foldseek tree at /root/reference, binary from oracle/build_ref_full.sh cpu",
                "links": {}, "runs": {}}
    for name, (module, pos, par) in RUNS.items():
        cmd = [FS, module] + pos + [name] + par
        run(cmd, work)
        manifest["runs"][name] = {"module": module, "positional": pos, "parameters": par}
    kept, dropped = write_defined_pref(work)
    manifest["pref_kmer_defined"] = {"pairs_kept": kept, "pairs_with_undefined_reference_result_dropped": dropped}
    for name, (module, pos, par) in RESCORE_RUNS.items():
        run([FS, module] + pos + [name] + par, work)
        manifest["runs"][name] = {"module": module, "positional": pos, "parameters": par}
    manifest["convert_runs"] = {}
    for name, (module, pos, par) in CONVERT_RUNS.items():
        run([FS, module] + pos + [name] + par, work)
        manifest["convert_runs"][name] = {"module": module, "positional": pos, "parameters": par}
    for ext in ("", ".index", ".dbtype"):       # structureindex.sh links the header DB next to the 3Di DB first (lndb)
        if not os.path.exists(os.path.join(work, "db_ss_h" + ext)):
            os.symlink(os.path.join(work, "db_h" + ext), os.path.join(work, "db_ss_h" + ext))
    for cmd in INDEXDB:
        run([FS] + cmd, work)
    manifest["indexdb"] = INDEXDB
    for name, (module, pos, par) in INDEX_RUNS.items():
        run([FS, module] + pos + [name] + par, work)
        manifest["runs_with_index"] = manifest.get("runs_with_index", {})
        manifest["runs_with_index"][name] = {"module": module, "positional": pos, "parameters": par}
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    for f in sorted(os.listdir(work)):
        p = os.path.join(work, f)
        if os.path.islink(p):            # makepaddedseqdb links the AA / header data files to the source DB's
            manifest["links"][f] = os.path.basename(os.readlink(p))
            continue
        if os.path.isdir(p) or "_tmp" in f or ".idx" in f:
            continue
        if f.endswith(".m8") and os.path.getsize(p) > 100000:      # large text outputs are frozen gzip-compressed (mtime 0: reproducible bytes)
            import gzip
            with open(os.path.join(OUT, f + ".gz"), "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as gz:
                gz.write(open(p, "rb").read())
            continue
        shutil.copy(p, os.path.join(OUT, f))
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    n = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"{len(os.listdir(OUT))} files, {n} bytes -> {OUT}")


if __name__ == "__main__":
    main()
