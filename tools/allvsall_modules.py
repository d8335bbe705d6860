"""configs[4] through the NATIVE module: all-vs-all of an N-structure synthetic DB on disk with `fsgpu-modules search` (k-mer
prefilter -s 4.5 --max-seqs 200 -c 0.8 + structurealign -e 0.01 -c 0.8, the parameters of easy-cluster's cascaded step), process
start, DB load and index build included.  With families > 0 the DB is `families` seed structures + N/families - 1 mutated relatives of each (synth._homologs), the shape
a clustering input has; with 0 it is N unrelated structures (only the self match survives -e 0.01).
usage: allvsall_modules.py [N=200000] [threads=4] [families=0]
env C5_WRAP="rocprofv3 --kernel-trace --stats -d /tmp/prof -o c5 --": command prefix for a profile of the module run."""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n=200000, threads="4", fam=0, db=None, keep=False):
    """writes the DB (generated here unless given), runs the module, returns the summary dict"""
    from foldseek_amd import dbio, synth
    w = tempfile.mkdtemp(prefix="fs_c5_")
    try:
        t0 = time.time()
        if db is None:
            seeds = None
            if fam > 0:
                sd = synth.make_db_fast(fam, None, seed=7, homologs_per_query=0, mask_frac=0, x_frac=0)
                seeds = ([sd.data3di[sd.offsets[i]:sd.offsets[i] + sd.lengths[i]].copy() for i in range(fam)], [sd.dataaa[sd.offsets[i]:sd.offsets[i] + sd.lengths[i]].copy() for i in range(fam)])
            db = synth.make_db_fast(n, seeds, seed=20260923, homologs_per_query=max(n // fam - 1, 1) if fam > 0 else 0)
        n = db.n
        dbio.write_seq_db_from_padded(os.path.join(w, "db_ss"), db, "3di")
        dbio.write_seq_db_from_padded(os.path.join(w, "db"), db, "aa")
        t_gen = time.time() - t0
        env = dict(os.environ, FSGPU_MODULE_TIMING="1")
        env.pop("FSGPU_SPIN_US", None)          # bench.py sets a polling wait for ITS three feeder threads; the module's feeders must sleep while they wait
        t0 = time.time()
        p = subprocess.run(os.environ.get("C5_WRAP", "").split() + [os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules"), "search", os.path.join(w, "db"), os.path.join(w, "db"), os.path.join(w, "aln"),
                            "--prefilter-mode", "0", "-s", "4.5", "--max-seqs", "200", "-c", "0.8", "--cov-mode", "0", "-e", "0.01", "--alignment-type", "2",
                            "--comp-bias-corr", "0", "--sort-by-structure-bits", "0", "--add-self-matches", "1", "--threads", str(threads)],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = time.time() - t0
        if p.returncode != 0:
            raise RuntimeError("fsgpu-modules search failed: " + p.stdout[-2000:])
        lines = sum(1 for _ in open(os.path.join(w, "aln")))
        timing = [l.strip() for l in p.stdout.splitlines() if "timing" in l]
        return {"families": fam, "workload": f"all-vs-all {n} structures, fsgpu-modules search (k-mer prefilter -s 4.5 --max-seqs 200 -c 0.8 + structurealign -e 0.01 -c 0.8, 3Di+AA), "
                                             f"{threads} host threads, end to end incl. process start / DB load from disk / index build / result write",
                "seconds": dt, "queries_per_s": n / dt, "residues_per_s": n * float(db.residues) / dt, "alignment_lines": lines, "db_write_s": t_gen, "host_threads": int(threads),
                "module_timing": timing}
    finally:
        if not keep:
            shutil.rmtree(w, ignore_errors=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    threads = sys.argv[2] if len(sys.argv) > 2 else "4"
    fam = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    print(json.dumps(run(n, threads, fam)))
