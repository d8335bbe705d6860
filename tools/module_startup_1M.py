"""Start-up cost of a module run against a 1M-structure target DB on disk (ASCII and padded layout), a handful of queries:
where the wall time of a single easy-search style invocation goes.  usage: module_startup_1M.py [N=1000000]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from foldseek_amd import dbio, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
w = tempfile.mkdtemp(prefix="fs_start_")
q3, qa = synth.make_queries(4, seed=3, lo=250, hi=450)
db = synth.make_db_fast(n, (q3, qa), seed=20260923, homologs_per_query=50)
dbio.write_seq_db(os.path.join(w, "q"), qa); dbio.write_seq_db(os.path.join(w, "q_ss"), q3)
dbio.write_seq_db_from_padded(os.path.join(w, "t_ss"), db, "3di"); dbio.write_seq_db_from_padded(os.path.join(w, "t"), db, "aa")
dbio.write_padded_db(os.path.join(w, "p_ss"), db, "3di"); dbio.write_padded_db(os.path.join(w, "p"), db, "aa")
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")
env = dict(os.environ, FSGPU_MODULE_TIMING="1")
for name, t in (("ASCII target", "t"), ("padded target", "p")):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([BIN, "search", os.path.join(w, "q"), os.path.join(w, t), os.path.join(w, f"aln_{t}_{rep}"), "--prefilter-mode", "1", "--alignment-type", "2",
                            "-a", "1", "--sort-by-structure-bits", "0", "--threads", "3"], env=env, capture_output=True, text=True)
        print(f"{name} run {rep}: rc {r.returncode} wall {time.time() - t0:.2f} s   {[l for l in r.stderr.splitlines() if 'timing' in l][-1:]}", flush=True)
