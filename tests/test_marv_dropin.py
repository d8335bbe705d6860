"""The boundary that LINKS: the reference's own binary, built from its unmodified sources with -DHAVE_CUDA=1 and this
repository's `class Marv` (include/marv.h + foldseek_amd/csrc/host/marv_shim.cpp over libfsgpu.so) in place of libmarv
(oracle/build_ref_full.sh gpu -> oracle/_ref_full/bin/foldseek-fsgpu, a built file that travels to the GPU box).

Here the REFERENCE's runFilterOnGpu (M/src/prefiltering/ungappedprefilter.cpp:41-326), its gpuserver
(M/src/util/gpuserver.cpp:24-101) and its shared-memory client run on the MI355X through Marv::scan, on the
reference-written padded DB of tests/golden/scop_v1, and must reproduce the result DBs its own CPU path wrote."""
import json
import os
import shutil
import signal
import subprocess
import time

import pytest

from test_scop_golden import GOLD, MANIFEST, read_db, scop  # noqa: F401  (scop is a fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS_GPU = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek-fsgpu")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(FS_GPU), reason="oracle/_ref_full/bin/foldseek-fsgpu not built (oracle/build_ref_full.sh gpu)")]


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "foldseek_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    return env


def _gpu_par(server=0):
    par = list(MANIFEST["runs"]["pref_ung_pad"]["parameters"])
    par[par.index("--gpu") + 1] = "1"
    par[par.index("--gpu-server") + 1] = str(server)
    par[par.index("--prefilter-mode") + 1] = "0"           # what the search workflow passes with --gpu 1
    return par


def test_reference_ungappedprefilter_gpu_path_through_our_marv(scop):
    out = str(scop / "mine_gpu")
    r = subprocess.run([FS_GPU, "ungappedprefilter", str(scop / "db_ss"), str(scop / "db_pad_ss"), out] + _gpu_par(),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=_env())
    assert r.returncode == 0, r.stdout[-3000:]
    assert read_db(out) == read_db(str(scop / "pref_ung_pad"))


def test_marv_shards_targets_over_its_devices(scop):
    """libmarv drives all visible devices from one Marv object and shards the TARGETS (one query at a time by contract); so does
    the shim.  FSGPU_MARV_SHARDS=3 puts three shards on the one device of the test box: interleaved target subsets, concurrent
    scans, merged top lists == the single-shard result == the CPU path's"""
    env = _env()
    env["FSGPU_MARV_SHARDS"] = "3"
    out = str(scop / "mine_gpu3")
    r = subprocess.run([FS_GPU, "ungappedprefilter", str(scop / "db_ss"), str(scop / "db_pad_ss"), out] + _gpu_par(),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert read_db(out) == read_db(str(scop / "pref_ung_pad"))


def test_reference_gpuserver_and_client_through_our_marv(scop):
    env = _env()
    srv = subprocess.Popen([FS_GPU, "gpuserver", str(scop / "db_pad_ss"), "--max-seqs", "1000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    try:
        out = str(scop / "mine_srv")
        r = subprocess.run([FS_GPU, "ungappedprefilter", str(scop / "db_ss"), str(scop / "db_pad_ss"), out] +
                           _gpu_par(server=1), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout[-3000:]
        assert read_db(out) == read_db(str(scop / "pref_ung_pad"))
    finally:
        srv.send_signal(signal.SIGINT)
        try:
            srv.wait(timeout=30)
        except subprocess.TimeoutExpired:
            srv.kill()
    assert srv.returncode == 0


def test_reference_search_workflow_gpu(scop):
    """`foldseek search --gpu 1` on DBs (F/data/structuresearch.sh: ungappedprefilter --gpu 1 + structurealign), the reference's own
    workflow end to end with the prefilter on the MI355X == the result DB its CPU modules wrote (aln_t2_a_pad)"""
    res, tmp = str(scop / "mine_res"), str(scop / "tmp")
    r = subprocess.run([FS_GPU, "search", str(scop / "db"), str(scop / "db_pad"), res, tmp, "--gpu", "1", "-a", "1", "--sort-by-structure-bits", "0",
                        "--threads", "1", "-v", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "ungappedprefilter" in r.stdout and "--gpu 1" in r.stdout
    assert read_db(res) == read_db(str(scop / "aln_t2_a_pad"))
