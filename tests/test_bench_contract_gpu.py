"""The bench line itself (-m gpu): `python bench.py` on a small configuration must print ONE JSON line with every key of the
measurement contract -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
dtype / data / config.workload, the `roofline` object of the dominant kernel and the `cpu_baseline` object -- and the numbers must
hang together (value = residues of the timed queries' scans / time, frac = achieved / peak, 50 planted homologs found)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--targets", "30000",
                        "--kmer-queries", "32", "--cpu-sample-targets", "30000", "--cpu-sample-queries", "4", "--kmer-cpu-queries", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("residues aligned/sec") and d["unit"] == "residues/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["data"], d["vs_baseline"]) == (1, 3, 1, "weak", "synthetic", None)
    c = d["config"]
    assert "workload" in c and c["targets"] == 30000 and "model" not in c
    # value = target residues seen by the timed queries / time
    per_step = c["queries_per_step"] * c["db_residues"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 45 <= d["alignments_per_query"] <= 50.5            # the planted homologs of every query are found and accepted
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0 < rf["valu"]["frac"] <= 1.0 and rf["kernel"] == "k_gapless"
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "residues/s"
    assert d["value"] > cb["value"]
