"""bench.py's contract is ONE JSON line on stdout.  Libraries write there too (librccl prints its version banner to the C stdout when the first communicator
comes up -- round 5 made the one-rank RCCL communicator the default at N = 1 and found a six-line stdout): the run points file descriptor 1 at stderr and
writes its line to a private duplicate of the real stdout.  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_only_the_result_line_reaches_stdout():
    code = ("import os, sys, bench\n"
            "bench._claim_stdout()\n"
            "os.write(1, b'RCCL version : banner from a C library\\n')\n"
            "print('a python print')\n"
            "bench.emit_line({'metric': 'x', 'value': 1.5, 'big': 'y' * 200000})\n"
            "os.write(1, b'printed at exit by a library\\n')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{"metric": "x", "value": 1.5') and len(lines[0]) > 200000
    assert "banner from a C library" in r.stderr and "a python print" in r.stderr and "printed at exit" in r.stderr


def test_without_the_claim_the_line_still_goes_to_stdout():
    r = subprocess.run([sys.executable, "-c", "import bench; bench.emit_line({'a': 1})"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout == '{"a": 1}\n'
