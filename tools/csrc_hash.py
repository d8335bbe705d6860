"""Hash of the device kernel sources (foldseek_amd/csrc/*.hip, *.hpp, *.h): ties a committed PMC pass (profiles/pmc_traffic*.json) to the
kernels it was collected on; bench.py prints `traffic: null` + a warning when the running sources differ."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(files=None):
    """files: names under foldseek_amd/csrc the measured kernel is compiled from (default: every device source)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "foldseek_amd", "csrc")
    paths = [os.path.join(d, f) for f in files] if files else glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.h"))
    for f in sorted(paths):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    import sys
    print(csrc_hash(sys.argv[1:] or None))
