#!/bin/bash
# all-vs-all through the native module under a list of environment settings: allvsall_ab.sh N THREADS FAMILIES "VAR=val ..." "VAR=val ..." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; T=$2; F=$3; shift 3
for setting in "$@"; do
  env $setting timeout 300 python $R/tools/allvsall_modules.py $N $T $F > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting: FAILED"; tail -3 /tmp/ab.err; continue; }
  python - "$setting" <<'P'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "| %.2f s, %d lines |" % (d["seconds"], d["alignment_lines"]), d["module_timing"][0] if d["module_timing"] else "")
P
done
