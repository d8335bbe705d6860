#!/bin/bash
# allvsall_ab2.sh N FAMILIES "THREADS VAR=val ..." ... : like allvsall_ab.sh with the thread count part of each setting
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; F=$2; shift 2
for setting in "$@"; do
  T=${setting%% *}; E=${setting#* }
  env $E timeout 300 python $R/tools/allvsall_modules.py $N $T $F > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting: FAILED"; tail -3 /tmp/ab.err; continue; }
  python - "$setting" <<'P'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "| %.2f s, %d lines |" % (d["seconds"], d["alignment_lines"]), d["module_timing"][0] if d["module_timing"] else "")
P
done
