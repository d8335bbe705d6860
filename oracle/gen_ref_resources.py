#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: turns three reference data files into C byte arrays inside oracle/_ref/ (git-ignored),
the way the reference's own resource compiler does (M/cmake/MMseqsResourceCompiler.cmake), so the compiled
reference objects find `evalue_nn_kerasify[_len]` and the driver can build SubstitutionMatrix from text."""
import sys, os
ref, out = sys.argv[1], sys.argv[2]
def arr(name, path):
    d = open(path, "rb").read()
    body = ",".join(str(b) for b in d)
    return f"static const unsigned char {name}[] = {{{body}}};\nstatic const unsigned int {name}_len = {len(d)};\n"
with open(os.path.join(out, "ref_resources.h"), "w") as f:
    f.write("#pragma once\n")
    f.write(arr("ref_mat3di_out", f"{ref}/data/mat3di.out"))
    f.write(arr("ref_blosum62_out", f"{ref}/lib/mmseqs/data/blosum62.out"))
with open(os.path.join(out, "evalue_nn.kerasify.h"), "w") as f:
    f.write("#pragma once\n")
    f.write(arr("evalue_nn_kerasify", f"{ref}/data/evalue_nn.kerasify"))
