#!/usr/bin/env python3
"""oracle/patch_ref_full.py <copy-of-reference> <this-repo>  -- TEST INFRASTRUCTURE ONLY.

Edits the out-of-tree COPY of the reference (never /root/reference, never anything in this repository's history) so that
its own CMake build works without cargo and, optionally, links this repository's Marv class instead of libmarv:

  1. lib/mmseqs/CMakeLists.txt: the corrosion / Rust block (`set(ENV{CARGO_NET_OFFLINE} ...` up to the EMSCRIPTEN
     endif) becomes a static library `block_aligner_c` built from foldseek_amd/csrc/host/block_aligner.cpp +
     oracle/ref_block_stub.cpp (same C ABI, lib/block-aligner/c/block_aligner.h).
  2. lib/mmseqs/CMakeLists.txt: `if (ENABLE_CUDA)` libmarv block: with -DFSGPU_MARV=1 the target `marv` is
     foldseek_amd/csrc/host/marv_shim.cpp (class Marv of include/marv.h over the fsgpu_* C ABI) + libfsgpu.so.
  3. lib/mmseqs/data/resources/K4000.crf: created empty (stripped blob).
  4. the INTEGRATION.md 2 / 2b adapters, active only in the FSGPU_MARV build (-DHAVE_FSGPU=1; the CPU build of the same copy is
     byte for byte the reference's code path): `structurealign --gpu 1` and `prefilter --gpu 1` call into libfsgpu.so
       src/strucclustutils/structurealign.cpp   + #include "structurealign_fsgpu.inc", + the --gpu 1 branch in front of the OpenMP region
       src/commons/LocalParameters.cpp          + structurealign accepts --gpu
       lib/mmseqs/src/prefiltering/Prefiltering.{h,cpp}  + member `fsgpu`, no CPU index table when set, runSplit -> runSplitFsgpu,
                                                           + #include "prefiltering_fsgpu.inc"
       lib/mmseqs/src/commons/Parameters.cpp    + prefilter accepts --gpu
     The two .inc files are copied from foldseek_amd/csrc/host/adapters/.  The applied hunks are written to
     oracle/_ref_full/adapter_hunks.diff (INTEGRATION.md quotes them).
"""
import os
import re
import sys

src, repo = sys.argv[1], sys.argv[2]
cm = os.path.join(src, "lib", "mmseqs", "CMakeLists.txt")
text = open(cm).read()

a = text.index("set(ENV{CARGO_NET_OFFLINE} true)")
b = text.index("include_directories(lib/block-aligner/c)", a)
b_end = text.index("endif()", b) + len("endif()")
block = """# --- oracle/patch_ref_full.py: block-aligner C ABI from the C++ restatement (no cargo in this image) ---
add_library(block_aligner_c STATIC
    ${FSGPU_REPO}/foldseek_amd/csrc/host/block_aligner.cpp
    ${FSGPU_REPO}/oracle/ref_block_stub.cpp)
target_include_directories(block_aligner_c PRIVATE ${CMAKE_CURRENT_SOURCE_DIR}/lib/block-aligner/c)
target_compile_definitions(block_aligner_c PRIVATE FS_HAVE_BLOCK_ALIGNER=1)
set_target_properties(block_aligner_c PROPERTIES COMPILE_FLAGS "${MMSEQS_CXX_FLAGS} -O3 -w")
include_directories(lib/block-aligner/c)
"""
text = text[:a] + block + text[b_end:]

m = re.search(r"if \(ENABLE_CUDA\)\n\s*set\(LIBRARY_ONLY 1.*?\nendif \(\)", text, re.S)
assert m, "libmarv block not found"
marv = """if (ENABLE_CUDA)
    if (FSGPU_MARV)
        # --- oracle/patch_ref_full.py: class Marv = include/marv.h + marv_shim.cpp over libfsgpu.so ---
        include_directories(BEFORE ${FSGPU_REPO}/include)
        add_library(marv STATIC ${FSGPU_REPO}/foldseek_amd/csrc/host/marv_shim.cpp)
        target_include_directories(marv PRIVATE ${FSGPU_REPO}/include)
        set_target_properties(marv PROPERTIES COMPILE_FLAGS "${MMSEQS_CXX_FLAGS} -O2" POSITION_INDEPENDENT_CODE ON)
        target_link_libraries(marv ${FSGPU_REPO}/foldseek_amd/libfsgpu.so)
    else ()
        set(LIBRARY_ONLY 1 CACHE INTERNAL "" FORCE)
        include_directories(lib/libmarv/src)
        add_subdirectory(lib/libmarv/src EXCLUDE_FROM_ALL)
        set_target_properties(marv PROPERTIES POSITION_INDEPENDENT_CODE ON)
    endif ()
endif ()"""
text = text[:m.start()] + marv + text[m.end():]
open(cm, "w").write(text)

# ---- 4. adapters ------------------------------------------------------------------------------------------------------------
import difflib
import shutil

hunks = []


def edit(rel, pairs):
    path = os.path.join(src, rel)
    old = open(path).read()
    new = old
    for anchor, repl in pairs:
        assert new.count(anchor) == 1, (rel, anchor[:60], new.count(anchor))
        new = new.replace(anchor, repl)
    open(path, "w").write(new)
    hunks.extend(difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=2))


for name, dst in (("structurealign_fsgpu.inc", "src/strucclustutils"), ("structurerescorediagonal_fsgpu.inc", "src/strucclustutils"),
                  ("prefiltering_fsgpu.inc", "lib/mmseqs/src/prefiltering")):
    shutil.copy(os.path.join(repo, "foldseek_amd", "csrc", "host", "adapters", name), os.path.join(src, dst, name))

edit("src/strucclustutils/structurealign.cpp", [
    ("int structurealign(int argc, const char **argv, const Command& command) {",
     "#ifdef HAVE_FSGPU\n#include \"structurealign_fsgpu.inc\"\n#endif\n\nint structurealign(int argc, const char **argv, const Command& command) {"),
    ("#pragma omp parallel\n    {\n        unsigned int thread_idx = 0;\n#ifdef OPENMP\n        thread_idx = static_cast<unsigned int>(omp_get_thread_num());\n#endif\n        EvalueNeuralNet evaluer(tAADbr",
     "    bool fsgpuDone = false;\n#ifdef HAVE_FSGPU\n    if (par.gpu == 1) {\n        fsgpuDone = fsgpuStructureAlign(par, tAADbr, t3DiDbr, qAADbr, q3DiDbr, qcadbr, tcadbr, resultReader, dbw, subMat3Di, subMatAA,\n                                        sameDB, needCalpha, needTMaligner, needLDDT);\n    }\n#endif\n    if (fsgpuDone == false)\n"
     "#pragma omp parallel\n    {\n        unsigned int thread_idx = 0;\n#ifdef OPENMP\n        thread_idx = static_cast<unsigned int>(omp_get_thread_num());\n#endif\n        EvalueNeuralNet evaluer(tAADbr"),
])
edit("src/strucclustutils/structurerescorediagonal.cpp", [
    ("int structureungappedalign(int argc, const char **argv, const Command& command) {",
     "#ifdef HAVE_FSGPU\n#include \"structurerescorediagonal_fsgpu.inc\"\n#endif\n\nint structureungappedalign(int argc, const char **argv, const Command& command) {"),
    ("#pragma omp parallel\n    {\n        unsigned int thread_idx = 0;\n#ifdef OPENMP\n        thread_idx = static_cast<unsigned int>(omp_get_thread_num());\n#endif\n        EvalueNeuralNet evaluer(tAADbr",
     "    bool fsgpuDone = false;\n#ifdef HAVE_FSGPU\n    if (par.gpu == 1) {\n        fsgpuDone = fsgpuRescoreDiagonal(par, *tAADbr, *t3DiDbr, qdbrAA, qdbr3Di, resultReader, dbw, subMat3Di, subMatAA, sameDB, needTMaligner || needLDDT);\n    }\n#endif\n    if (fsgpuDone == false)\n"
     "#pragma omp parallel\n    {\n        unsigned int thread_idx = 0;\n#ifdef OPENMP\n        thread_idx = static_cast<unsigned int>(omp_get_thread_num());\n#endif\n        EvalueNeuralNet evaluer(tAADbr"),
])
edit("src/commons/LocalParameters.cpp", [
    ("    structurealign = combineList(structurealign, align);\n",
     "    structurealign = combineList(structurealign, align);\n#ifdef HAVE_FSGPU\n    structurealign.push_back(&PARAM_GPU);\n#endif\n"),
    ("    structurerescorediagonal = combineList(structurerescorediagonal, align);\n",
     "    structurerescorediagonal = combineList(structurerescorediagonal, align);\n#ifdef HAVE_FSGPU\n    structurerescorediagonal.push_back(&PARAM_GPU);\n#endif\n"),
])
edit("lib/mmseqs/src/commons/Parameters.cpp", [
    ("    prefilter.push_back(&PARAM_V);\n", "    prefilter.push_back(&PARAM_V);\n#ifdef HAVE_FSGPU\n    prefilter.push_back(&PARAM_GPU);\n#endif\n"),
])
edit("lib/mmseqs/src/prefiltering/Prefiltering.h", [
    ("    bool runSplit(const std::string &resultDB, const std::string &resultDBIndex, size_t split, bool merge);\n",
     "    bool runSplit(const std::string &resultDB, const std::string &resultDBIndex, size_t split, bool merge);\n"
     "    // device path (prefiltering_fsgpu.inc): set by the constructor when --gpu 1 and the parameters are ones it reproduces\n"
     "    bool fsgpu;\n    bool fsgpuUsable(const Parameters &par);\n    bool runSplitFsgpu(const std::string &resultDB, const std::string &resultDBIndex, bool merge);\n"),
])
edit("lib/mmseqs/src/prefiltering/Prefiltering.cpp", [
    ("    if (splitMode == Parameters::QUERY_DB_SPLIT) {\n        // create the whole index table\n        getIndexTable(0, 0, tdbr->getSize());\n",
     "    fsgpu = false;\n#ifdef HAVE_FSGPU\n    fsgpu = (splitMode == Parameters::QUERY_DB_SPLIT && splits == 1) ? fsgpuUsable(par) : false;\n#endif\n"
     "    if (fsgpu) {\n        // the index table is built on the device (runSplitFsgpu)\n        sequenceLookup = NULL;\n        indexTable = NULL;\n    } else if (splitMode == Parameters::QUERY_DB_SPLIT) {\n        // create the whole index table\n        getIndexTable(0, 0, tdbr->getSize());\n"),
    ("    Debug(Debug::INFO) << \"Process prefiltering step \" << (split + 1) << \" of \" << splits << \"\\n\\n\";\n",
     "    Debug(Debug::INFO) << \"Process prefiltering step \" << (split + 1) << \" of \" << splits << \"\\n\\n\";\n#ifdef HAVE_FSGPU\n    if (fsgpu) {\n        return runSplitFsgpu(resultDB, resultDBIndex, merge);\n    }\n#endif\n"),
])
with open(os.path.join(src, "lib/mmseqs/src/prefiltering/Prefiltering.cpp"), "a") as f:
    f.write("\n#ifdef HAVE_FSGPU\n#include \"prefiltering_fsgpu.inc\"\n#endif\n")
hunks.append("--- a/lib/mmseqs/src/prefiltering/Prefiltering.cpp (end of file)\n+++ b/lib/mmseqs/src/prefiltering/Prefiltering.cpp\n+#ifdef HAVE_FSGPU\n+#include \"prefiltering_fsgpu.inc\"\n+#endif\n")
out = os.path.join(os.path.dirname(os.path.abspath(src)), "adapter_hunks.diff")
open(out, "w").write("".join(hunks))
print("adapter hunks ->", out)

# HAVE_FSGPU + the C ABI headers for every target that builds on mmseqs-framework (foldseek's own sources included)
cm2 = os.path.join(src, "lib", "mmseqs", "src", "CMakeLists.txt")
t2 = open(cm2).read()
anchor = "    target_link_libraries(mmseqs-framework marv)\n"
assert t2.count(anchor) == 1
t2 = t2.replace(anchor, anchor + "    if (FSGPU_MARV)\n        target_compile_definitions(mmseqs-framework PUBLIC -DHAVE_FSGPU=1)\n"
                                 "        target_include_directories(mmseqs-framework PUBLIC ${FSGPU_REPO}/include)\n    endif ()\n")
open(cm2, "w").write(t2)

crf = os.path.join(src, "lib", "mmseqs", "data", "resources", "K4000.crf")
if not os.path.exists(crf):
    open(crf, "wb").close()
print("patched", cm)
