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

crf = os.path.join(src, "lib", "mmseqs", "data", "resources", "K4000.crf")
if not os.path.exists(crf):
    open(crf, "wb").close()
print("patched", cm)
