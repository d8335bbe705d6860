#!/bin/bash
# oracle/build_ref_full.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds the WHOLE reference binary (`foldseek`) from a patched, out-of-tree copy of /root/reference so that the
# reference's own createdb / makepaddedseqdb / prefilter / ungappedprefilter / structurealign can write the DBs and
# result DBs the parity tests are frozen against (tests/golden/scop_v1, generator tests/golden/make_scop_golden.py).
#
#   oracle/_ref_full/src     patched copy of the reference tree   (git-ignored, gpurun-ignored: never shipped)
#   oracle/_ref_full/build   CPU build  (HAVE_AVX2=1, no CUDA)     (same)
#   oracle/_ref_full/build_gpu  the SAME sources with -DHAVE_CUDA=1, `marv` = include/marv.h + marv_shim.cpp over
#                            libfsgpu.so: the reference's own ungappedprefilter.cpp / gpuserver.cpp / GpuUtil.cpp
#                            compiled and linked against this repository's Marv class with zero source hunks
#   oracle/_ref_full/bin/foldseek        the reference, CPU path      (travels to the GPU box: CPU baseline + checker)
#   oracle/_ref_full/bin/foldseek-fsgpu  the reference + our Marv + the INTEGRATION.md 2 / 2b adapters (structurealign / prefilter
#                                        --gpu 1 call libfsgpu.so; hunks in oracle/_ref_full/adapter_hunks.diff)  (travels to the GPU box: drop-in tests)
#
# The stock build cannot run here: M/CMakeLists.txt:210-251 imports the Rust crate lib/block-aligner through
# corrosion and no cargo/rustc exists in this image.  The patch (oracle/patch_ref_full.py) replaces exactly that
# block by a static library `block_aligner_c` made of our C++ restatement (foldseek_amd/csrc/host/block_aligner.cpp)
# + the placeholder symbols of oracle/ref_block_stub.cpp, and creates an empty K4000.crf (a stripped large blob,
# F/.MISSING_LARGE_BLOBS; only used by `mmseqs` nucleotide modules).  Nothing else in the copy is touched.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
REF="${REF:-/root/reference}"
OUT="$HERE/_ref_full"
JOBS="${JOBS:-8}"
WHAT="${1:-all}"          # cpu | gpu | all

[ -d "$REF/src" ] || { echo "no reference tree at $REF" >&2; exit 2; }
mkdir -p "$OUT/bin"
if [ ! -f "$OUT/src/.patched_v3" ]; then
    rm -rf "$OUT/src"
    cp -a "$REF" "$OUT/src"
    python3 "$HERE/patch_ref_full.py" "$OUT/src" "$REPO"
    touch "$OUT/src/.patched_v3"
fi

COMMON=(-G Ninja -DCMAKE_BUILD_TYPE=Release -DHAVE_AVX2=1 -DENABLE_PROSTT5=0 -DENABLE_STRUCTTY=0
        -DFSGPU_REPO="$REPO" -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -Wno-dev)

if [ "$WHAT" = cpu ] || [ "$WHAT" = all ]; then
    cmake -S "$OUT/src" -B "$OUT/build" "${COMMON[@]}" -DENABLE_CUDA=0 > "$OUT/cmake_cpu.log" 2>&1 || { tail -30 "$OUT/cmake_cpu.log"; exit 1; }
    ninja -C "$OUT/build" -j "$JOBS" foldseek > "$OUT/ninja_cpu.log" 2>&1 || { grep -B2 -A12 "error\|FAILED" "$OUT/ninja_cpu.log" | head -80; exit 1; }
    cp "$OUT/build/src/foldseek" "$OUT/bin/foldseek"
    strip "$OUT/bin/foldseek"
fi
if [ "$WHAT" = gpu ] || [ "$WHAT" = all ]; then
    make -s -C "$REPO/foldseek_amd/csrc"
    cmake -S "$OUT/src" -B "$OUT/build_gpu" "${COMMON[@]}" -DENABLE_CUDA=1 -DENABLE_HIP=1 -DFSGPU_MARV=1 > "$OUT/cmake_gpu.log" 2>&1 || { tail -30 "$OUT/cmake_gpu.log"; exit 1; }
    ninja -C "$OUT/build_gpu" -j "$JOBS" foldseek > "$OUT/ninja_gpu.log" 2>&1 || { grep -B2 -A12 "error\|FAILED" "$OUT/ninja_gpu.log" | head -80; exit 1; }
    cp "$OUT/build_gpu/src/foldseek" "$OUT/bin/foldseek-fsgpu"
    strip "$OUT/bin/foldseek-fsgpu"
fi
ls -la "$OUT/bin"
