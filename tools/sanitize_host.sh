#!/bin/bash
# Host-side C++ (foldseek_amd/csrc/host/*.cpp, the module binary) under AddressSanitizer + UBSan, no GPU needed:
# builds a sanitized libfsgpu.so / fsgpu-modules into .san/ (git- and gpurun-ignored; the device objects are linked as they are),
# puts them in place of the product build for the duration of the given command and restores the product build afterwards.
#   tools/sanitize_host.sh python -m pytest tests -q -m "not gpu"
# Reports go to .san/asan.log.* / .san/ubsan.log.* (none = clean).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/foldseek_amd/csrc"
[ -f fsgpu.o ] || { echo "build the product first: python -c 'import __graft_entry__ as g; g.build()'" >&2; exit 1; }
mkdir -p "$ROOT/.san/host" "$ROOT/.san/bin"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
for f in host/*.cpp; do
    b=$(basename "$f" .cpp); [ "$b" = marv_shim ] && continue
    g++ -O1 -g -std=c++17 -fPIC -Wall -mavx2 -mfma $SAN -I../../include -I../data -c "$f" -o "$ROOT/.san/host/$b.o" &
done
wait
g++ -shared -fPIC $SAN -o "$ROOT/.san/libfsgpu.so" fsgpu.o fsgpu_kmer.o fsgpu_diag.o fsgpu_btrace.o fsgpu_sw3_na.o fsgpu_sw3_aa.o "$ROOT"/.san/host/*.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
g++ -O1 -g -std=c++17 $SAN -I../../include -o "$ROOT/.san/bin/fsgpu-modules" host/main_modules.cc -L"$ROOT/.san" -lfsgpu -Wl,-rpath,'$ORIGIN/..' -lpthread
cd "$ROOT"
cp foldseek_amd/libfsgpu.so .san/libfsgpu.so.orig
cp foldseek_amd/bin/fsgpu-modules .san/fsgpu-modules.orig
restore() { cp .san/libfsgpu.so.orig foldseek_amd/libfsgpu.so; cp .san/fsgpu-modules.orig foldseek_amd/bin/fsgpu-modules; }
trap restore EXIT
cp .san/libfsgpu.so foldseek_amd/libfsgpu.so
cp .san/bin/fsgpu-modules foldseek_amd/bin/fsgpu-modules
rm -f .san/asan.log* .san/ubsan.log*
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$ROOT/.san/asan.log
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$ROOT/.san/ubsan.log
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)     # python loads the library through ctypes
set +e
"$@"
rc=$?
ls .san/asan.log* .san/ubsan.log* 2>/dev/null && { echo "sanitizer reports above" >&2; rc=1; }
exit $rc
