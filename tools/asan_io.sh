#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the native host library (csrc/epg_io.cpp with its own inflate,
# deflate, CRC-32 and the AVX-512 row parser: ~2 k lines that read untrusted files).  CPU build only -- nothing here touches
# a GPU.  Usage:  tools/asan_io.sh [--quick] [scratch-dir]
#   builds the library with -fsanitize=address,undefined -O1 -g into the scratch dir (default /tmp/epg_asan), points the
#   Python binding at it (EPILOGOS_IO_LIB), preloads libasan into the interpreter and runs
#     tests/test_native_io.py tests/test_roi.py     (parser, writers, codecs, STEP 4 helpers; incl. the file-level fuzz)
#     tools/fuzz_inflate.py --streams 100000          (--quick: 3000; the in-memory differential fuzz against zlib)
# Exit code 0 = no sanitizer report, no test failure, no fuzz mismatch.  A log of the last full run is kept in profiles/.
set -euo pipefail
QUICK=0
if [ "${1:-}" = "--quick" ]; then QUICK=1; shift; fi
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/epg_asan}"
mkdir -p "$OUT"
CXX="${CXX:-g++}"
ASAN_LIB="$($CXX -print-file-name=libasan.so)"
if [ ! -e "$ASAN_LIB" ] || [ "$ASAN_LIB" = "libasan.so" ]; then echo "asan_io: libasan.so not found for $CXX" >&2; exit 77; fi
"$CXX" -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -fPIC -shared -pthread \
    -I"$ROOT/include" "$ROOT/epilogos_amd/csrc/epg_io.cpp" -lz -o "$OUT/libepilogos_io_asan.so"
export EPILOGOS_IO_LIB="$OUT/libepilogos_io_asan.so"
export LD_PRELOAD="$ASAN_LIB"
# CPython itself leaks by design at exit; a report anywhere else aborts the run (exit code != 0)
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
cd "$ROOT"
if [ "$QUICK" = 1 ]; then
    python3 -m pytest tests/test_native_io.py -x -q -p no:cacheprovider -k "not differential_fuzz and not rolling_max and not sanitizer and not in_memory"
    python3 tools/fuzz_inflate.py --streams 3000
else
    python3 -m pytest tests/test_native_io.py tests/test_roi.py -x -q -p no:cacheprovider
    python3 tools/fuzz_inflate.py --streams 100000
fi
echo "asan_io: clean"
