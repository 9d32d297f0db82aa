#!/bin/bash
# Builds the HOST side of formats.hip / beam.hip / ctx.hip with AddressSanitizer + UndefinedBehaviorSanitizer (hipcc --cuda-host-only:
# no device code, no GPU needed) and runs tests/cabi/asan_text.cpp against it.  Exit code: 0 = clean, 1 = a sanitizer report or a
# driver failure, 77 = the sanitizer build is not possible in this environment (the caller skips).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/fa_asan_text}"
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT" && cd "$OUT" || exit 77
FLAGS="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
for f in formats beam ctx; do
    /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 --cuda-host-only -ffp-contract=off -c "$ROOT/fluidaudio_amd/csrc/$f.hip" -o $f.o > build.log 2>&1 || { tail -5 build.log; exit 77; }
done
# the host objects reference their (absent) device images: stand-ins that are never loaded, since the driver launches no kernel
{ echo '/* generated */'; for s in $(nm -u formats.o beam.o ctx.o | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u); do
    echo "const char $s[4096] __attribute__((aligned(4096))) = \"__CLANG_OFFLOAD_BUNDLE__\";"; done; } > fat.c
$LLVM/clang -c fat.c -o fat.o >> build.log 2>&1 || exit 77
$LLVM/clang++ $FLAGS -I "$ROOT/include" -c "$ROOT/tests/cabi/asan_text.cpp" -o drv.o >> build.log 2>&1 || { tail -5 build.log; exit 77; }
$LLVM/clang++ -fsanitize=address,undefined drv.o formats.o beam.o ctx.o fat.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o drv >> build.log 2>&1 || { tail -5 build.log; exit 77; }
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 timeout 600 ./drv > run.log 2>&1
rc=$?
tail -5 run.log | cut -c1-400
[ $rc -eq 0 ] && grep -q '^done:' run.log && ! grep -q 'runtime error\|ERROR: AddressSanitizer\|LeakSanitizer' run.log
