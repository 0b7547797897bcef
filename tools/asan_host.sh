#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libradtxfr_hip.so (device code compiled as usual: the
# sanitizer flags go to the host compilation only, -Xarch_host), and the CPU test suite run against it: the ABI's argument
# checks and error paths, the host-side table builders (rtx_tud_gtable, the Chebyshev / Lagrange matrices, the hot-tile
# bound) and everything else that runs without a GPU. CPU container only (GPU sanitizer runs are not available on the pool).
#   bash tools/asan_host.sh            -> prints the pytest summary; non-zero exit on any sanitizer report
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/radtxfr_asan
mkdir -p "$OUT"
cd "$ROOT/radtxfr_amd/csrc"
pids=""
for f in rtx_lines rtx_voigt rtx_voigt_scatter rtx_sdvoigt rtx_tud rtx_radiance rtx_resample rtx_comm; do
  hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
        -ffp-contract=off -Wno-unused-function -c $f.hip -o "$OUT/$f.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -fsanitize=address,undefined -shared -o "$OUT/libradtxfr_asan.so" "$OUT"/*.o -ldl
RT=$(dirname "$(dirname "$(command -v hipcc)")")/lib/llvm/bin/clang
RT=$("$RT" -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  RADTXFR_LIB="$OUT/libradtxfr_asan.so" python -m pytest tests -x -q -m "not gpu"
