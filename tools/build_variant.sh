#!/bin/bash
# A/B builds for timing experiments: tools/build_variant.sh NAME "-DRTX_SC_ABLATE=3 ..."  -> build/NAME.so
# run with RADTXFR_LIB=build/NAME.so (build/ is git-ignored but travels to the GPU box)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/$NAME"
cd "$ROOT/radtxfr_amd/csrc"
rm -f "$ROOT/build/$NAME.so" "$ROOT"/build/$NAME/*.o
pids=""
for f in rtx_lines rtx_voigt rtx_voigt_scatter rtx_sdvoigt rtx_tud rtx_radiance rtx_resample rtx_comm; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $EXTRA -Wno-unused-function -c $f.hip -o "$ROOT/build/$NAME/$f.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile failed for variant $NAME" >&2; exit 1; }; done
hipcc -shared --offload-arch=gfx950 -o "$ROOT/build/$NAME.so" "$ROOT"/build/$NAME/*.o -ldl
echo "$ROOT/build/$NAME.so"
