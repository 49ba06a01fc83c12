#!/bin/bash
# Build libstabstitch_hip.so for gfx950 (MI355X) in-tree.  Usage: stabstitch2_amd/csrc/build.sh
# -ffp-contract=off: the samplers reproduce the reference's separate mul/add sequence (its out-of-range taps cancel
# exactly only without fma contraction); every intended fma is an explicit fmaf()/MFMA in the sources.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libstabstitch_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
    "$HERE/conv.hip" "$HERE/corr.hip" "$HERE/geom.hip" "$HERE/render.hip" "$HERE/smooth.hip" "$HERE/metrics.hip" \
    "$HERE/frameio.hip" \
    -o "$OUT"
echo "built $OUT"
