#!/bin/bash
# Build libstabstitch_hip.so for gfx950 (MI355X) in-tree.
#   stabstitch2_amd/csrc/build.sh           -> stabstitch2_amd/libstabstitch_hip.so         (the product)
#   stabstitch2_amd/csrc/build.sh tuning    -> tools/libstabstitch_hip_tuning.so            (-DSS_TUNING: adds the
#       ss_debug_set / ss_debug_ptr knobs and the per-workgroup s_memtime stamps used by tools/ab_*.py, diag_phases.py)
# -ffp-contract=off: the samplers reproduce the reference's separate mul/add sequence (its out-of-range taps cancel
# exactly only without fma contraction); every intended fma is an explicit fmaf()/MFMA in the sources.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libstabstitch_hip.so"
DEFS=""
if [ "$1" = "tuning" ]; then
    OUT="$HERE/../../tools/libstabstitch_hip_tuning.so"
    DEFS="-DSS_TUNING"
fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden $DEFS \
    "$HERE/conv.hip" "$HERE/wino.hip" "$HERE/corr.hip" "$HERE/geom.hip" "$HERE/render.hip" "$HERE/smooth.hip" "$HERE/metrics.hip" \
    "$HERE/frameio.hip" "$HERE/stem.hip" "$HERE/wino43.hip" \
    -o "$OUT"
echo "built $OUT"
