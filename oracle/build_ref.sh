#!/usr/bin/env bash
# Builds the reference's OWN voxel traversal (raynet/ray_marching/ray_tracing.pyx)
# from where it lies under /root/reference, into oracle/_ref/ (git-ignored, but it
# travels to the GPU box with the snapshot).  No reference source is copied into
# the repo: cython reads the .pyx in place and writes only generated C + the .so
# under oracle/_ref/.  We do not run the reference's setup.py.
#
# The reference's six .cu files (its CUDA kernels) are built for gfx950 by
# oracle/build_ref_cu.py, called at the end of this script: Template-substituted as
# the reference itself does at run time and otherwise unchanged, they compile with
# hipcc as they are (rounds 1-4 had wrongly filed them as unbuildable).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${RAYNET_REFERENCE:-/root/reference}"
PYX="$REF/raynet/ray_marching/ray_tracing.pyx"
OUT="$HERE/_ref"
if [ ! -f "$PYX" ]; then
    echo "build_ref: $PYX not present (not the build container) - keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT"
PY=python3
INC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
SUFFIX=$($PY -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
cython -3 "$PYX" -o "$OUT/ray_tracing.c" 2>/dev/null
# gcc -O2 on x86-64 emits no FMA: pure fp32 add/mul/div, the arithmetic the
# bit-exact index-map parity is defined against.
gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -I"$INC" \
    "$OUT/ray_tracing.c" -o "$OUT/ray_tracing$SUFFIX" -lm
rm -f "$OUT/ray_tracing.c"   # keep only the binary
echo "built $OUT/ray_tracing$SUFFIX"
# the reference-kernel code objects are test infrastructure: a compiler that is missing or refuses one
# of them must not fail the product build (the tests that need them skip without the manifest)
$PY "$HERE/build_ref_cu.py" || echo "build_ref: reference .cu code objects NOT built (tests/test_reference_kernels.py will skip)" >&2
