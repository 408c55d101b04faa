#!/usr/bin/env bash
# A/B of library BUILDS on one box: every variant is built once (extra hipcc flags), then the
# variants take turns, ROUNDS times each, one process per turn (tools/ab_options.py base, with
# the per-family kernel times).  Usage on the GPU box:
#   tools/ab_builds.sh "base:" "chunk16:-DRN_XCD_CHUNK=16" ...
#   CONFIG=config4 ROUNDS=3 STEPS=10 tools/ab_builds.sh ...
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${ROUNDS:-3}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -fPIC -shared"
labels=()
for v in "$@"; do
  label=${v%%:*}; flags=${v#*:}
  labels+=("$label")
  ( cd $R/raynet_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $flags -I $R/include raynet_hip.hip -o /tmp/ab_$label.so 2>&1 | grep -E "error" | head -3 ) &
done
wait
for r in $(seq $ROUNDS); do
  for label in "${labels[@]}"; do
    LIB=/tmp/ab_$label.so PROF=${PROF:-1} ROUNDS=1 STEPS=${STEPS:-20} python $R/tools/ab_options.py ${OPTS:-base} 2>/dev/null | tail -1 | sed "s/^base */$label /" >> /tmp/ab_$label.txt
  done
done
for label in "${labels[@]}"; do
  python - "$label" <<'PY'
import re, sys, statistics
label = sys.argv[1]
rows = open("/tmp/ab_%s.txt" % label).read().strip().splitlines()
ms = [float(re.search(r"median ([0-9.]+)", r).group(1)) for r in rows]
fam = {}
for r in rows:
    for k, v in re.findall(r"(\w+)=([0-9.]+)", r):
        fam.setdefault(k, []).append(float(v))
print("%-22s %.3f ms/step (runs %s)  %s" % (label, statistics.median(ms), " ".join("%.3f" % m for m in ms),
      " ".join("%s=%.3f" % (k, statistics.median(v)) for k, v in sorted(fam.items()))))
PY
  rm -f /tmp/ab_$label.txt
done
