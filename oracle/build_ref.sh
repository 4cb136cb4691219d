#!/bin/bash
# Build the REAL reference (microsoft/maro: Python + 4 Cython extensions) out of /root/reference into a scratch folder, so that
# oracle/gen_golden*.py and oracle/check_*_dropin.py can import it.  ORACLE TOOLING: nothing under maro_amd/ uses the result,
# and the GPU box never sees it (the goldens under tests/golden/ are what travels).  Recipe = SURVEY.md §8(c)
# (scripts/compile_cython.sh:17 + setup.py:45-85 of the reference); ~60 s.
# Usage: oracle/build_ref.sh [dest=/tmp/oracle] [pack_dir]
#   pack_dir (e.g. oracle/_ref, git-ignored): additionally write pack_dir/maro_ref.tgz = the RUNTIME of that build only — the
#   `maro` packages the timed paths import (simulator, backends with the compiled extension modules, data_lib, event_buffer,
#   utils, vector_env, rl + examples/cim/rl for config 5's sampler) plus the import stubs — so that bench.py's `cpu_baseline_reference` leg can time the reference's own
#   Env.step / VectorEnv ON THE GPU BOX's host cores (the archive travels with the snapshot like the built .so files; it is a
#   build output, never committed, and only bench.py's cpu-baseline leg unpacks it, into a temp folder).
set -euo pipefail
REF=${MARO_REFERENCE:-/root/reference}
DEST=${1:-/tmp/oracle}
PACK=${2:-}
if [ -n "$PACK" ]; then mkdir -p "$PACK"; PACK=$(cd "$PACK" && pwd); fi   # (absolute: the script changes directory below)
mkdir -p "$DEST/home"
if [ ! -d "$DEST/maro_src" ]; then cp -r "$REF" "$DEST/maro_src"; fi
cd "$DEST/maro_src"
if ! ls maro/backends/frame.cpython-*.so >/dev/null 2>&1; then
  cython maro/backends/backend.pyx maro/backends/np_backend.pyx maro/backends/raw_backend.pyx maro/backends/frame.pyx \
      --cplus -3 -E NODES_MEMORY_LAYOUT=ONE_BLOCK -X embedsignature=True
  python3 setup.py build_ext -i > "$DEST/build_ext.log" 2>&1
fi
# stubs for packages the reference imports at module import but that the hot path never calls (SURVEY.md §8c caveats 2, 3)
mkdir -p "$DEST/stubs/holidays" "$DEST/stubs/geopy" "$DEST/stubs/zmq/eventloop" "$DEST/stubs/tornado"
[ -f "$DEST/stubs/holidays/__init__.py" ] || cat > "$DEST/stubs/holidays/__init__.py" <<'PY'
class US:
    def __init__(self, *a, **k): pass
    def __contains__(self, d): return False
PY
[ -f "$DEST/stubs/geopy/__init__.py" ] || { echo "" > "$DEST/stubs/geopy/__init__.py"; cat > "$DEST/stubs/geopy/distance.py" <<'PY'
import math
class distance:
    """haversine stand-in: only the ORDER of neighbour distances matters to the citi_bike topology generator"""
    def __init__(self, a, b):
        la1, lo1, la2, lo2 = map(math.radians, (a[0], a[1], b[0], b[1]))
        h = math.sin((la2 - la1) / 2) ** 2 + math.cos(la1) * math.cos(la2) * math.sin((lo2 - lo1) / 2) ** 2
        self.km = 2 * 6371.0088 * math.asin(math.sqrt(h))
        self.kilometers = self.km
PY
}
HOME="$DEST/home" SKIP_DEPLOYMENT=TRUE PYTHONPATH="$DEST/maro_src" python3 - <<'PY'
from maro.simulator import Env
env = Env("cim", "toy.4p_ssdd_l0.0", durations=20)
m, de, done = env.step(None)
print("reference Env ok: first decision at tick", de.tick)
PY
if [ -n "$PACK" ]; then
  STAGE=$(mktemp -d)
  mkdir -p "$STAGE/maro_ref/maro"
  ( cd "$DEST/maro_src/maro" && cp *.py "$STAGE/maro_ref/maro/" )
  for sub in simulator backends data_lib event_buffer utils vector_env rl; do
    ( cd "$DEST/maro_src/maro" && tar -cf - --exclude='*.cpp' --exclude='*.c' --exclude='*.pyx' --exclude='*.pxd' --exclude='__pycache__' \
        --exclude='raw' --exclude='vm_scheduling' "$sub" ) | tar -xf - -C "$STAGE/maro_ref/maro"
  done
  # the handful of cli / streamit modules that `import maro.simulator` pulls in at import time (logger params, the citi_bike data
  # pipeline's names, the streamit no-op client)
  ( cd "$DEST/maro_src/maro" && tar -cf - --exclude='__pycache__' cli/__init__.py cli/utils/__init__.py cli/utils/params.py cli/data_pipeline streamit ) \
      | tar -xf - -C "$STAGE/maro_ref/maro"
  # config 5's CPU baseline runs the reference's own CIMEnvSampler: maro.rl (above) + the CIM RL example's shaping code
  # ... and the two examples the GPU drop-in tests run unchanged against the HIP engine (tests/test_gpu_dropin.py)
  ( cd "$DEST/maro_src" && tar -cf - --exclude='__pycache__' examples/__init__.py examples/cim/rl examples/vector_env/hello.py \
        examples/citi_bike/greedy ) | tar -xf - -C "$STAGE/maro_ref"
  cp -r "$DEST/stubs" "$STAGE/maro_ref/stubs"
  ( cd "$STAGE" && tar -czf "$PACK/maro_ref.tgz" maro_ref )
  rm -rf "$STAGE"
  echo "runtime archive: $PACK/maro_ref.tgz ($(du -h "$PACK/maro_ref.tgz" | cut -f1))"
fi
echo "reference built in $DEST/maro_src  (PYTHONPATH=$DEST/maro_src HOME=$DEST/home)"
