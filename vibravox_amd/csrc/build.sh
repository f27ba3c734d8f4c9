#!/usr/bin/env bash
# Build libeben_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir] [extra hipcc flags...]
# One object per source, compiled in parallel and rebuilt only when the source, a header or the flags changed.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${1:-$here/../lib}"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
extra=("${@:2}")
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function
       -I"$root/include" -I"$here" -DEBEN_BUILDING=1 "${extra[@]}")
obj="$here/../lib/obj"
mkdir -p "$obj"
stamp="$(printf '%s\n' "${flags[@]}" | cat - "$here"/*.h "$root"/include/*.h | md5sum | cut -d' ' -f1)"
srcs=("$here"/*.hip "$here"/exact_fp32/*.hip)   # exact_fp32/: the kernel families only the exact-fp32 reference plan reaches
jobs="${EBEN_BUILD_JOBS:-$(nproc)}"
pids=()
fail=0
for s in "${srcs[@]}"; do
  b="$(basename "$s" .hip)"
  o="$obj/$b.o"
  tag="$obj/$b.stamp"
  want="$stamp $(md5sum < "$s" | cut -d' ' -f1)"
  if [[ -f "$o" && -f "$tag" && "$(cat "$tag")" == "$want" ]]; then continue; fi
  ( "$HIPCC" "${flags[@]}" -c "$s" -o "$o" && echo "$want" > "$tag" ) &
  pids+=($!)
  while (( $(jobs -rp | wc -l) >= jobs )); do wait -n || fail=1; done
done
for p in "${pids[@]}"; do wait "$p" || fail=1; done
(( fail == 0 )) || { echo "build failed" >&2; exit 1; }
objs=()
for s in "${srcs[@]}"; do objs+=("$obj/$(basename "$s" .hip).o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libeben_hip.so"
echo "built $out/libeben_hip.so"
