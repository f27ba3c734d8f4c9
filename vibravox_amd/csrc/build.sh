#!/usr/bin/env bash
# Build libeben_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="${1:-$here/../lib}"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
  -Wall -Wno-unused-function \
  -I"$root/include" -I"$here" \
  -DEBEN_BUILDING=1 \
  "$here/tapconv.hip" "$here/tapconv2.hip" "$here/tapconv3.hip" "$here/thinconv.hip" "$here/conv_dw.hip" "$here/conv_dw2.hip" "$here/conv_dw3.hip" "$here/direct.hip" \
  -o "$out/libeben_hip.so" "${@:2}"
echo "built $out/libeben_hip.so"
