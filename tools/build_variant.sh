#!/usr/bin/env bash
# Scratch variant of libeben_hip.so: ONE source recompiled with extra flags, linked against the other objects of the regular build.
# Usage: tools/build_variant.sh <name> <source stem> <extra hipcc flags...>   -> vibravox_amd/lib/var/libeben_<name>.so  (EBEN_HIP_LIB selects it)
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"; name=$1; stem=$2; shift 2
cs=$root/vibravox_amd/csrc; obj=$root/vibravox_amd/lib/obj; out=$root/vibravox_amd/lib/var; mkdir -p $out /tmp/eben_var
src=$cs/$stem.hip; [ -f $src ] || src=$cs/exact_fp32/$stem.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$root/include -I$cs -DEBEN_BUILDING=1 "$@" -c $src -o /tmp/eben_var/${stem}_$name.o
objs=(); for o in $obj/*.o; do [ "$(basename $o)" = "$stem.o" ] || objs+=($o); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" /tmp/eben_var/${stem}_$name.o -o $out/libeben_$name.so
echo built $out/libeben_$name.so
