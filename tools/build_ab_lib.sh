#!/bin/bash
# A/B or probe build of ONE kernel source: tools/libmmg_ab_<name>.so = the product library with <source>.hip recompiled under extra defines.
#   tools/build_ab_lib.sh ntprobe gemm_bf16 -DNT_PROBE        tools/build_ab_lib.sh bwdw_probe cnblock_bwdw -DBW_PROBE
# (run after `make -C mmg-clip_amd/csrc`; MMGCLIP_HIP_LIB=tools/libmmg_ab_<name>.so selects it; the .so files are git-ignored)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/mmg-clip_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-value -ffp-contract=off "$@" -c $src.hip -o /tmp/mmg_ab_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/libmmg_ab_$name.so" /tmp/mmg_ab_$name.o $(ls *.o | grep -v "^$src.o$")
echo "built tools/libmmg_ab_$name.so"
