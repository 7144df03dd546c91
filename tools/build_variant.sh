#!/bin/bash
# A/B measurement build of the HIP library with extra -D flags: tools/build_variant.sh <name> -DVISFS_BA_TILE_N=0 ...
# -> visfs_amd/lib/libvisfs_ba_hip_<name>.so; select it with VISFS_BA_LIB=<path> (visfs_amd/backend.py).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o visfs_amd/lib/libvisfs_ba_hip_$name.so visfs_amd/csrc/ba_kernels.hip visfs_amd/csrc/ba_api.cpp -lpthread
