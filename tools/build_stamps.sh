#!/bin/bash
# Diagnostic build of the HIP library with in-kernel real-time stamps (-DVISFS_BA_STAMPS): visfs_amd/lib/libvisfs_ba_hip_stamps.so.
# Never quote run times of this build; read the SHARES of its stamps (tools/pcg_stamps.py, tools/small_solve_stamps.py).
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVISFS_BA_STAMPS -o visfs_amd/lib/libvisfs_ba_hip_stamps.so visfs_amd/csrc/ba_kernels.hip visfs_amd/csrc/ba_api.cpp -lpthread
