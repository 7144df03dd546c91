#!/bin/bash
# Diagnostic: the graph build of visfs_ba_solve_window on host threads under Python (VISFS_BA_THREADS), workers placed on the caller's L3
# domain (default) or left to the scheduler (VISFS_BA_POOL_AFFINITY=0).  usage: tools/thread_scaling.sh
export PYTHONPATH=$PWD
lscpu | grep -i "model name\|socket\|l3\|numa node(s)"
for aff in 1 0; do
for t in 1 2 4 8; do
  echo "## VISFS_BA_THREADS=$t VISFS_BA_POOL_AFFINITY=$aff"
  VISFS_BA_POOL_AFFINITY=$aff VISFS_BA_THREADS=$t python tools/upload_laps.py C4 C2 2>&1 | grep -A40 "solve_window #2" | grep "tasks\|window prepare\|window finish"
done
done
