set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD TMPDIR=/tmp; ROOT=$PWD
cd /tmp
for M in 0 2; do
  export VISFS_BA_FRAME_GRAPH=$M
  for CFG in C2 PROD; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/r04_l_stats_${CFG}_m$M" -o f -- python3 "$ROOT/tools/frame_loop.py" $CFG 40 > "$O/r04_l_${CFG}_m$M.log" 2>/dev/null
  done
done
cd $ROOT
for M in 0 2; do for CFG in C2 PROD; do echo "== $CFG mode $M: $(cat $O/r04_l_${CFG}_m$M.log)"; python3 - $O/r04_l_stats_${CFG}_m$M/f_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"   total kernel time {tot/1e3/43:.1f} us per call")
for r in rows[:9]:
    n=r['Name'].replace('visfs_ba::','').replace('void ','')
    print(f"  {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.2f} us  {float(r['Percentage']):5.1f}%  {n[:100]}")
PY
done; done
