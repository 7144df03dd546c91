set -o pipefail
O=gpurun_out; mkdir -p $O; export PYTHONPATH=$PWD
( timeout -k 10 400 python3 tools/soak_random.py 6000 7200 > $O/r04_soak_random.log 2>&1; echo "rc=$?" >> $O/r04_soak_random.log; tail -3 $O/r04_soak_random.log )
( timeout -k 10 250 python3 tools/soak_ceres.py 2400 3000 0 > $O/r04_soak_ceres.log 2>&1; echo "rc=$?" >> $O/r04_soak_ceres.log; tail -2 $O/r04_soak_ceres.log )
( timeout -k 10 250 python3 tools/soak_ceres.py 2400 3000 1 > $O/r04_soak_dogleg.log 2>&1; echo "rc=$?" >> $O/r04_soak_dogleg.log; tail -2 $O/r04_soak_dogleg.log )
( timeout -k 10 200 python3 tools/soak_batch.py 1500 1800 12 > $O/r04_soak_batch.log 2>&1; echo "rc=$?" >> $O/r04_soak_batch.log; tail -2 $O/r04_soak_batch.log )
( VISFS_BA_BATCH_SPEC=1 timeout -k 10 200 python3 tools/soak_batch.py 1500 1800 12 > $O/r04_soak_batch_fused_unit.log 2>&1; echo "rc=$?" >> $O/r04_soak_batch_fused_unit.log; tail -2 $O/r04_soak_batch_fused_unit.log )
( timeout -k 10 200 python3 tools/soak_frames.py 60 > $O/r04_soak_frames.log 2>&1; echo "rc=$?" >> $O/r04_soak_frames.log; tail -2 $O/r04_soak_frames.log )
( VISFS_BA_FRAME_GRAPH=2 timeout -k 10 200 python3 tools/soak_frames.py 60 > $O/r04_soak_frames_replay.log 2>&1; echo "rc=$?" >> $O/r04_soak_frames_replay.log; tail -2 $O/r04_soak_frames_replay.log )
