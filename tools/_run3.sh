cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nb in 0; do
R3D_NO_NB=$nb R3D_LIB_OVERRIDE=tools/libray3d_hip_timing.so R3D_TIMING_STAGE=all R3D_TIMING_DUMP=gpurun_out/gantt_256_nonb$nb.txt python tools/stage_times.py 256 1 > gpurun_out/gantt_256_nonb$nb.log 2>&1
python tools/fwd_gantt.py gpurun_out/gantt_256_nonb$nb.txt > gpurun_out/gantt_256_nonb${nb}_summary.txt 2>&1
grep -i "Integration\|span\|waiting\|idle\|busy" gpurun_out/gantt_256_nonb${nb}_summary.txt
done
(time timeout 900 python -m pytest tests -m gpu -x -q -k "every_mode_and_form or full_size or bench or ragged or narrow" ) > gpurun_out/r05_run3_tests.log 2>&1
tail -4 gpurun_out/r05_run3_tests.log
bash tools/ab_env.sh 256 "R3D_NO_NB=1" "R3D_NO_NB=0" > gpurun_out/r05_nb_ab_256.txt 2>&1
cat gpurun_out/r05_nb_ab_256.txt
bash tools/ab_env.sh 128 "R3D_NO_NB=1" "R3D_NO_NB=0" > gpurun_out/r05_nb_ab_128.txt 2>&1
cat gpurun_out/r05_nb_ab_128.txt
bash tools/ab_env.sh 192 "R3D_NO_NB=1" "R3D_NO_NB=0" > gpurun_out/r05_nb_ab_192.txt 2>&1
cat gpurun_out/r05_nb_ab_192.txt
