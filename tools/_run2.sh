cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q -k "full_size or fixture or bench or smoke or clip_calls_run or h36m or cfg4" ) > gpurun_out/r05_run2_tests.log 2>&1
tail -4 gpurun_out/r05_run2_tests.log
bash tools/ab_env.sh 256 "R3D_NO_NB=1" "R3D_NO_NB=0" > gpurun_out/r05_nb_ab_256.txt 2>&1
cat gpurun_out/r05_nb_ab_256.txt
bash tools/ab_env.sh 128 "R3D_NO_NB=1" "R3D_NO_NB=0" > gpurun_out/r05_nb_ab_128.txt 2>&1
cat gpurun_out/r05_nb_ab_128.txt
