cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh 256 "R3D_TOP_PAIR=0" "R3D_TOP_PAIR=1"
bash tools/ab_env.sh 1024 "R3D_TOP_PAIR=0" "R3D_TOP_PAIR=1"
