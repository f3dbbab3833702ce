cd $GRAFT_REPO_ROOT
for B in 256; do
echo "== W4=1 SOLO B=$B staged"; R3D_W4=1 R3D_W4_SOLO=1 R3D_STAGED=1 python tools/stage_times.py $B 15 2>&1 | tail -15
echo "== W4=1 B=$B staged"; R3D_W4=1 R3D_STAGED=1 python tools/stage_times.py $B 15 2>&1 | tail -15
echo "== W4=0 B=$B staged"; R3D_W4=0 R3D_STAGED=1 python tools/stage_times.py $B 15 2>&1 | tail -15
done
bash tools/ab_env.sh 256 "R3D_W4=0" "R3D_W4=1 R3D_W4_SOLO=1" "R3D_W4=1 R3D_W4_SOLO=1 R3D_STAGED=1"
