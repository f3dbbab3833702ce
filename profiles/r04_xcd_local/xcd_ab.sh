cd $GRAFT_REPO_ROOT
R3D_USE_HOOKS_LIB=1 R3D_XCD_ALIGN=1 R3D_XCD_LOCAL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_window or reference_fixture_in_every or window_counts" 2>&1 | tail -4
bash tools/traffic_ab.sh "R3D_XCD_ALIGN=0" "R3D_XCD_ALIGN=1" "R3D_XCD_ALIGN=1 R3D_XCD_LOCAL=1"
