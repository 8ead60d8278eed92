cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sds.py -m gpu -x -q -k "graphed" > gpurun_out/r4b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4b_pytest.txt
timeout 600 python tools/config_step_profile.py 2 > gpurun_out/r4b_config2.txt 2>&1; cat gpurun_out/r4b_config2.txt | grep -v Warning | head -34
for T in 32768 65536 131072; do echo "== MVIP_BWD_TILE_POINTS=$T"; MVIP_BWD_TILE_POINTS=$T timeout 600 python tools/config_step_profile.py 2 2>&1 | grep -E "iteration:|mlp_" | head -8; done
timeout 600 python tools/config_step_profile.py 3 > gpurun_out/r4b_config3.txt 2>&1; cat gpurun_out/r4b_config3.txt | grep -v Warning | head -24
