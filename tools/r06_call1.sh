cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_timed_path.py -x -q 2>&1 | tail -15
SERIAL=1 VARIANTS=$'\n-DPR_CFG128=16,512,0,1,2\n-DPR_CFG128=16,512,0,1,2 -DPR_NP=8 -DPR_CHUNK_N=256 -DPR_FLAGS_N=512\n-DPR_NP=8' bash tools/r06_ab.sh 2>&1 | tee gpurun_out/r06_ab1.txt
