cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; done
