set -x
python -m pytest tests -q --tb=short -m gpu 2>&1 | tail -15
python __graft_entry__.py smoke 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head -20
