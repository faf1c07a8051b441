run() { # run <label> <env...> -- <bench args>
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local out
  out=$(env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f vol/s %.3f ms'%(d['value'], d['ms_per_step']))")
  echo "$label: $out"
}
for rep in 1 2; do
run "bf16 default" X=1 -- --dtype bf16 --batch 8 --steps 10 --warmup 3
run "bf16 TXL=5" BTS_LP_S1D_TXL=5 -- --dtype bf16 --batch 8 --steps 10 --warmup 3
run "bf16 TXL=4" BTS_LP_S1D_TXL=4 -- --dtype bf16 --batch 8 --steps 10 --warmup 3
run "bf16 nofuse gn1" BTS_LP_FUSE_GN1_BWD=0 -- --dtype bf16 --batch 8 --steps 10 --warmup 3
run "infer default" X=1 -- --infer --dtype f16 --steps 20 --warmup 5
run "infer TXL=5" BTS_LP_S1D_TXL=5 -- --infer --dtype f16 --steps 20 --warmup 5
run "infer TXL=4" BTS_LP_S1D_TXL=4 -- --infer --dtype f16 --steps 20 --warmup 5
run "infer S1Z=0" BTS_LP_S1Z=0 -- --infer --dtype f16 --steps 20 --warmup 5
done
