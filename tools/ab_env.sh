# Scratch A/B harness: the bench step under environment / library variants, interleaved so that box drift cancels.
# Usage: bash tools/ab_env.sh LABEL1 "ENV1=.. ENV2=.." LABEL2 "ENV=.." ...   (prints ms/step per variant and repetition)
# A library variant is a scratch build: bash vibravox_amd/csrc/build.sh vibravox_amd/lib/var_x -DEBEN_...=..; EBEN_HIP_LIB=<path>.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { # label, env assignments...
  local label=$1; shift
  local ms=$(env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$label $ms"
}
[ $# -eq 0 ] && set -- base "X=1" devkernarg "HIP_FORCE_DEV_KERNARG=1"
for rep in 1 2 3; do
  args=("$@")
  for ((i = 0; i < ${#args[@]}; i += 2)); do run "${args[i]}" ${args[i+1]}; done
done
