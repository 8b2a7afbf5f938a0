cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for cfg in "7 6144 4" "7 7168 4" "8 6144 4" "8 8192 4" "8 7168 4"; do
  set -- $cfg
  r=$(PLSLAM_HIP_LIB=$PWD/gpu_dbg/libplslam_hip_w$1.so timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 --batch $2 --nsplit $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "waves $1 batch $2 nsplit $3: $r"
done
