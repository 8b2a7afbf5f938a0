cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q --timeout 600 -k "line or e2e or adaptor" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_launch'])"
