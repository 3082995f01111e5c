export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 400 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | grep "#3\|#5\|#2 "
