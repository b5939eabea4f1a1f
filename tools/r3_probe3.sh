set -u
timeout 900 python -m pytest tests/test_gpu_racket_ball.py -q -x 2>&1 | grep -v "^E    .*array\|^E   .*where" | tail -12 | cut -c1-300
for v in "--racket-ball" "--racket-ball --joint-limits 0"; do echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | cut -c1-130)"; done
