set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_racket_ball.py tests/test_gpu_vec_task.py tests/test_gpu_ppo.py -q -x 2>&1 | grep -v "^E    .*array\|^E   .*where" | tail -30 | cut -c1-400
for v in "--racket-ball" "--racket-ball --per-clip-shapes"; do echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | cut -c1-200)"; done
