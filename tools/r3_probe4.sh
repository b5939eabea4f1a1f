set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_racket_ball.py tests/test_gpu_task_ops.py -q -x 2>&1 | grep -v "^E    .*array\|^E   .*where" | tail -12 | cut -c1-300
for i in 1 2; do for v in "" "--racket-ball"; do echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | cut -c1-130)"; done; done
