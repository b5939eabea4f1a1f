set -u
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_racket_ball.py tests/test_gpu_physics.py -q -x 2>&1 | tail -8 > $O/t_phys.log
tail -4 $O/t_phys.log
for v in "--racket-ball" "--racket-ball --joint-limits 0" "--racket-ball --ball-body-contacts 0"; do echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | cut -c1-200)"; done
