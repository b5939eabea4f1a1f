set -u
O=gpurun_out; mkdir -p $O
timeout 900 python tools/soak.py 6000 2>&1 | tail -4 > $O/soak.log; cat $O/soak.log
timeout 900 python tools/soak.py 6000 racket 2>&1 | tail -4 > $O/soak_racket_ball.log; cat $O/soak_racket_ball.log
timeout 600 python -m pytest tests/test_gpu_vec_task.py -q 2>&1 | tail -1
