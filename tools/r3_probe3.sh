set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_racket_ball.py -q -s 2>&1 > $O/t_rows_full.log
grep "^\[rows\]\|^\[outliers\]\|passed\|failed\|Error\|assert" $O/t_rows_full.log > $O/t_rows.log
tail -5 $O/t_rows_full.log | cut -c1-300; rm -f $O/t_rows_full.log
grep -c "outliers" $O/t_rows.log
timeout 600 python tools/limit_cost.py > $O/limit_cost.txt 2>&1; cat $O/limit_cost.txt
