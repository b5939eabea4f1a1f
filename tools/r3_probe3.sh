set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_physics.py -q -s 2>&1 > $O/t_rows_full.log
grep "^\[rows\]\|^\[outliers\]\|^\[limits\]\|passed\|failed\|Error\|assert" $O/t_rows_full.log > $O/t_rows.log
grep -v "^\[rows\]\|^\[selection\]" $O/t_rows_full.log | tail -30 | cut -c1-300; rm -f $O/t_rows_full.log
