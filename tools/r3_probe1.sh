set -u
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ppo_reference.py tests/test_gpu_ppo.py -q -x 2>&1 | tail -25 > $O/t_ppo.log
tail -5 $O/t_ppo.log
for v in "" "--ppo-no-overlap"; do
  echo "[ppo $v]"; timeout 400 python bench.py --ppo --ppo-epochs 3 $v 2>&1 | tail -2 | cut -c1-700
done > $O/bench_ppo_r3a.log 2>&1
cat $O/bench_ppo_r3a.log
