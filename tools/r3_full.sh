set -u
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
