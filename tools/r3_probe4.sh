set -u
for v in "--groups 2" "--groups 4" "--groups 2 --racket-ball"; do echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1 | cut -c1-130)"; done
