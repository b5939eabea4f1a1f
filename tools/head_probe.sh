#!/bin/bash
# baseline probes of the current kernel: phase cycle counters, SQ counter passes, per-wave timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
V2P_DEBUG=1 V2P_PHASE_TIMING=1 timeout 300 python bench.py --steps 64 --warmup 0 --no-cpu-baseline > $O/phase.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_head.log 2>&1
V2P_DEBUG=1 V2P_WAVE_TIMES=$O/wave_times.bin timeout 300 python bench.py --steps 29 --warmup 0 --no-cpu-baseline > $O/wt.log 2>&1
python tools/wave_times.py $O/wave_times.bin > $O/wave_times.txt 2>&1; rm -f $O/wave_times.bin
bash tools/valu_probe.sh > $O/valu.log 2>&1
grep -E "phase" $O/phase.log | cut -c1-1500; tail -1 $O/bench_head.log | cut -c1-400; tail -20 $O/wave_times.txt
