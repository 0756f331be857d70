#!/bin/bash
# Power, clocks and temperature while one op of the bench runs back to back: tools/power_probe.sh [verify|x25519|sign]
OP=${1:-verify}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --op $OP --steps 4000 --warmup 2 --cpu-sample 4096 > gpurun_out/pw_bench_$OP.log 2>&1 &
PID=$!
sleep 22
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | head -6; echo ---; sleep 0.7; done
wait $PID
tail -1 gpurun_out/pw_bench_$OP.log | cut -c1-160
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
