cd $GRAFT_REPO_ROOT
(python3 bench.py --steps 2500 --warmup 2 --cpu-sample 4096 > gpurun_out/pw_bench.log 2>&1 &) 
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | head -6; echo ---; sleep 0.7; done
sleep 3
tail -1 gpurun_out/pw_bench.log | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
