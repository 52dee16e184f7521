#!/bin/bash
# GPU clock / power while bench.py runs (is the step kernel power-limited?): tools/clock_probe.sh <lib.so> ...  ->  gpurun_out/clk_<lib>.txt
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 -c "import torch" 2>/dev/null
for lib in "$@"; do
  USIM_LIB=$PWD/robotic-ultrasound-imaging_amd/lib/$lib python3 bench.py --no-cpu-baseline --steps 2000000 --warmup 100 > gpurun_out/clk_$lib.json 2>/dev/null &
  BP=$!
  : > gpurun_out/clk_$lib.txt
  while kill -0 $BP 2>/dev/null; do
    echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr '\n' ' ')" >> gpurun_out/clk_$lib.txt
    sleep 2
  done
  tail -1 gpurun_out/clk_$lib.json | cut -c1-200
done
