#!/bin/bash
# What the matrix pipes sustain with every CU busy, by operand data (random / post-ReLU-like / zeros), with the socket power
# and shader clock rocm-smi reports meanwhile.  Output -> profiles/r03_power_ceiling.txt
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
echo "idle: $(smi)"
for data in 0 1 2; do
  tools/ubench/mfma_lds $data 3000 1 > /tmp/ub_$data.txt &     # ~2.5 s of back-to-back launches of one tile shape
  pid=$!
  sleep 1.6; echo "  during data=$data: $(smi)"
  wait $pid; tail -1 /tmp/ub_$data.txt
done
echo "--- tile shapes x active CUs (grid 256 / 64 / 8), random operands"
tools/ubench/mfma_lds 0 2
echo "--- the same with post-ReLU-like B operands (about half zeros)"
tools/ubench/mfma_lds 1 2 | head -8
