#!/bin/bash
# MX-fp8 MFMA layout / rate probe with socket power and shader clock sampled beside it.  Output -> profiles/r04_mx_probe.txt
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
echo "idle: $(smi)"
tools/ubench/mx_probe 2.0 > /tmp/mx.txt &
pid=$!
for i in $(seq 1 14); do sleep 1.5; echo "  t=$((i*3/2))s: $(smi)"; done
wait $pid
cat /tmp/mx.txt
