#!/bin/bash
# Samples rocm-smi clocks / power while a command keeps the GPU busy; prints up to 4 samples taken under load (sclk > 500 MHz):
#   tools/clk_watch.sh <label> <command ...>
label=$1; shift
"$@" > /tmp/clk_cmd.out 2>&1 &
pid=$!
n=0
while kill -0 $pid 2>/dev/null; do
  s=$(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|fclk|mclk|Power \(W\)' | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/; s/.*\(W\): *([0-9.]+).*/\1W/' | tr '\n' ' ')
  sclk=$(echo $s | awk '{print int($3)}')
  if [ "${sclk:-0}" -gt 500 ] && [ $n -lt 4 ]; then echo "[$label] fclk mclk sclk power = $s"; n=$((n+1)); fi
  sleep 0.3
done
tail -1 /tmp/clk_cmd.out
