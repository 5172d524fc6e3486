#!/bin/bash
# samples socket power and the shader / memory / fabric clocks while a command runs (tools only): power_watch.sh <tag> <cmd...>
# prints a time line (one sample per 0.5 s: seconds, power, sclk, mclk, fclk) so that mode switches of a long run can be
# laid beside the per-run step times in gpurun_out/pw_<tag>.log
tag=$1; shift
mkdir -p gpurun_out
"$@" > gpurun_out/pw_$tag.log 2>&1 &
pid=$!
t0=$(date +%s.%N)
while kill -0 $pid 2>/dev/null; do
  s=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk" | sed -E 's/.*(sclk|mclk|fclk)[^(]*\(([0-9]+Mhz)\).*/\1=\2/; s/.*Power \(W\): *([0-9.]+).*/P=\1/' | tr '\n' ' ')
  echo "$(echo "$(date +%s.%N) - $t0" | bc | cut -c1-5) $s"
  sleep 0.5
done > gpurun_out/pw_$tag.smi
grep "ms/step" gpurun_out/pw_$tag.log
