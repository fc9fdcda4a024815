#!/bin/bash
# Samples the shader clock and socket power (rocm-smi) while a command runs:  tools/clock_watch.sh OUT.txt cmd args...
out=$1; shift
"$@" > "${out%.txt}_cmd.txt" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> "$out"
  echo "--" >> "$out"
  sleep 0.5
done
wait $pid
