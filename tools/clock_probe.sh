#!/bin/bash
# Samples shader clock / power with rocm-smi while a workload runs (one line per sample): is a kernel class clock-limited by power?
# usage: tools/clock_probe.sh <label> <command...>
label=$1; shift
"$@" > /tmp/clock_probe_cmd.log 2>&1 &
pid=$!
sleep 6
while kill -0 $pid 2>/dev/null; do
  s=$(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';')
  echo "$label $s"
  sleep 1
done
wait $pid
tail -n 1 /tmp/clock_probe_cmd.log
