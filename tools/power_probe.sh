#!/bin/bash
# sample board power / clocks while a command runs: tools/power_probe.sh <logfile> -- <command...>
log=$1; shift 2
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' >> $log
  echo >> $log
  sleep 0.25
done
wait $pid
