#!/bin/bash
# clock_watch.sh <out> -- <cmd...>: samples shader clock and board power of EVERY amdgpu hwmon node from sysfs every ~20 ms
# while <cmd> runs (a box may expose several cards; the summaries take the busiest one).  One line per node and sample.
OUT=$1; shift; shift
HS=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
echo "# hwmon nodes: $(echo $HS | tr '\n' ' ')" > $OUT
for H in $HS; do echo "# pci $(basename $(readlink -f $H/device)) node $H" >> $OUT; done
"$@" &
PID=$!
while kill -0 $PID 2>/dev/null; do
  T=$(date +%s.%N)
  for H in $HS; do
    echo "$T sclk $(cat $H/freq1_input 2>/dev/null) mclk $(cat $H/freq2_input 2>/dev/null) power $(cat $H/power1_average 2>/dev/null) $(cat $H/power1_input 2>/dev/null) node $H" >> $OUT
  done
  sleep 0.02
done
wait $PID
