#!/bin/bash
# Package power and shader clock beside the interpolating-synthesis loop (tools/interp_mix.hip) and,
# on the same box, beside the k_synth7 mix it would replace (tools/power_mix.hip mixstore).
# usage: tools/interp_mix.sh > gpurun_out/interp_mix.txt
hipcc -O3 --offload-arch=gfx950 tools/interp_mix.hip -o /tmp/imix || exit 1
hipcc -O3 --offload-arch=gfx950 tools/power_mix.hip -o /tmp/pmix || exit 1
for m in ${MODES:-idle pmix:mixstore pmix:mix pmix:store4 fir8 fir8c fir6 fir8ns fir6ns st32 st16}; do
  bin=/tmp/imix
  case $m in pmix:*) bin=/tmp/pmix; m=${m#pmix:};; esac
  timeout -k 5 30 $bin $m 7 > /tmp/imix_$m.txt 2>&1 &
  pid=$!
  sleep 3
  p1=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ')
  sleep 1
  p2=$(rocm-smi --showpower 2>/dev/null | grep -E "Package Power" | sed 's/.*: //' | tr '\n' ' ')
  wait $pid || { echo "$m failed"; cat /tmp/imix_$m.txt; exit 1; }
  echo "$(cat /tmp/imix_$m.txt) | smi: $p1| $p2"
done
