#!/bin/bash
# Package power and shader clock beside each ingredient of k_synth7's loop (tools/power_mix.hip).
# usage: tools/power_mix.sh > gpurun_out/power_mix.txt
hipcc -O3 --offload-arch=gfx950 tools/power_mix.hip -o /tmp/pmix || exit 1
for m in ${MODES:-idle pk_fma pk_add sqrt lds_w64 lds_r64 lds_r128 exchange store mix mixstore}; do
  timeout -k 5 30 /tmp/pmix $m 7 > /tmp/pmix_$m.txt 2>&1 &
  pid=$!
  sleep 3
  p1=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ')
  sleep 1
  p2=$(rocm-smi --showpower 2>/dev/null | grep -E "Package Power" | sed 's/.*: //' | tr '\n' ' ')
  wait $pid || { echo "$m failed"; cat /tmp/pmix_$m.txt; exit 1; }
  echo "$(cat /tmp/pmix_$m.txt) | smi: $p1| $p2"
done
