#!/bin/bash
# usage: tools/prof_round.sh r04
# round-end evidence: bench line, rocprofv3 kernel stats, HBM traffic counters (separate passes,
# --pmc never combined with the hip/hsa/sys traces).  Then: python tools/traffic.py r03
R=${1:-r04}
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/$R
python bench.py --steps 10 --warmup 2 > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
cat gpurun_out/$R/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-ceilings --no-check --no-other-modes --no-other-configs > gpurun_out/$R/stats.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ceilings --no-check --no-other-modes --no-other-configs > gpurun_out/$R/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ceilings --no-check --no-other-modes --no-other-configs > gpurun_out/$R/write.log 2>&1
echo write done
# config 5 (streamed, per GPU): the line with its spot check, and the kernel stats of the same command
python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_config5.json 2> gpurun_out/$R/bench_config5.err
cat gpurun_out/$R/bench_config5.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats5 -- python3 bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline --no-ceilings --no-check --sustain 0 > gpurun_out/$R/stats5.log 2>&1
echo config5 done
