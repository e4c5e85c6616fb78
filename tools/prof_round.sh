#!/bin/bash
# round-end evidence: bench line, rocprofv3 kernel stats, HBM traffic counters (separate passes)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
python bench.py --steps 10 --warmup 2 > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err
cat gpurun_out/r01/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01/stats.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01/write.log 2>&1
echo write done
