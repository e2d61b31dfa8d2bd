#!/bin/bash
# round 4, call 5: bench lines after the multi entry + conv engine side figure
mkdir -p gpurun_out
timeout 900 python bench.py --config cfg3 --steps 5 --warmup 2 > gpurun_out/r04_bench_cfg3.json 2> gpurun_out/r04_bench_cfg3.err; tail -c 3000 gpurun_out/r04_bench_cfg3.json; tail -5 gpurun_out/r04_bench_cfg3.err
timeout 900 python bench.py --no-side > gpurun_out/r04_bench_cfg2_quick.json 2> gpurun_out/r04_bench_cfg2_quick.err; tail -c 1500 gpurun_out/r04_bench_cfg2_quick.json; tail -3 gpurun_out/r04_bench_cfg2_quick.err
