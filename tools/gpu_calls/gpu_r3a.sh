#!/bin/bash
# round-3 GPU call A: parity of the packed epilogue, same-box A/B against the round-2 library, the sliced overlap experiment,
# the bench line, kernel stats incl. the side kernels
R=$PWD; O=$R/gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  SSR_DEV_LIB=tools/_build/libssrhip_r02.so NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed 's/^/r02: /'
  NO_CHECK=1 python tools/exp_stage.py 2>&1 | tail -1 | sed 's/^/new: /'
done | tee $O/ab_stage.log
python tools/exp_overlap.py 2>&1 | tee $O/overlap.log | tail -8
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2_side -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_cfg2_side.log 2>&1; echo "rocprof rc=$?"
find $O -name "*kernel_stats.csv" | head
