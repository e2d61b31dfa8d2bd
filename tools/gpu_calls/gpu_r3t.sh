#!/bin/bash
# round-3 evidence pass: full GPU test suite, the four bench configs, rocprof stats + PMC traffic (summaries only come back)
R=$PWD; O=$R/gpurun_out/r3t; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --config cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python bench.py --config cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc=$?"
PMC=1 bash tools/collect_profiles_r03.sh r03 2>&1 | tail -30
