#!/bin/bash
# round 6 call z (the round's last): profile set of the head (one bench command per rocprofv3 summary + the two TCC counter passes), the whole GPU suite,
# smoke, every bench line, evaluate() with and without the IIR keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
PMC=1 bash tools/collect_profiles_r06.sh r06 2>&1 | tail -30 | cut -c1-200
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12 | cut -c1-400 | tee gpurun_out/r6z_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r6z_bench.json 2> gpurun_out/r6z_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r6z_bench.json | cut -c1-700
for C in apitrue cfg3 cfg4 cfg5; do timeout 600 python bench.py --config $C --no-side > gpurun_out/r6z_bench_$C.json 2>/dev/null; echo "$C rc=$?"; tail -1 gpurun_out/r6z_bench_$C.json | cut -c1-300; done
{ PASSES=5 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-200
  PASSES=7 timeout 400 python tools/exp_e2e.py 2>&1 | grep "evaluate()" | cut -c1-220; } | tee gpurun_out/r6z_e2e.log
