#!/bin/bash
# round 6 call m: (1) vector-memory path counters (TA / TD / TCP) of the shipped paired k_stft_wave and k_ssim - is the address unit what
# the 25 % of wave cycles in s_waitcnt vmcnt wait for?  (2) kernel trace of an evaluate() pass with the 36 IIR keys at the shipped
# defaults: GPU busy fraction, the gaps, the per-kernel split (where a further gain on that pass would have to come from)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R=$PWD
OUT=$R/gpurun_out/pmc_r6m; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr" "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  NO_CHECK=1 PAIRS=1024 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- python $R/tools/exp_stage.py > $OUT/p$i.log 2>&1 || { echo "pass $i ($SET) failed"; tail -3 $OUT/p$i.log; }
done
cd $R
{ echo "== clock"; python tools/pmc_clock.py $OUT/p1
  echo "== TA / TD / TCP counters, medians per launch (tools/exp_stage.py, 1024 pairs)"; python tools/pmc_summary.py $OUT k_stft k_ssim; } > gpurun_out/r6m_vmem_path.txt 2>&1
cat gpurun_out/r6m_vmem_path.txt
find $OUT -name "*.csv" -size +1M -delete
echo "== evaluate() with the 36 IIR keys: kernel trace"
cd /tmp
rm -rf $R/gpurun_out/r6m_prof
PASSES=3 IIR="butter,cheby,ellip,bessel:4000,8000,12000:2,4,8" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6m_prof -o s -- python $R/tools/exp_e2e.py > $R/gpurun_out/r6m_e2e.log 2>&1
cd $R; grep "evaluate()" gpurun_out/r6m_e2e.log | cut -c1-200
head -16 gpurun_out/r6m_prof/*kernel_stats.csv | cut -c1-200 | tee gpurun_out/r6m_kernel_stats_head.csv
T=$(find gpurun_out/r6m_prof -name "*kernel_trace.csv" | head -1)
WINDOW_MS=700 python tools/trace_gaps.py $T | tee gpurun_out/r6m_gaps_last_pass.txt
python - "$T" <<'EOF' | tee gpurun_out/r6m_last_pass_kernels.txt
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); t_end = rows[-1][1]; rows = [r for r in rows if r[0] >= t_end - 700e6]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in rows: agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]: print("%-62s %4d launches %9.2f ms" % (k, n, ms))
EOF
find gpurun_out/r6m_prof -name "*kernel_trace.csv" -delete
