#!/bin/bash
# Round-3 measurement artefacts under gpurun_out/ (copy the summaries into profiles/).  One counter group per --pmc run,
# never combined with a trace domain other than the kernel trace (MI355X_MICROARCH.md).
#   K1 is traced per REGIME, in separate processes, so that bytes / AverageNs of a CSV row is one regime's number:
#     bench.py --quick            -> in-pipeline launches only (warm-up + timed job + self-check)
#     tools/k1_cold_target.py     -> cold launches only
#   each also run WITHOUT the tool: the difference of the HIP-event averages is the tool's per-dispatch overhead.
set -u
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
trace() {  # name command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- "$@" > $O/${R}_${name}.stdout 2> $O/prof_trace.err
  python tools/prof_summarize.py trace $O/prof_trace $O/${R}_${name}_kernel_stats.csv
  rm -rf $O/prof_trace
}
python bench.py --steps 20 --warmup 3 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err
python bench.py --steps 20 --warmup 3 --quick > $O/${R}_bench_line_quick.json 2>> $O/${R}_bench_line.err
trace k1_inpipeline python bench.py --steps 20 --warmup 3 --quick
mv $O/${R}_k1_inpipeline.stdout $O/${R}_bench_line_quick_profiled.json
# one stream (no encoder beside K1): the tool's host-side launch cost cannot shift the overlap of two streams
python bench.py --steps 20 --warmup 3 --quick --no-overlap > $O/${R}_bench_line_quick1s.json 2>> $O/${R}_bench_line.err
trace k1_inpipeline1s python bench.py --steps 20 --warmup 3 --quick --no-overlap
mv $O/${R}_k1_inpipeline1s.stdout $O/${R}_bench_line_quick1s_profiled.json
python tools/k1_cold_target.py > $O/${R}_k1_cold_events.json 2>> $O/${R}_bench_line.err
trace k1_cold python tools/k1_cold_target.py
mv $O/${R}_k1_cold.stdout $O/${R}_k1_cold_events_profiled.json
trace enc python tools/encoder_prof.py
rm -f $O/${R}_enc.stdout
(cd tools/native && ./build_reduce_lab.sh > /dev/null 2>&1)
if [ -x tools/native/reduce_lab ]; then
  tools/native/reduce_lab pipe > $O/${R}_reduce_pipe_lab.log 2>&1        # policy sweep (no tool)
  tools/native/reduce_lab pipe1 > $O/${R}_reduce_pipe1_lab.log 2>&1      # shipped policy only (no tool)
  trace reduce_pipe1 tools/native/reduce_lab pipe1                       # the same under the kernel trace
  mv $O/${R}_reduce_pipe1.stdout $O/${R}_reduce_pipe1_lab_profiled.log
fi
python tools/roofline_check.py $O $R > $O/${R}_roofline_check.txt 2>&1
pmc() {  # name target counters...
  local name=$1 target=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $O/prof_pmc -- python $target > $O/prof_pmc.log 2>&1
  python tools/prof_summarize.py pmc $O/prof_pmc $O/${R}_pmc_${name}.csv
  rm -rf $O/prof_pmc
}
pmc fetch tools/pmc_target.py FETCH_SIZE
pmc write tools/pmc_target.py WRITE_SIZE
pmc gemm_a tools/pmc_gemm.py GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
pmc gemm_b tools/pmc_gemm.py SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv $O/roofline_traffic.json
head -1 $O/${R}_pmc_gemm_a.csv > $O/${R}_pmc_gemm.csv
for x in a b; do tail -n +2 $O/${R}_pmc_gemm_$x.csv >> $O/${R}_pmc_gemm.csv; rm -f $O/${R}_pmc_gemm_$x.csv; done
echo done
