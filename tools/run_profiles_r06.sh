#!/bin/bash
# Round-6 measurement artefacts under gpurun_out/ (copy the summaries into profiles/).  One counter group per --pmc run, never
# combined with a trace domain other than the kernel trace (MI355X_MICROARCH.md).  Stages: run_profiles_r06.sh r06 bench trace pmc gemm misc
set -u
R=${1:-r06}; shift || true
STAGES=${*:-bench trace pmc gemm misc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
trace() {  # name command...   (keeps the raw trace directory for `context` until the next trace)
  local name=$1; shift
  rm -rf $O/prof_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- "$@" > $O/${R}_${name}.stdout 2> $O/prof_trace.err
  python tools/prof_summarize.py trace $O/prof_trace $O/${R}_${name}_kernel_stats.csv
}
pmc() {  # name target counters...
  local name=$1 target=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $O/prof_pmc -- python $target > $O/prof_pmc.log 2>&1
  python tools/prof_summarize.py pmc $O/prof_pmc $O/${R}_pmc_${name}.csv
  rm -rf $O/prof_pmc
}
for S in $STAGES; do case $S in
bench)
  python bench.py --steps 20 --warmup 5 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err
  python bench.py --steps 20 --warmup 5 --quick > $O/${R}_bench_line_quick.json 2>> $O/${R}_bench_line.err
  ;;
trace)
  trace k1_inpipeline python bench.py --steps 20 --warmup 5 --quick
  mv $O/${R}_k1_inpipeline.stdout $O/${R}_bench_line_quick_profiled.json
  # which part of the quick run launches MIOpen's naive convolution kernel
  python tools/prof_summarize.py context $O/prof_trace $O/${R}_naive_conv_context.txt naive_conv
  trace quick_nocheck python bench.py --steps 20 --warmup 5 --quick --no-self-check
  python tools/prof_summarize.py context $O/prof_trace $O/${R}_naive_conv_context_no_self_check.txt naive_conv
  rm -f $O/${R}_quick_nocheck.stdout
  python tools/k1_cold_target.py > $O/${R}_k1_cold_events.json 2>> $O/${R}_bench_line.err
  trace k1_cold python tools/k1_cold_target.py
  mv $O/${R}_k1_cold.stdout $O/${R}_k1_cold_events_profiled.json
  python tools/roofline_check.py $O $R > $O/${R}_roofline_check.txt 2>&1
  # K6: a kernel-trace row for the cosine GEMM at configs[3] shapes, both modes (VERDICT r05 missing #5)
  trace k6 python tools/pmc_gemm.py
  rm -f $O/${R}_k6.stdout
  rm -rf $O/prof_trace
  ;;
pmc)
  pmc fetch tools/pmc_target.py FETCH_SIZE
  pmc write tools/pmc_target.py WRITE_SIZE
  python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv $O/roofline_traffic.json
  pmc half_fetch tools/pmc_target_half.py FETCH_SIZE
  pmc half_write tools/pmc_target_half.py WRITE_SIZE
  python tools/pmc_traffic_half.py $O/${R}_pmc_half_fetch.csv $O/${R}_pmc_half_write.csv > $O/${R}_pmc_half_traffic.txt 2>&1
  ;;
gemm)
  pmc gemm tools/pmc_gemm.py GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
  ;;
misc)
  python tools/k3_bench.py > $O/${R}_k3_wave.txt 2>&1
  python tools/scores_bench.py > $O/${R}_scores_bench.txt 2>&1
  python tools/siglip_bench.py 2>&1 | grep "image B\|text B\|difference" > $O/${R}_siglip_bench.txt
  python tools/encoder_bench.py > $O/${R}_encoder_bench.txt 2>&1
  ;;
tests)
  python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -25 > $O/${R}_gpu_tests.txt
  ;;
esac; done
echo done
