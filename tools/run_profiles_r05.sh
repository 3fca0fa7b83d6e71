#!/bin/bash
# Round-5 measurement artefacts under gpurun_out/ (copy the summaries into profiles/).  One counter group per --pmc run, never
# combined with a trace domain other than the kernel trace (MI355X_MICROARCH.md).  Stages can be selected: run_profiles_r05.sh r05 bench trace pmc
set -u
R=${1:-r05}; shift || true
STAGES=${*:-bench trace pmc gemm misc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
trace() {  # name command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- "$@" > $O/${R}_${name}.stdout 2> $O/prof_trace.err
  python tools/prof_summarize.py trace $O/prof_trace $O/${R}_${name}_kernel_stats.csv
  rm -rf $O/prof_trace
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
  python tools/k1_cold_target.py > $O/${R}_k1_cold_events.json 2>> $O/${R}_bench_line.err
  trace k1_cold python tools/k1_cold_target.py
  mv $O/${R}_k1_cold.stdout $O/${R}_k1_cold_events_profiled.json
  python tools/roofline_check.py $O $R > $O/${R}_roofline_check.txt 2>&1
  # K2 inside the bench's configs[3] collect leg (ViT-B/16 x 12 blocks, embed on the same stream), grouped and layer by layer
  python tools/k2_leg_probe.py 0 2> /dev/null | grep overlap > $O/${R}_k2_leg_events.txt
  SEMANTICLENS_AMD_GROUP_LAYERS=0 python tools/k2_leg_probe.py 0 2> /dev/null | grep overlap | sed 's/^/layer by layer (SEMANTICLENS_AMD_GROUP_LAYERS=0): /' >> $O/${R}_k2_leg_events.txt
  trace k2_leg python tools/k2_leg_probe.py 0
  grep overlap $O/${R}_k2_leg.stdout > $O/${R}_k2_leg_events_profiled.txt; rm -f $O/${R}_k2_leg.stdout
  # configs[4] collect leg (ConvNeXt-L stage outputs, channels_last)
  trace cfg4_leg python tools/k1_cfg4_probe.py 0
  grep overlap $O/${R}_cfg4_leg.stdout > $O/${R}_cfg4_leg_events_profiled.txt; rm -f $O/${R}_cfg4_leg.stdout
  ;;
pmc)
  pmc fetch tools/pmc_target.py FETCH_SIZE
  pmc write tools/pmc_target.py WRITE_SIZE
  python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv $O/roofline_traffic.json
  pmc k2_fetch tools/pmc_target_k2.py FETCH_SIZE
  pmc k2_write tools/pmc_target_k2.py WRITE_SIZE
  python tools/pmc_traffic_k2.py $O/${R}_pmc_k2_fetch.csv $O/${R}_pmc_k2_write.csv > $O/${R}_pmc_k2_traffic.txt 2>&1
  ;;
gemm)
  # the cosine GEMM at configs[3] shapes: matrix-pipe duty, then where a wave's cycles go (issue stalls, LDS, waits)
  pmc gemm tools/pmc_gemm.py GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
  pmc gemm_waits tools/pmc_gemm.py SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
  pmc gemm_insts tools/pmc_gemm.py SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
  ;;
misc)
  python tools/k3_bench.py > $O/${R}_k3_wave.txt 2>&1
  SL_K3_ATEN_IMPL=lane python tools/k3_bench.py --no-fuzz > $O/${R}_k3_lane.txt 2>&1
  python tools/scores_bench.py > $O/${R}_scores_bench.txt 2>&1
  python tools/siglip_bench.py 2>&1 | grep "image B\|text B\|difference" > $O/${R}_siglip_bench.txt
  python tools/encoder_bench.py > $O/${R}_encoder_bench.txt 2>&1
  ;;
tests)
  python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/${R}_gpu_tests.txt
  ;;
esac; done
echo done
