#!/bin/bash
# Round-4 measurement artefacts under gpurun_out/ (copy the summaries into profiles/).  One counter group per --pmc run, never
# combined with a trace domain other than the kernel trace (MI355X_MICROARCH.md).  K1 per regime as in round 3; new: K2 in the
# bench's configs[3] leg, the so400m tower per kernel, PMC traffic of colreduce2, the MFMA ceiling and the power probe.
set -u
R=${1:-r04}
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
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err
python bench.py --steps 20 --warmup 5 --quick > $O/${R}_bench_line_quick.json 2>> $O/${R}_bench_line.err
trace k1_inpipeline python bench.py --steps 20 --warmup 5 --quick
mv $O/${R}_k1_inpipeline.stdout $O/${R}_bench_line_quick_profiled.json
python bench.py --steps 20 --warmup 5 --quick --no-overlap > $O/${R}_bench_line_quick1s.json 2>> $O/${R}_bench_line.err
trace k1_inpipeline1s python bench.py --steps 20 --warmup 5 --quick --no-overlap
mv $O/${R}_k1_inpipeline1s.stdout $O/${R}_bench_line_quick1s_profiled.json
python tools/k1_cold_target.py > $O/${R}_k1_cold_events.json 2>> $O/${R}_bench_line.err
trace k1_cold python tools/k1_cold_target.py
mv $O/${R}_k1_cold.stdout $O/${R}_k1_cold_events_profiled.json
(cd tools/native && ./build_reduce_lab.sh > /dev/null 2>&1)
if [ -x tools/native/reduce_lab ]; then
  tools/native/reduce_lab pipe1 > $O/${R}_reduce_pipe1_lab.log 2>&1
  trace reduce_pipe1 tools/native/reduce_lab pipe1
  mv $O/${R}_reduce_pipe1.stdout $O/${R}_reduce_pipe1_lab_profiled.log
fi
python tools/roofline_check.py $O $R > $O/${R}_roofline_check.txt 2>&1
# K2 inside the bench's configs[3] collect leg (ViT-B/16 x 12 blocks, embed on the same stream): events, then the trace
python tools/k2_leg_probe.py 0 2> /dev/null | grep overlap > $O/${R}_k2_leg_events.txt
trace k2_leg python tools/k2_leg_probe.py 0
grep overlap $O/${R}_k2_leg.stdout > $O/${R}_k2_leg_events_profiled.txt; rm -f $O/${R}_k2_leg.stdout
python tools/k2_lab.py SL_COLREDUCE_IMPL=v2 SL_COLREDUCE_IMPL=vgpr 2>&1 | grep -v amdgpu.ids > $O/${R}_k2_lab_final.txt
# the so400m image tower per kernel at 64 and 256 images
for B in 64 256; do trace siglip_b$B python tools/siglip_prof.py $B; rm -f $O/${R}_siglip_b$B.stdout; done
python tools/siglip_bench.py 2>&1 | grep "image B\|text B\|difference" > $O/${R}_siglip_bench.txt
trace enc python tools/encoder_prof.py
rm -f $O/${R}_enc.stdout
python tools/encoder_bench.py > $O/${R}_encoder_bench.txt 2>&1
# counters
pmc fetch tools/pmc_target.py FETCH_SIZE
pmc write tools/pmc_target.py WRITE_SIZE
python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv $O/roofline_traffic.json
pmc k2_fetch tools/pmc_target_k2.py FETCH_SIZE
pmc k2_write tools/pmc_target_k2.py WRITE_SIZE
python tools/pmc_traffic_k2.py $O/${R}_pmc_k2_fetch.csv $O/${R}_pmc_k2_write.csv > $O/${R}_pmc_k2_traffic.txt 2>&1
pmc gemm_a tools/pmc_gemm.py GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES
head -1 $O/${R}_pmc_gemm_a.csv > $O/${R}_pmc_gemm.csv; tail -n +2 $O/${R}_pmc_gemm_a.csv >> $O/${R}_pmc_gemm.csv; rm -f $O/${R}_pmc_gemm_a.csv
# the matrix pipe with nothing to feed it, and clocks / power under the GEMM and the read stream
(cd tools/native && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_ceiling.hip -o mfma_ceiling > /dev/null 2>&1)
tools/native/mfma_ceiling 5 > $O/${R}_mfma_ceiling_final.txt 2>&1
bash tools/power_probe.sh > $O/${R}_power_probe.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/${R}_gpu_tests.txt
echo done
