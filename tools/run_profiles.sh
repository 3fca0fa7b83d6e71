#!/bin/bash
# Produces the round's measurement artefacts under gpurun_out/ (copy the summaries into profiles/):
#   kernel trace + stats of the bench command, PMC passes (one counter group per run, as MI355X_MICROARCH.md prescribes)
#   over tools/pmc_target.py (K1, HBM traffic) and tools/pmc_gemm.py (cosine GEMM: matrix-pipe duty, LDS, L2 requests).
set -u
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py --steps 20 --warmup 3 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-leg > $O/${R}_bench_line_profiled.json 2> $O/prof_trace.err
python tools/prof_summarize.py trace $O/prof_trace $O/${R}_bench_kernel_stats.csv
rm -rf $O/prof_trace
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
pmc gemm_c tools/pmc_gemm.py SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc gemm_d tools/pmc_gemm.py FETCH_SIZE
pmc gemm_e tools/pmc_gemm.py TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python tools/pmc_traffic.py $O/${R}_pmc_fetch.csv $O/${R}_pmc_write.csv $O/roofline_traffic.json
head -1 $O/${R}_pmc_gemm_a.csv > $O/${R}_pmc_gemm.csv
for x in a b c d e; do tail -n +2 $O/${R}_pmc_gemm_$x.csv >> $O/${R}_pmc_gemm.csv; rm -f $O/${R}_pmc_gemm_$x.csv; done
echo done
