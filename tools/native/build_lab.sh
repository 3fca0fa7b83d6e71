#!/bin/bash
# Builds tools/native/gemm3_lab (+ _probe with per-workgroup cycle stamps) and prints the 8-phase kernels' register use.
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS -DSL_GEMM_CLOCKPROBE gemm3_lab.hip -o gemm3_lab_probe -save-temps=obj 2>&1 | grep -v "argument unused" || true
grep "8phase.*\(num_vgpr\|private_seg\)" gemm3_lab-hip-amdgcn-amd-amdhsa-gfx950.s || true
echo "saddr-form DMA instructions: $(grep -c 'global_load_lds_dwordx4 v[0-9]*, s\[' gemm3_lab-hip-amdgcn-amd-amdhsa-gfx950.s)"
cp gemm3_lab-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/gemm3_lab.s
rm -f gemm3_lab-host* gemm3_lab-hip* gemm3_lab.hip-hip*
