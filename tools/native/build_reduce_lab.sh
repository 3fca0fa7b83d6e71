#!/bin/bash
# usage: build_reduce_lab.sh [extra -D flags]   ->  tools/native/reduce_lab
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc -I../../include "$@" reduce_lab.hip ../../semanticlens_amd/csrc/runtime.hip -o reduce_lab 2>&1 | grep -v "argument unused" || true
ls -la reduce_lab
