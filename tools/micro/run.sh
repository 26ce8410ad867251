#!/bin/bash
# Development aid: builds and runs the two micro-benchmarks DESIGN.md §4.1/§6 quote (on the GPU box: gpurun -- 'bash tools/micro/run.sh')
set -e
cd "$(dirname "$0")"
out=${GRAFT_REPO_ROOT:-../..}/gpurun_out
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/valu_rate valu_rate.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_lines gather_lines.hip
timeout -k 10 120 /tmp/valu_rate | tee "$out/micro_valu_rate.txt"
for a in "2048 36" "2048 64" "32768 36"; do timeout -k 10 60 /tmp/gather_lines $a; done | tee "$out/micro_gather_lines.txt"
