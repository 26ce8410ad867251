#!/bin/bash
# usage: RTMI_COMMIT=<sha> [RTMI_PMC_CONFIG='{json}'] tools/pmc_run.sh <tag> <bench args...>   (run on the GPU box through gpurun)
# Separate rocprofv3 --pmc passes (no tracing domains combined with counters), CSV output under gpurun_out/.
# The profiled frame runs on ONE internal stream (RTMI_STREAMS=1): every launch has the GPU to itself, the launch set bench.py's
# roofline object times.
export RTMI_STREAMS=1
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-counters --no-d2h-leg "$@" > $out/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $out/p$i.log)"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $out ${RTMI_PMC_CONFIG:+"$RTMI_PMC_CONFIG"}
