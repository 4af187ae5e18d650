#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + the PMC passes of the default bench command.
#   tools/collect_profiles.sh <tag>     ->  gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...
# FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
set -e
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $CMD > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $CMD > $out/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq -- $CMD > $out/sq.log 2>&1
echo "sq done"
