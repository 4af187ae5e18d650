#!/bin/bash
# Run ON THE GPU BOX: SQ / TCP counters of bvh_trace_kernel for library variants.  tools/prof_bvh.sh <lib.so>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=gpurun_out/prof_bvh_$tag
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/sq -- python3 tools/exp_bench_lib.py $lib 65536 > $out/sq.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/tc -- python3 tools/exp_bench_lib.py $lib 65536 > $out/tc.log 2>&1
  echo "== $tag"
  python3 tools/pmc_summary_one.py $out/sq bvh_trace
  python3 tools/pmc_summary_one.py $out/tc bvh_trace
done
