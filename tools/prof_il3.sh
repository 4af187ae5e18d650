#!/bin/bash
# Run ON THE GPU BOX: PMC passes of the inner-light kernels alone (tools/exp_il3.py).  tools/prof_il3.sh <tag> <precision codes...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/il3_$tag
mkdir -p $out
export TF_TIMING_ONLY=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq -- python3 tools/exp_il3.py 7424837 "$@" > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $out/lds -- python3 tools/exp_il3.py 7424837 "$@" > $out/lds.log 2>&1 || { tail -5 $out/lds.log; exit 1; }
for k in inner_light3_kernel inner_light2_kernel; do for d in sq lds; do echo "== $k $d"; python3 tools/pmc_summary_one.py $out/$d $k; done; done
