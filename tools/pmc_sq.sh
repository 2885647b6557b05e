#!/bin/bash
# tools/pmc_sq.sh <tag> [bench args]: instruction-mix / issue counters of the stage kernels, one --pmc pass per group
# (run on the GPU box through gpurun).  Output: gpurun_out/sq_<tag>/summary.txt
TAG=$1; shift
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --tend-iters 3 --placement-tries 1 "$@" > /dev/null 2> $OUT/err_$i.txt || echo "pass $i failed" >> $OUT/errors.txt
  echo "pass $i done"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "k_stage" $OUT/summary.txt | cut -c1-190
