#!/bin/bash
# Developer tool (GPU box): PMC passes over tools/gemm_one.py for a few shapes. Counters in their own runs.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
for shape in "2016 1024 256" "2016 256 1024" "2016 256 256" "16064 512 2048"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1_$tag -- python3 $R/tools/gemm_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2_$tag -- python3 $R/tools/gemm_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p3_$tag -- python3 $R/tools/gemm_one.py $shape 10 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -- python3 $R/tools/gemm_one.py $shape 10 > /dev/null 2>&1
done
find $OUT -name "*.csv" | wc -l
