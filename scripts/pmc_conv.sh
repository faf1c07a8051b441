# PMC passes for one conv kernel (counters in separate runs, kernel-trace only; see MI355X guide)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" | sort -u > $O/counters.txt
wc -l $O/counters.txt
for spec in "fwd 1 128 32 32" "wgrad 1 128 32 32"; do
  tag=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O -o ${tag}_a -- python3 $R/scripts/one_conv.py $spec 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O -o ${tag}_b -- python3 $R/scripts/one_conv.py $spec 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O -o ${tag}_c -- python3 $R/scripts/one_conv.py $spec 3 > /dev/null 2>&1
done
ls $O
