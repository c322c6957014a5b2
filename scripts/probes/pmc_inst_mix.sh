# Dynamic instruction mix of k_step by PMC (run on the GPU box): per-type VALU counters in two passes over one episode pair
mkdir -p gpurun_out/mix; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/mix/avail.txt
for pass in A B C; do
  case $pass in
    A) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM";;
    B) C="SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MISC";;
    C) C="SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/mix/pmc_$pass -o run --output-format csv -- python3 scripts/probes/pmc_phase.py > gpurun_out/mix/pmc_$pass.log 2>&1
  python scripts/probes/pmc_phase_summary.py $(find gpurun_out/mix/pmc_$pass -name "*counter_collection.csv") > gpurun_out/mix/phase_$pass.txt 2>&1
  echo "== pass $pass"; cat gpurun_out/mix/phase_$pass.txt; rm -rf gpurun_out/mix/pmc_$pass
done
