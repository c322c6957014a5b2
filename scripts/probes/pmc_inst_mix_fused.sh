# Dynamic instruction mix of the fused kernels (k_rollout, k_cycle) by PMC, per wave and launch (run on the GPU box)
mkdir -p gpurun_out/mix; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for pass in A B; do
  case $pass in
    A) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM";;
    B) C="SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/mix/fused_$pass -o run --output-format csv -- python3 scripts/pmc_workload.py > gpurun_out/mix/fused_$pass.log 2>&1
  python3 - $(find gpurun_out/mix/fused_$pass -name "*counter_collection.csv") <<'PY'
import csv, sys
from collections import defaultdict
rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_rollout" in k or "k_cycle<" in k:
        d = rows[(k.split("(")[0][:40], int(r["Dispatch_Id"]))]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (k, i), d in rows.items():
    if k not in best or d.get("SQ_WAVES", 0) >= best[k].get("SQ_WAVES", 0): best[k] = d
for k, d in best.items():
    w = d.get("SQ_WAVES", 1.0)
    print(k, {n: round(v / w, 1) for n, v in sorted(d.items())})
PY
  rm -rf gpurun_out/mix/fused_$pass
done
