mkdir -p gpurun_out/r4e; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in base w1; do
  for pass in A B; do
    if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; else C="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; fi
    SBR_AMD_LIB=build/libsbr_amd_$v.so timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/r4e/pmc_${v}_$pass -o run --output-format csv -- python3 scripts/probes/pmc_phase.py > gpurun_out/r4e/pmc_${v}_$pass.log 2>&1
    python scripts/probes/pmc_phase_summary.py $(find gpurun_out/r4e/pmc_${v}_$pass -name "*counter_collection.csv") > gpurun_out/r4e/phase_${v}_$pass.txt 2>&1
    echo "== $v $pass"; cat gpurun_out/r4e/phase_${v}_$pass.txt; rm -rf gpurun_out/r4e/pmc_${v}_$pass
  done
done
