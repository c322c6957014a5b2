mkdir -p gpurun_out/r4r; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in base tree; do
  if [ $v = base ]; then export SBR_AMD_LIB=build/libsbr_amd_base.so; else unset SBR_AMD_LIB; fi
  python scripts/probes/two_groups_raw.py one --steps 384 --reps 5 2>&1 | grep -v amdgpu > gpurun_out/r4r/raw_$v.log; cat gpurun_out/r4r/raw_$v.log
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/r4r/trace_$v -o run --output-format csv -- python3 scripts/probes/two_groups_raw.py one --steps 384 --reps 3 > /dev/null 2>&1
  python scripts/probes/trace_overlap.py $(find gpurun_out/r4r/trace_$v -name "*kernel_trace.csv") > gpurun_out/r4r/gap_$v.log 2>&1; cat gpurun_out/r4r/gap_$v.log; rm -rf gpurun_out/r4r/trace_$v
done
