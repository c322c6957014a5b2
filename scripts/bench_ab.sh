mkdir -p gpurun_out/r4q
for rep in 1 2 3; do
  for v in base tree; do
    if [ $v = base ]; then export SBR_AMD_LIB=build/libsbr_amd_base.so; else unset SBR_AMD_LIB; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v default  value %.4e  ms/step %.5f  avg_launch_us %.2f'%(d['value'], d['ms_per_step'], r['avg_launch_us']))"
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v driver   value %.4e  ms/step %.5f  avg_launch_us %.2f'%(d['value'], d['ms_per_step'], r['avg_launch_us']))"
  done
done
