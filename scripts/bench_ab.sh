#!/bin/bash
# Same-box A/B of two builds of the library through bench.py: another build (SBR_AMD_LIB, default build/libsbr_amd_base.so) against
# the in-tree library, alternating, three rounds; whole episodes (the default run) and the driver's command.
# usage: bash scripts/bench_ab.sh <outdir> [other library]
out=${1:-gpurun_out/ab}
other=${2:-build/libsbr_amd_base.so}
mkdir -p $out
for rep in 1 2 3; do
  for v in other tree; do
    if [ $v = other ]; then export SBR_AMD_LIB=$other; else unset SBR_AMD_LIB; fi
    python bench.py --no-cpu-baseline --no-large-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v default  value %.4e  ms/step %.5f  avg_launch_us %.2f'%(d['value'], d['ms_per_step'], r['avg_launch_us']))"
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; b=r['larger_batches']['262144']; print('$v driver   value %.4e  ms/step %.5f  avg_launch_us %.2f | 262144 envs: ms/step %.5f frac_wall %.4f'%(d['value'], d['ms_per_step'], r['avg_launch_us'], b['ms_per_step'], b['frac_wall']))"
  done
done 2>&1 | tee $out/bench_ab.log
