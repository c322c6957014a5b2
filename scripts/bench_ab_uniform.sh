#!/bin/bash
# Same-box A/B (build/libsbr_amd_base.so against the in-tree library) of the secondary lines: the uniform policy, the fused rollout, the
# per-cycle kernel.  usage: bash scripts/bench_ab_uniform.sh
for rep in 1 2; do
  for v in base tree; do
    if [ $v = base ]; then export SBR_AMD_LIB=build/libsbr_amd_base.so; else unset SBR_AMD_LIB; fi
    for w in "--policy uniform" "--workload config5" "--workload cycle"; do
      python bench.py --no-cpu-baseline --no-large-leg $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w  value %.4e  ms/step %.5f  avg_launch_us %.2f'%(d['value'], d['ms_per_step'], r['avg_launch_us']))"
    done
  done
done
