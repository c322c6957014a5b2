"""Static analysis of the gfx950 ISA of one kernel: instruction mix per loop (innermost backward branches).
usage: python scripts/isa_loops.py file.s kernel_symbol_prefix"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^%s[^\n]*\n(.*?)\.Lfunc_end' % re.escape(sys.argv[2]), s, re.S | re.M)
lines = []
for l in m.group(1).split('\n'):
    l = l.split(';')[0].strip()
    if not l:
        continue
    if l.endswith(':') or not l.startswith('.'):
        lines.append(l)
lab = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
ins_all = [l.split()[0] for l in lines if not l.endswith(':')]
print('static instructions:', len(ins_all))
loops = []
for i, l in enumerate(lines):
    mm = re.match(r's_(?:cbranch_\w+|branch)\s+(\.LBB\S+)', l)
    if mm and mm.group(1) in lab and lab[mm.group(1)] < i:
        loops.append((lab[mm.group(1)], i))
# innermost loops only
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
for a, b in sorted(set(loops)):
    seg = [x.split()[0] for x in lines[a:b + 1] if not x.endswith(':')]
    c = collections.Counter(seg)
    f64 = sum(v for k, v in c.items() if '_f64' in k)
    tag = 'INNER' if (a, b) in inner else 'outer'
    print('%s loop lines %d-%d: %d instr | f64 %d (fma %d mul %d add %d rcp %d div_scale %d) | readlane %d writelane %d accvgpr %d '
          'v_mov %d cndmask %d s_nop %d s_other %d | vmem %d' % (
              tag, a, b, len(seg), f64, c['v_fma_f64'] + c['v_fmac_f64_e32'], c['v_mul_f64'], c['v_add_f64'], c['v_rcp_f64_e32'],
              c['v_div_scale_f64'], c['v_readlane_b32'], c['v_writelane_b32'], c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32'],
              c['v_mov_b32_e32'] + c['v_mov_b64_e32'], c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64'], c['s_nop'],
              sum(v for k, v in c.items() if k.startswith('s_') and k != 's_nop'),
              sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))))
