"""Static checks of the gfx950 ISA that hipcc generates for the stepping kernels (cross-compiled here, no GPU needed):
the figures DESIGN.md and bench.py quote - float64 operations per RK4 substep, no IEEE division on the ordinary path, no
scratch, two waves per SIMD - are asserted against the compiler's output, so that they cannot drift from the code."""
import collections
import os
import re
import shutil
import subprocess

import pytest
from conftest import ROOT

from gym_sbr2_amd import build as B

K_STEP = "_Z6k_stepIffLi256ELb0ELi1ELi1EE"  # k_step<float, float, 256, false, 1, 1>: the kernel bench.py times (scheme 1)
K_STEP_SMALL = "_Z6k_stepIffLi64ELb0ELi1ELi1EE"    # the 64-thread-workgroup build used up to 49152 envs
K_STEP_RK4 = "_Z6k_stepIffLi256ELb0ELi0ELi1EE"   # cfg.scheme = 0: ten RK4 substeps per interval
K_STEP_2W = "_Z6k_stepIffLi256ELb0ELi1ELi2EE"  # the same above 65536 envs: parked call state, two waves per SIMD
K_ROLLOUT = "_Z9k_rolloutILb0ELi1ELi1EE"      # k_rollout<false, 1, 1>: scheme 1, register budget for one wave per SIMD
K_ROLLOUT_2W = "_Z9k_rolloutILb0ELi1ELi2EE"
K_ROLLOUT_RK4 = "_Z9k_rolloutILb0ELi0ELi2EE"
K_CYCLE = "_Z7k_cycleIffLi1ELi1EE"
K_CYCLE_RK4 = "_Z7k_cycleIffLi0ELi2EE"
K_RESET = "_Z7k_resetIfLb0ELi256EE"
K_RESET_CARRY = "_Z7k_resetIfLb1ELi256EE"
K_RESET_WIDE = "_Z7k_resetIfLb0ELi512EE"      # above one wave per SIMD: 512-thread workgroups (one 84 KiB table copy per CU, two waves per SIMD)
K_CYCLE_RESET = "_Z13k_cycle_resetIfLb0EE"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "sbr_amd.s"
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([B.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", str(out), B.SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_text(asm, symbol):
    m = re.search(r"^%s[^\n:]*:[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(symbol), asm, re.S | re.M)
    assert m, symbol
    return m.group(1)


def instructions(text):
    out = []
    for line in text.split("\n"):
        line = line.split(";")[0].strip()
        if line and not line.endswith(":") and not line.startswith("."):
            out.append(line)
    return out


def inner_loops(text):
    """[(instruction list)] of the innermost loops (backward branches that contain no other backward branch)."""
    lines = []
    for raw in text.split("\n"):
        l = raw.split(";")[0].strip()
        if l and (l.endswith(":") or not l.startswith(".")):
            lines.append(l)
    label = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\S+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i:
            loops.append((label[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    return [[x for x in lines[a:b + 1] if not x.endswith(":")] for a, b in inner]


def all_loops(text):
    """[(instruction list)] of EVERY loop (one per backward branch, label to branch), nested or not."""
    lines = []
    for raw in text.split("\n"):
        l = raw.split(";")[0].strip()
        if l and (l.endswith(":") or not l.startswith(".")):
            lines.append(l)
    label = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\S+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i:
            out.append([x for x in lines[label[m.group(1)]:i + 1] if not x.endswith(":")])
    return out


def f64_mix(ins):
    c = collections.Counter(i.split()[0] for i in ins)
    return {"fma": c["v_fma_f64"] + c["v_fmac_f64_e32"], "mul": c["v_mul_f64"], "add": c["v_add_f64"], "rcp": c["v_rcp_f64_e32"],
            "div": c["v_div_fmas_f64"], "lane": c["v_readlane_b32"] + c["v_writelane_b32"], "scratch": sum(v for k, v in c.items() if k.startswith("scratch_"))}


def meta(asm, symbol, key):
    """A field of the kernel's entry in the code object's metadata (amdhsa.kernels).  The entries are YAML maps whose keys are
    sorted, so some precede `.name` and some follow it: take the whole entry (it starts at `  - .agpr_count:`)."""
    for blk in re.split(r"\n  - (?=\.agpr_count:)", asm[asm.index("amdhsa.kernels:"):]):
        if re.search(r"\.name:\s+%s\S*\n" % re.escape(symbol), blk):
            m = re.search(r"^\s+\.%s:\s+(\d+)" % key, "\n    " + blk, re.M)
            assert m, (symbol, key)
            return int(m.group(1))
    raise AssertionError((symbol, "not in the metadata"))


def test_rk4_substep_loops_have_the_quoted_instruction_mix(asm):
    import bench
    loops = inner_loops(kernel_text(asm, K_STEP_RK4))             # cfg.scheme = 0
    rk4 = [f64_mix(l) for l in loops if f64_mix(l)["rcp"] == 8 and f64_mix(l)["fma"] > 250]     # unrolled by two: 8 reciprocals
    # (the Butcher-5 step loops of scheme 1 hold 5 reciprocals: test_butcher5_step_loops below)
    assert len(rk4) >= 2
    per_substep = sorted({(m["fma"] * 2 + m["mul"] + m["add"] + m["rcp"]) // 2 for m in rk4})
    # bench.py's roofline.fp64_valu counts exactly these loops (FMA = 2 FLOP)
    assert per_substep[0] == bench.FP64_FLOP_PER_SUBSTEP["plain"] and per_substep[-1] == bench.FP64_FLOP_PER_SUBSTEP["dosing"], per_substep
    # the dosing loop in scaled-mass variables - at most 310 instructions per substep (measured 296.5;
    # the concentration form of rounds 2-3 had 343.5), the closed loop unchanged at 275
    per_substep_instr = sorted(len(l) / 2.0 for l in loops if f64_mix(l)["rcp"] == 8 and f64_mix(l)["fma"] > 250)
    assert per_substep_instr[-1] <= 310 and per_substep_instr[0] <= 276, per_substep_instr
    for l in loops:
        m = f64_mix(l)
        if m["rcp"] == 8 and m["fma"] > 250:
            assert m["div"] == 0 and m["lane"] == 0 and m["scratch"] == 0       # nothing but arithmetic in the hot loops
            assert len(l) <= 2 * (m["fma"] + m["mul"] + m["add"] + m["rcp"]) // 2 + 12   # <= 6 non-arithmetic instructions per substep


def test_butcher5_step_loops(asm):
    """cfg.scheme = 1 (round 5): the step loops of the adaptive Butcher-5 integrator (sbr_b5a) - one per form (closed reactor /
    carbon dosing in scaled-mass variables) and inlined copy.  A step is the loop from its head to the back edge that follows the
    next step's first stage (a shorter back edge, taken when no lane needs another step, skips that stage): six right-hand sides,
    <= 522 instructions without dosing and <= 582 with (measured 498 / 556; an RK4 substep has four and 272 / 297), nothing
    but arithmetic: no division, no lane operation, no scratch.  bench.py's FP64_FLOP_PER_B5_STEP are these
    loops' counts (FMA = 2)."""
    import bench
    for k in (K_STEP, K_STEP_2W, K_ROLLOUT, K_CYCLE):
        # (lane operations in a step loop are SGPR spills into VGPR lanes: none in k_step and k_cycle, the one-wave build of
        # k_rollout holds four per step since the library is built with -ffp-contract=off - under 1 % of the loop)
        steps = [l for l in all_loops(kernel_text(asm, k)) if f64_mix(l)["rcp"] == 6 and len(l) < 580 and f64_mix(l)["lane"] <= (4 if k == K_ROLLOUT else 0)]
        assert len(steps) >= (1 if k == K_CYCLE else 2), k
        flop = sorted({m["fma"] * 2 + m["mul"] + m["add"] + m["rcp"] for m in map(f64_mix, steps)})
        assert flop[0] == bench.FP64_FLOP_PER_B5_STEP["plain"], (k, flop)
        if k != K_CYCLE:
            assert flop[-1] == bench.FP64_FLOP_PER_B5_STEP["dosing"], (k, flop)
        for l in steps:
            m = f64_mix(l)
            arith = m["fma"] + m["mul"] + m["add"] + m["rcp"]
            # 290+169+12+6 / 328+179+24+6; the rest: moves, branches (the 256-register build of k_step: up to 30 more register copies)
            assert arith in (477, 537) and len(l) <= arith + (75 if k == K_STEP_2W else 45), (k, len(l), arith)
            assert m["div"] == 0 and m["scratch"] == 0, k
        # no AGPR traffic inside the step loops of the first (ordinary) control interval; the out-of-line copy for the second
        # interval of a phase-boundary call (3 calls per episode) may hold a few moves
        assert sum(1 for l in steps if not any("accvgpr" in i for i in l)) >= (1 if k == K_CYCLE else 2), k
    # the scheme-1 kernels carry no RK4 loop for the control intervals (k_step's only RK4 is gone with the idle phase; k_cycle
    # keeps the fill phase's), the scheme-0 kernels are what rounds 1-4 shipped
    rk4 = lambda k: [l for l in inner_loops(kernel_text(asm, k)) if f64_mix(l)["rcp"] == 8 and f64_mix(l)["fma"] > 250]   # noqa: E731
    assert len(rk4(K_STEP)) == 0 and len(rk4(K_ROLLOUT)) == 0 and len(rk4(K_STEP_RK4)) >= 2 and len(rk4(K_ROLLOUT_RK4)) >= 2
    assert len(instructions(kernel_text(asm, K_STEP))) < 0.8 * 11200      # 7 614 instructions; 11 142 with both schemes in one kernel


def test_fused_multiply_adds_are_the_ones_the_source_spells_out():
    """The library promises the same bits for an env whatever batch (i.e. whichever instantiation of a kernel) it is stepped in.
    With hipcc's default -ffp-contract=fast the backend fuses a * b + c where it sees fit, and it saw fit in three more places of
    k_step<..., WAVES = 2> than of k_step<..., WAVES = 1> (one ulp in the NO3-PID's integral; found by the GPU bit-identity test
    of the two).  The build flag is part of the contract."""
    assert "-ffp-contract=off" in B.FLAGS and "-fno-fast-math" in B.FLAGS


def test_k_step_has_no_scratch_no_division_on_the_ordinary_path(asm):
    ins = instructions(kernel_text(asm, K_STEP))
    m = f64_mix(ins)
    assert m["scratch"] == 0
    assert m["div"] <= 8, m["div"]          # <= 8 (terminal / fallback paths only); measured 4
    assert meta(asm, K_STEP, "private_segment_fixed_size") == 0
    # Round 5: the Butcher-5 steps keep five 9-vectors and twelve per-lane step constants live where RK4 kept three and four,
    # so k_step needs more than 256 registers (measured 292, the excess parked in AGPRs OUTSIDE the step loops) and one wave is
    # resident per SIMD.  At the bench's 65 536 envs (1 024 waves on 1 024 SIMDs) that is the occupancy anyway; launches of
    # 131 072 envs and more lose the overlap of two resident waves (DESIGN.md section 5 has the measured price).
    assert meta(asm, K_STEP, "vgpr_count") <= 336
    # ... which is why launches above 65 536 envs run k_step<..., WAVES = 2>: the call's state parked in LDS around the step
    # loops, 256 registers, no scratch, two waves (two 64 KiB workgroups) per SIMD (CU)
    two = f64_mix(instructions(kernel_text(asm, K_STEP_2W)))
    assert meta(asm, K_STEP_2W, "vgpr_count") <= 256 and meta(asm, K_STEP_2W, "agpr_count") == 0
    assert meta(asm, K_STEP_2W, "private_segment_fixed_size") == 0 and two["scratch"] == 0 and two["div"] <= 8
    assert meta(asm, K_STEP_2W, "group_segment_fixed_size") <= 80 * 1024
    assert meta(asm, K_STEP_RK4, "vgpr_count") <= 256 and meta(asm, K_STEP_RK4, "private_segment_fixed_size") == 0    # scheme 0: as shipped in round 4
    assert meta(asm, K_CYCLE_RK4, "vgpr_count") <= 256 and meta(asm, K_ROLLOUT_2W, "vgpr_count") <= 256
    assert meta(asm, K_ROLLOUT, "private_segment_fixed_size") == 0 and meta(asm, K_CYCLE, "private_segment_fixed_size") == 0
    # k_reset without a scratch segment (its 68 B were spill slots of 17 scalar registers: the start state read
    # again from the argument registers after the fill loop; held in VGPRs now)
    assert meta(asm, K_RESET_WIDE, "vgpr_count") <= 256
    for k in (K_RESET, K_RESET_CARRY, K_RESET_WIDE, K_CYCLE_RESET):
        assert meta(asm, k, "private_segment_fixed_size") == 0 and f64_mix(instructions(kernel_text(asm, k)))["scratch"] == 0, k
    small = f64_mix(instructions(kernel_text(asm, K_STEP_SMALL)))
    assert small["scratch"] == 0 and small["div"] <= 8 and meta(asm, K_STEP_SMALL, "private_segment_fixed_size") == 0


def test_k_step_leading_arguments_are_preloaded(asm):
    m = re.search(r"\.amdhsa_kernel %s\S*\n(.*?)\.end_amdhsa_kernel" % re.escape(K_STEP), asm, re.S)
    assert m
    n = re.search(r"\.amdhsa_user_sgpr_kernarg_preload_length\s+(\d+)", m.group(1))
    assert n and int(n.group(1)) >= 14      # x, ctrl, n, action, obs, state, reward arrive in SGPRs


def test_secondary_kernels_carry_no_ieee_divisions(asm):
    """the k_step recipe applied to the per-cycle kernel and the resets.  Wave-uniform quotients (phase
    schedule, 1/(n td), So_sat/1800) are taken on the host with the reference's own operations, per-lane ones go through
    ONE reciprocal (the 13 flow-weighted influent means, the five effluent particulates, the reset observation's blend)."""
    assert f64_mix(instructions(kernel_text(asm, K_CYCLE)))["div"] <= 6 and f64_mix(instructions(kernel_text(asm, K_CYCLE_RK4)))["div"] <= 6   # measured 0 (round 2: 26)
    for k in (K_RESET, K_RESET_CARRY):
        assert f64_mix(instructions(kernel_text(asm, k)))["div"] <= 4, k         # measured 0 (round 2: 19 / 20)
    assert f64_mix(instructions(kernel_text(asm, K_CYCLE_RESET)))["div"] <= 4     # measured 2 (round 2: 15)
    assert meta(asm, K_ROLLOUT, "vgpr_count") <= 320


def _lines_with_comments(text):
    out = []
    for raw in text.split("\n"):
        l = raw.split(";")[0].strip()
        if l and not l.startswith(".") and not l.endswith(":"):
            out.append(l)
    return out


VALU_PREFIX = ("v_",)       # everything the vector ALU executes; VMEM (global_/buffer_/scratch_), LDS (ds_), SALU (s_) are not


@pytest.mark.parametrize("kernel", [K_STEP, K_STEP_2W, K_STEP_SMALL, K_STEP_RK4, K_ROLLOUT, K_CYCLE])
def test_inline_asm_invariants(asm, kernel):
    """what makes the hand-written assembly of the stepping kernels safe is asserted, not left to convention.
    Inline assembly is invisible to the compiler's hazard recognizer, so a refactor can silently reintroduce:
      (a) a VALU instruction right behind a 16-byte store - on gfx940+ a VMEM store of more than 64 bits keeps reading its data
          registers for two more cycles (round 4: the next store's address arithmetic overwrote them; `s_nop 1` in st_out16);
      (b) an SGPR that a VALU instruction wrote (v_readlane) used as the address of an asm store (round 4, the unshipped
          paired-rows patch: a memory-access fault) - every asm store uses the VGPR-address form `..., off`;
      (c) a kernarg warm-up load (SBR_WARM_LINES) past the end of the kernel's argument segment."""
    text = kernel_text(asm, kernel)
    ins = _lines_with_comments(text)
    wide_sc1 = [i for i, l in enumerate(ins) if l.startswith("global_store_dwordx4") and "sc1" in l]
    if kernel in (K_STEP, K_STEP_2W, K_STEP_SMALL, K_STEP_RK4):
        assert len(wide_sc1) >= 9                      # the output rows leave as 16-byte write-through stores
    for i in wide_sc1:
        nxt = ins[i + 1]
        assert nxt.startswith("s_nop") or not nxt.startswith(VALU_PREFIX), (kernel, ins[i], nxt)         # (a)
        ops = [o.strip() for o in ins[i].split(None, 1)[1].split(",")]
        assert ops[0].startswith("v") and ops[2].split()[0] == "off", (kernel, ins[i])                     # (b)
    # every s_nop-guarded pair really is `store ; s_nop 1` (the asm statement was not split by an edit)
    assert all(ins[i + 1] == "s_nop 1" for i in wide_sc1), kernel
    # (c) the warm-up: a run of >= 20 consecutive s_load_dword into one register with 0x40-spaced literal offsets
    seg = meta(asm, kernel, "kernarg_segment_size")
    warm = []
    for l in ins:
        m = re.match(r"s_load_dword (s\d+), (s\[\d+:\d+\]), (0x[0-9a-f]+)$", l)
        warm.append((m.group(1), int(m.group(3), 16)) if m else None)
    runs, cur = [], []
    for w in warm:
        if w is not None and (not cur or w[0] == cur[-1][0]):
            cur.append(w)
        else:
            if len(cur) >= 20:
                runs.append(cur)
            cur = [w] if w is not None else []
    if len(cur) >= 20:
        runs.append(cur)
    if kernel in (K_STEP, K_STEP_2W, K_STEP_SMALL, K_STEP_RK4):
        assert len(runs) == 1, (kernel, len(runs))
        offs = [o for _, o in runs[0]]
        assert max(offs) + 4 <= seg, (max(offs), seg)                     # inside the segment ...
        assert (max(offs) // 64) == ((seg - 1) // 64)                      # ... and its last 64-byte line is touched
        assert {o // 64 for o in offs} >= set(range(1, (seg - 1) // 64 + 1))   # every line but the preloaded first one
    else:
        assert not runs
