#!/bin/bash
# Profiling recipe of a round, to be run ON THE GPU BOX (inside one gpurun call):
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r04'        then here:  cp gpurun_out/r04/profiles/* profiles/ &&
#   python scripts/design_glance.py r04 --write   (DESIGN.md's table and README's headline are generated from these files)
# Writes everything under gpurun_out/<tag>/; scripts/pmc_summarise.py (run on the box, it needs no GPU) distils the passes
# into profiles/<tag>_*, and the bench lines are taken LAST, so that their roofline.traffic can quote the PMC summary of this
# very library (bench.py matches it by source hash).  Separate passes on purpose: --kernel-trace --stats for durations; one
# --pmc pass per counter (FETCH_SIZE, WRITE_SIZE), each only with --kernel-trace; the SQ pass last.  python3 is the program
# directly after `--`.
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
# which library the profile is of: the content hash gym_sbr2_amd.build keeps next to it (bench.py attaches the PMC figures of
# a committed profile only to a library with the same hash)
python3 -c "from gym_sbr2_amd import build as b; b.build_library(); print(open(b.HASH).read().strip())" > $out/library_source_hash.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_config2 -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-large-leg > $out/bench_config2_profiled.json 2> $out/rocprof_trace.err
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o run --output-format csv -- python3 scripts/pmc_workload.py > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o run --output-format csv -- python3 scripts/pmc_workload.py > $out/pmc_write.json 2> $out/pmc_write.err
echo "WRITE_SIZE pass done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace -d $out/pmc_sq -o run --output-format csv -- python3 scripts/pmc_workload.py > $out/pmc_sq.json 2> $out/pmc_sq.err || echo "SQ pass failed (optional)"
python3 scripts/pmc_summarise.py $tag
# the two measured constants of bench.py's roofline.serial_bound (stamp build of the same sources + the empty-launch period)
timeout -k 10 400 python3 scripts/serial_bound.py $tag > $out/serial_bound.json 2> $out/serial_bound.err || echo "serial_bound failed"
echo "serial bound: $(cut -c1-200 $out/serial_bound.json)"
# larger and smaller launches of the same workload (bench.py attaches them to the default line as roofline.larger_batches);
# taken BEFORE the default line so that it can quote them
for n in 32768 131072 262144; do
  timeout -k 10 300 python3 bench.py --envs-per-gpu $n --no-cpu-baseline > profiles/${tag}_bench_config2_n$n.json 2> $out/bench_n$n.err
  echo "bench n=$n: $(cut -c1-120 profiles/${tag}_bench_config2_n$n.json)"
done
# the two-waves-per-SIMD build of k_step under the kernel trace (whole episodes at 262144 envs per launch)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_n262144 -o run --output-format csv -- python3 bench.py --envs-per-gpu 262144 --no-cpu-baseline > $out/bench_n262144_profiled.json 2> $out/trace_n262144.err \
  && head -8 $(find $out/trace_n262144 -name "*kernel_stats.csv" | head -1) | cut -c1-400 > profiles/${tag}_bench_config2_n262144_kernel_stats.csv || echo "n262144 trace failed (optional)"
# ... and under the counters (separate passes again): traffic and issue activity with two waves on a SIMD
( export PMC_ENVS=262144; o=$out/pmc_n262144; mkdir -p $o
  timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $o/pmc_fetch -o run --output-format csv -- python3 scripts/pmc_workload.py > $o/f.json 2> $o/f.err \
  && timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $o/pmc_write -o run --output-format csv -- python3 scripts/pmc_workload.py > $o/w.json 2> $o/w.err \
  && timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace -d $o/pmc_sq -o run --output-format csv -- python3 scripts/pmc_workload.py > $o/s.json 2> $o/s.err \
  && python3 scripts/pmc_large_batch.py $o 262144 > profiles/${tag}_pmc_n262144.json ) || echo "PMC passes at 262144 envs failed (optional)"
# cfg.scheme = 0 (ten RK4 substeps per interval: what rounds 1-4 shipped) on the same box, same workload
timeout -k 10 300 python3 bench.py --scheme 0 --no-cpu-baseline > profiles/${tag}_bench_config2_scheme0.json 2> $out/bench_scheme0.err
timeout -k 10 300 python3 bench.py --scheme 0 --workload config5 --no-cpu-baseline > profiles/${tag}_bench_config5_scheme0.json 2>> $out/bench_scheme0.err
timeout -k 10 300 python3 bench.py --scheme 0 --workload cycle --no-cpu-baseline > profiles/${tag}_bench_cycle_scheme0.json 2>> $out/bench_scheme0.err
echo "bench scheme 0: $(cut -c1-120 profiles/${tag}_bench_config2_scheme0.json)"
for w in config2 config1 config5 cycle; do
  timeout -k 10 300 python3 bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err
  cp $out/bench_$w.json profiles/${tag}_bench_$w.json
  echo "bench $w: $(cut -c1-160 $out/bench_$w.json)"
done
# SURVEY.md 8d's synthetic inputs (U[0, 8] x U[0, 15] on all eight scenarios) and the reference's own action model, as SECONDARY lines
timeout -k 10 300 python3 bench.py --policy uniform --no-cpu-baseline --no-large-leg > profiles/${tag}_bench_config2_uniform.json 2> $out/bench_uniform.err
timeout -k 10 300 python3 bench.py --policy walk --no-cpu-baseline --no-large-leg > profiles/${tag}_bench_config2_walk.json 2> $out/bench_walk.err
echo "bench uniform: $(cut -c1-120 profiles/${tag}_bench_config2_uniform.json)"; echo "bench walk: $(cut -c1-120 profiles/${tag}_bench_config2_walk.json)"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_config2_driver_style.json 2> $out/bench_driver.err
cp $out/bench_config2_driver_style.json profiles/${tag}_bench_config2_driver_style.json
echo "bench driver-style: $(cut -c1-160 $out/bench_config2_driver_style.json)"
timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tail -2 > profiles/${tag}_pytest_gpu.log || true
echo "gpu tests: $(tail -1 profiles/${tag}_pytest_gpu.log)"
mkdir -p $out/profiles && cp profiles/${tag}_* $out/profiles/
# gpurun merges at most 64 MiB back: the raw per-dispatch tables of the passes have been distilled above
rm -rf $out/trace_config2 $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/trace_n262144 $out/pmc_n262144
du -sh $out | tail -1
echo "profile_round $tag finished"
