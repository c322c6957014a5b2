set -e -o pipefail
out=gpurun_out/r01; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf $out/pmc_fetch $out/pmc_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o run --output-format csv -- python3 scripts/pmc_workload.py > $out/pmc_fetch.json 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o run --output-format csv -- python3 scripts/pmc_workload.py > $out/pmc_write.json 2> $out/pmc_write.err
echo done
