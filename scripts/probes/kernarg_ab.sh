for v in 0 1 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python scripts/gpu_ab.py base -- 16384 65536 2>&1 | head -1; done
