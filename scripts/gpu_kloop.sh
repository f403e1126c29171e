#!/bin/bash
# assembly K loop: parity of the conv stack first (under a timeout: a wrong barrier count would hang), then the interleaved A/B
# against the C++ loop build (tools/libvstab_hip_cxxloop.so, scripts/build_variant_lib.sh cxxloop -DVSTAB_NO_ASM_KLOOP)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x --timeout=300 > gpurun_out/pytest_kloop.log 2>&1; rc=$?
tail -n 12 gpurun_out/pytest_kloop.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash scripts/ab_bench.sh "VSTAB_LIB=tools/libvstab_hip_cxxloop.so" "VSTAB_X=1" ${1:-3} > gpurun_out/ab_kloop.txt 2>&1 || { tail gpurun_out/ab_kloop.txt; exit 1; }
cat gpurun_out/ab_kloop.txt
