#!/bin/bash
# full GPU suite + smoke + bit check of the assembly K loops against the C++-loop build, one box visit
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03k}
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/pytest_$T.log 2>&1; rc=$?
tail -n 8 gpurun_out/pytest_$T.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1 || { tail gpurun_out/smoke_$T.log; exit 1; }
tail -n 2 gpurun_out/smoke_$T.log
if [ -f tools/libvstab_hip_cxxloop.so ]; then
  timeout -k 10 200 python3 scripts/kloop_bitcheck.py > gpurun_out/bitcheck_asm_$T.txt 2>/dev/null || exit 1
  VSTAB_LIB=tools/libvstab_hip_cxxloop.so timeout -k 10 200 python3 scripts/kloop_bitcheck.py > gpurun_out/bitcheck_cxx_$T.txt 2>/dev/null || exit 1
  if cmp -s gpurun_out/bitcheck_asm_$T.txt gpurun_out/bitcheck_cxx_$T.txt; then echo "bit check: assembly and C++ K loops give identical outputs"; cat gpurun_out/bitcheck_asm_$T.txt;
  else echo "bit check: DIFFERENT"; diff gpurun_out/bitcheck_asm_$T.txt gpurun_out/bitcheck_cxx_$T.txt; exit 1; fi
fi
