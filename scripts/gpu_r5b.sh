#!/bin/bash
# round 5, second visit (build tools/libvstab_hip_cxx64.so first: scripts/build_variant_lib.sh cxx64 -DVSTAB_NO_ASM_KLOOP_64): the 64 x 128 tile's assembly K loop (bit check against the build that keeps hipcc's loop for that tile, then the
# interleaved A/B at the two one-sample shapes), the one-call clip step (tests + bench_stream), the changed tests.
set -u
tag=${1:-r05b}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kloop.py tests/test_gpu_parity.py tests/test_gpu_skinny.py tests/test_gpu_clip.py -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 15 $o/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for s in "1 256 256" "1 384 512"; do
  python3 scripts/kloop_bitcheck.py $s 2>/dev/null | grep "^out.predict\|^out.flow" > $o/bit_asm.txt
  VSTAB_LIB=tools/libvstab_hip_cxx64.so python3 scripts/kloop_bitcheck.py $s 2>/dev/null | grep "^out.predict\|^out.flow" > $o/bit_cxx.txt
  if cmp -s $o/bit_asm.txt $o/bit_cxx.txt && [ -s $o/bit_asm.txt ]; then echo "bitcheck $s: identical ($(wc -l < $o/bit_asm.txt) tensors)"; else echo "bitcheck $s: DIFFERENT"; diff $o/bit_asm.txt $o/bit_cxx.txt | head; exit 1; fi
done
for s in "--batch 1 --height 256 --width 256" "--batch 1 --height 384 --width 512"; do
  echo "== $s: A = 64x128 tile on hipcc's loop, B = assembly loop"
  bash scripts/ab_bench.sh "VSTAB_LIB=tools/libvstab_hip_cxx64.so" "VSTAB_X=1" 3 $s --steps 400 --warmup 50 --no-secondary --no-flow-err > $o/ab_${tag}_$(echo $s | tr -d ' -' | cut -c1-24).txt 2>&1 || { tail $o/ab_${tag}_*.txt; exit 1; }
  grep round $o/ab_${tag}_$(echo $s | tr -d ' -' | cut -c1-24).txt
done
for c in 1 8; do
  timeout -k 10 300 python3 bench_stream.py --clips $c > $o/stream${c}_$tag.json 2> $o/stream${c}_$tag.err || { tail -5 $o/stream${c}_$tag.err; exit 1; }
  cut -c1-420 $o/stream${c}_$tag.json
done
