#!/bin/bash
# rocprofv3 passes over the bench command: kernel-trace stats, then PMC passes (each in its own run).
# usage: scripts/gpu_profile.sh <tag>
set -u
tag=${1:-run}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
# EXTRA="--batch 1 --height 384 --width 512" profiles another shape
EXTRA=${EXTRA:-}
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $EXTRA"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $BENCH > $out/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/stats.log; exit 1; }
find $out/stats -name '*kernel_stats*.csv' | head -1 | xargs -r head -30
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout -k 10 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-events $EXTRA > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
done
# keep only the CSVs small enough to merge back
find $out -name '*.csv' -size +8M -delete
du -sh $out
