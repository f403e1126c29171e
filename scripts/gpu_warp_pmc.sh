#!/bin/bash
# PMC passes over tools/warp_bench (one counter set per run): usage gpu_warp_pmc.sh <tag> <warp_bench args...>
set -u
tag=${1:-run}; shift || true
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/warppmc_$tag; rm -rf $out; mkdir -p $out
i=0
for pmc in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
find $out -name '*.csv' -size +8M -delete
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("vstab::", "").split("(")[0]
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in cnt.values() for c in d})
for k, d in cnt.items():
    print(k)
    for c in names:
        if c in d:
            print(f"    {c:<44} {sum(d[c]) / len(d[c]):16.1f}   (n={len(d[c])})")
PY
