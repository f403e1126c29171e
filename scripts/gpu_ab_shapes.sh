#!/bin/bash
# interleaved A/B of two library builds over three shapes: usage gpu_ab_shapes.sh <libA or -> <libB or -> [rounds]
set -u
export TMPDIR=/tmp
A=$1; B=$2; R=${3:-2}
for s in "--batch 8 --height 512 --width 512 --steps 100 --warmup 10" "--batch 16 --height 1080 --width 1920 --steps 4 --warmup 1" "--batch 32 --height 720 --width 1280 --steps 4 --warmup 1"; do
  for i in $(seq 1 $R); do
    for v in "$A" "$B"; do
      if [ "$v" = "-" ]; then unset VSTAB_LIB; else export VSTAB_LIB=$v; fi
      out=$(python3 bench.py $s --no-cpu-baseline --no-secondary 2>/tmp/x.err) || { tail -3 /tmp/x.err; exit 1; }
      echo "$(echo $s | cut -c1-40) [$v] $(echo "$out" | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") $(grep '^conv1 ' /tmp/x.err | awk '{print $2, $5}')"
    done
  done
done
