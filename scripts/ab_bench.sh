#!/bin/bash
# interleaved A/B of two environments of bench.py in one box visit: usage ab_bench.sh "<env A>" "<env B>" [rounds] [bench args...]
A="$1"; B="$2"; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then e="$A"; else e="$B"; fi
    out=$(env $e python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/tmp/ab_$v.err) || { tail -5 /tmp/ab_$v.err; exit 1; }
    echo "$v [$e] round $i: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms  dom', d['roofline']['frac'], ' all', d['roofline']['all_mfma_launches']['frac'])")"
  done
done
grep -A17 "^launch" /tmp/ab_A.err | awk '{print "A  " $0}'; grep -A17 "^launch" /tmp/ab_B.err | awk '{print "B  " $0}'
