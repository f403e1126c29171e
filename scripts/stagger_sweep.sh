# VSTAB_STAGGER = bit << 8 | units (units x 4096 cycles)
for bit in 3 4 8; do for d in 2 5; do
  sg=$((bit * 256 + d))
  out=$(VSTAB_STAGGER=$sg python3 bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/tmp/sg.err) || { tail -3 /tmp/sg.err; exit 1; }
  echo "bit $bit delay $d: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")  $(grep -E '^conv2 |^conv3 |^deconv2 |^conv3_1 ' /tmp/sg.err | awk '{print $1, $2}' | tr '\n' ' ')"
done; done
out=$(python3 bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/tmp/sg.err); echo "none: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")  $(grep -E '^conv2 |^conv3 |^deconv2 |^conv3_1 ' /tmp/sg.err | awk '{print $1, $2}' | tr '\n' ' ')"
