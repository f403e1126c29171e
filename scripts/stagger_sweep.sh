for sg in 0 2 4 8 16 512 1024 2048 1028 2056; do
  out=$(VSTAB_STAGGER=$sg python3 bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/tmp/sg.err) || { tail -3 /tmp/sg.err; exit 1; }
  echo "stagger $sg: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")  $(grep -E '^conv2 |^conv3 |^conv4 |^deconv3 ' /tmp/sg.err | awk '{printf \"%s %s  \", $1, $2}')"
done
