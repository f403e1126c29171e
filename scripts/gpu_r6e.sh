#!/bin/bash
# round 6: XCD-contiguous input transforms at one sample per frame: interleaved A/B (old = -DVSTAB_NO_XCD_INPUT build)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06g}
one() { name=$1; lib=$2; shift 2; env $lib python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
st() { name=$1; lib=$2; shift 2; env $lib python3 bench_stream.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step_async'], d['value'])"; }
OLD=VSTAB_LIB=tools/libvstab_hip_noxcdin.so; NEW=VSTAB_X=0
for i in 1 2 3 4; do
  one b1_old $OLD --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
  one b1_new $NEW --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
  one cfg0_old $OLD --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --no-kernel-events
  one cfg0_new $NEW --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --no-kernel-events
  st stream1_old $OLD --clips 1
  st stream1_new $NEW --clips 1
done
one b1ev_old $OLD --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --event-every 4
one b1ev_new $NEW --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --event-every 4
for n in b1ev_old b1ev_new; do echo "== $n"; grep -A17 "^launch" $o/ab_${tag}_$n.err | cut -c1-100; done
