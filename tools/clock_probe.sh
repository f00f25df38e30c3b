#!/bin/bash
# Direct evidence for the data-dependent clock: the GPU's shader clock and socket power sampled (read-only
# rocm-smi queries, 5 per second) while (a) the K-split slab kernel loops on the 512 -> 1024 3x3 @19x19 launch with RANDOM operands,
# (b) the same loop with ZERO operands, (c) the train step runs.  usage: bash tools/clock_probe.sh > gpurun_out/clock_probe.txt
cd "$(dirname "$0")/.."
sample() {   # label, pid: sample until the process ends
  local label=$1 pid=$2 n=0 s_clk=0 s_pw=0 lo=99999 hi=0
  while kill -0 $pid 2>/dev/null; do
    out=$(rocm-smi --showclocks --showpower 2>/dev/null)
    clk=$(echo "$out" | grep -i "sclk clock level" | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/' | head -1)
    pw=$(echo "$out" | grep -i "Socket Graphics Package Power" | sed -E 's/.*: ([0-9.]+)$/\1/' | head -1)
    # (only samples taken while the GPU works: the interpreter's start-up and the tool's own set-up read 100-250 W)
    if [ -n "$clk" ] && [ -n "$pw" ] && [ "${pw%.*}" -ge 450 ]; then
      n=$((n+1)); s_clk=$((s_clk+clk)); s_pw=$(python3 -c "print($s_pw+$pw)")
      [ "$clk" -lt "$lo" ] && lo=$clk; [ "$clk" -gt "$hi" ] && hi=$clk
    fi
    sleep 0.2
  done
  [ $n -gt 0 ] && python3 -c "print('%-44s %3d busy samples: sclk mean %4.0f MHz (min %d, max %d), socket power mean %4.0f W' % ('$label', $n, $s_clk/$n, $lo, $hi, $s_pw/$n))"
}
SLAB_HINTS=13 python tools/slab_micro.py 150 200 0 > /tmp/cp_rand.txt 2>&1 &
sample "slab kernel 512->1024 3x3 @19, random operands" $!
grep "fwd" /tmp/cp_rand.txt | tail -1
ZERO=1 SLAB_HINTS=13 python tools/slab_micro.py 150 200 0 > /tmp/cp_zero.txt 2>&1 &
sample "slab kernel 512->1024 3x3 @19, zero operands" $!
grep "fwd" /tmp/cp_zero.txt | tail -1
python bench.py --steps 400 --warmup 5 --no-extra --no-cpu-baseline --no-roofline > /tmp/cp_step.txt 2>/dev/null &
sample "train step (configs[1])" $!
tail -1 /tmp/cp_step.txt | cut -c1-120
