#!/bin/bash
# Per-step kernel time table of the bench workload from rocprofv3 --kernel-trace --stats: two runs (A: few steps, B: more
# steps) whose difference removes the one-time autotuning launches.  Prints per-kernel ms/step; writes
# gpurun_out/<tag>_kernel_stats_{a,b}.csv and gpurun_out/<tag>_per_step.txt (copy to profiles/).
# usage: tools/rocprof_bench.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
extra=("$@")
run() {   # name steps
  local name=$1 steps=$2
  out=$root/gpurun_out/prof_${tag}_$name
  rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/bench.py --worker --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-extra "${extra[@]}" > $root/gpurun_out/prof_${tag}_$name.log 2>&1)
  cp "$(find $out -name '*kernel_stats.csv' | head -1)" $root/gpurun_out/${tag}_kernel_stats_$name.csv
  find $out -name "*.csv" -size +2M -delete
}
STEPS_A=4; STEPS_B=24
run a $STEPS_A
run b $STEPS_B
python - $root/gpurun_out/${tag}_kernel_stats_a.csv $root/gpurun_out/${tag}_kernel_stats_b.csv $((STEPS_B-STEPS_A)) <<'PY' | tee $root/gpurun_out/${tag}_per_step.txt
import csv, sys, re, collections
a = {r['Name']: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r['Name']: r for r in csv.DictReader(open(sys.argv[2]))}
n = float(sys.argv[3])
rows = []
for k, rb in b.items():
    ra = a.get(k)
    dt = float(rb['TotalDurationNs']) - (float(ra['TotalDurationNs']) if ra else 0.0)
    dc = float(rb['Calls']) - (float(ra['Calls']) if ra else 0.0)
    if dc > 0:
        rows.append((dt / n / 1e6, dc / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('per-step kernel time (difference of a %s-step and a %s-step run, one-time tuning launches cancel): %.3f ms' % ('4', '24', tot))
fam = collections.OrderedDict()
def family(k):
    for f, pat in (('conv fwd/dgrad (igemm_*, conv3x3_slab*, direct*)', 'igemm|conv3x3_slab|direct3x3|direct1x1|direct_s2dgrad'), ('wgrad', 'wgrad_dma|wgrad_kernel'), ('wgrad fold', 'wgrad_reduce'),
                   ('bn_act_fwd', 'bn_act_fwd'), ('bn_bwd_reduce', 'bn_bwd_reduce'), ('bn_bwd_apply', 'bn_bwd_apply'),
                   ('bn finalisers', 'bn_finalize|bn_bwd_finalize'), ('adam', 'adam_multi'), ('pack', 'pack_weights'),
                   ('pools/upsample/slices', 'maxpool|upsample|slice|f32_to_view|nchw'), ('head', 'decode|assign|pairs|dense|giou|finalize_kernel|bias_grad')):
        if re.search(pat, k):
            return f
    return 'other'
for ms, calls, k in rows:
    f = family(k)
    d = fam.setdefault(f, [0.0, 0.0])
    d[0] += ms; d[1] += calls
print('%-28s %9s %9s' % ('family', 'ms/step', 'launches'))
for f, (ms, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print('%-28s %9.3f %9.1f' % (f, ms, calls))
print()
for ms, calls, k in rows[:40]:
    print('%8.3f ms  %7.1f launches  %8.2f us avg  %s' % (ms, calls, 1e3 * ms / calls, k[:110]))
PY
tail -c 300 $root/gpurun_out/prof_${tag}_b.log
