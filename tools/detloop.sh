# repeated fresh-process runs of the bit-identical-repeats test under different switches (which one breaks it, how often)
n=${1:-10}
for cfg in "CY_CONV_DIRECT=1" "CY_CONV_DIRECT=1 CY_HEADS_SIDE=0" "CY_CONV_DIRECT=0" "CY_CONV_DIRECT=1 CY_WGRAD_SIDE_STREAM=0"; do
  fails=0
  for i in $(seq $n); do
    r=$(env $cfg python -m pytest tests/test_gpu_r2.py -m gpu -x -q -k "deterministic_mode and bf16" 2>&1 | grep -E "^E  |passed|failed" | tr "\n" " " | cut -c1-160)
    case "$r" in *failed*) fails=$((fails+1)); echo "   $cfg run $i: $r";; esac
  done
  echo "$cfg: $fails failures of $n"
done
