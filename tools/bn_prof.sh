export TMPDIR=/tmp
root=$(pwd)
for shp in 19x512 38x256 76x128; do
for v in 4; do
  out=$root/gpurun_out/bnp_${shp}_$v; rm -rf $out
  (cd /tmp && BN_SHAPES=$shp CY_BN_MINPASS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/tools/bn_micro.py > /dev/null 2>&1)
  echo "== $shp minpass $v"
  python - "$(find $out -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'bn_' in n or 'elementwise' in n:
        print('  %-60s calls %5s avg %8.2f us min %8.2f' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done; done
