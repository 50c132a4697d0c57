# match_ablate.sh <outdir> lib.so ...: SQ_INSTS_VALU / SALU / LDS per wave-cycle and the rate of the 11v11 bench for each build
# (builds with one piece of the cycle compiled out -- the results are wrong, the instruction counts are what is read)
OUT=$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for lib in "$@"; do
  tag=$(basename $lib .so)
  S2D_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/$tag -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1 &&
  python3 - $OUT/$tag $tag <<'P'
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'match_rollout' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
line = open(sys.argv[1] + '.log').read().strip().splitlines()[-1]
try: rate = json.loads(line)['value'] / 1e9
except Exception: rate = float('nan')
print('%-16s' % sys.argv[2], ' '.join('%s %7.1f' % (c[9:], v / n / (4096 * 64)) for c, (v, n) in sorted(acc.items()) if c != 'SQ_WAVES'), ' (under the profiler: %.3f G)' % rate)
P
done
