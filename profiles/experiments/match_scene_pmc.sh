# match_scene_pmc.sh <outdir>: instruction counts per wave-cycle of the 11v11 kernel in a quiet scene and in waiting scenes
OUT=$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for scene in quiet kickin aftergoal; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/$scene -- python3 profiles/experiments/match_scene_driver.py $scene > $OUT/$scene.log 2>&1 &&
  python3 - $OUT/$scene $scene <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0]); dur = []
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'match_rollout' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'match_rollout' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('%-10s' % sys.argv[2], ' '.join('%s %7.1f' % (c[9:], v / n / (4096 * 40)) for c, (v, n) in sorted(acc.items()) if c != 'SQ_WAVES'), ' launch us (x64/40): %.1f' % (sorted(dur)[len(dur) // 2] * 64 / 40))
P
done
