# pmc_write_path.sh <outdir>: L2 -> memory write-path counters of the headline rollout, per dispatch, next to each dispatch's duration
# (do slow launches differ from fast ones in what the L2's write requests meet?).  Two counter passes + a summary.
OUT=$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
P2="TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/p1 -- python3 profiles/experiments/roll_driver.py --fuse 256 --launches 300 > $OUT/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/p2 -- python3 profiles/experiments/roll_driver.py --fuse 256 --launches 300 > $OUT/p2.log 2>&1
python3 - $OUT <<'P'
import csv, glob, sys, collections, statistics
out = sys.argv[1]
for p in ('p1', 'p2'):
    dur = {}
    for f in glob.glob(f'{out}/{p}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rollout_ws' in r['Kernel_Name']:
                dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    ctr = collections.defaultdict(dict)
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rollout_ws' in r['Kernel_Name']:
                ctr[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    ids = [i for i in dur if i in ctr]
    ids.sort(key=lambda i: dur[i])
    if not ids:
        print(p, 'no dispatches'); continue
    q = max(1, len(ids) // 5)
    print(f'{p}: {len(ids)} dispatches, duration us min {dur[ids[0]]:.1f} median {dur[ids[len(ids) // 2]]:.1f} max {dur[ids[-1]]:.1f}')
    for name in sorted(ctr[ids[0]]):
        fast = statistics.mean(ctr[i][name] for i in ids[:q]); slow = statistics.mean(ctr[i][name] for i in ids[-q:])
        print(f'   {name:40s} fastest fifth {fast:16.0f}   slowest fifth {slow:16.0f}   ratio {slow / max(fast, 1):6.2f}')
    print('   duration: fastest fifth %.1f us, slowest fifth %.1f us' % (statistics.mean(dur[i] for i in ids[:q]), statistics.mean(dur[i] for i in ids[-q:])))
P
