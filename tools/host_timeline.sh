#!/bin/bash
# timeline (kernels and copies) of the LAST host-pointer call of tools/host_trace.py:
#   tools/host_timeline.sh <verify|x25519|sign> [log2n]   -> gpurun_out/host_tl_<op>.txt
OP=${1:-verify}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/host_tl_$OP
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -- python3 $REPO/tools/host_trace.py $OP $2 > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - $OUT <<'PY' | tee $OUT.txt
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + '/tr/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('ed::', '').replace('void ', ''), 'q' + r['Queue_Id']))
for f in glob.glob(d + '/tr/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'] + ' ' + r.get('Bytes', '?'), 'copy'))
ev.sort()
# the last call = everything after the longest idle gap in the last part of the trace
gaps = [(ev[i + 1][0] - max(e[1] for e in ev[max(0, i - 50):i + 1]), i) for i in range(len(ev) // 2, len(ev) - 1)]
cut = max(gaps)[1] + 1
last = ev[cut:]
t0 = last[0][0]
for s, e, name, q in last:
    if (e - s) < 20000 and 'k_' not in name and len(last) > 60: continue
    print('%-44s %-5s start %8.3f  end %8.3f  dur %7.3f ms' % (name[:44], q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
