#!/bin/bash
# counters of the kernels of ONE single-item call (instructions per wave, cycles, hence the clock the chip ran at):
#   tools/single_call_pmc.sh <verify1|sign1|genpub1|x255191>   -> gpurun_out/single_pmc_<op>.txt
OP=${1:-verify1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/single_pmc_$OP
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python3 $REPO/tools/host_trace.py $OP > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 - $OUT <<'PY' | tee $OUT.txt
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(d + '/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        key = (int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0].replace('ed::', '').replace('void ', ''))
        rows[key][r['Counter_Name']] = rows[key].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
dur = {}
for f in glob.glob(d + '/pmc/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
last = sorted(rows)[-8:]
for k in last:
    c = rows[k]
    w = c.get('SQ_WAVES', 0) or 1
    us = dur.get(k[0], 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0)
    print('%-34s %7.1f us  waves %5d  valu/wave %8.0f  salu/wave %7.0f  lds/wave %6.0f  gui cycles %9.0f  -> %.2f GHz' % (
        k[1][:34], us, w, c.get('SQ_INSTS_VALU', 0) / w, c.get('SQ_INSTS_SALU', 0) / w, c.get('SQ_INSTS_LDS', 0) / w, gui, gui / us / 1e3 if us else 0))
PY
