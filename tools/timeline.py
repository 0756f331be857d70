#!/usr/bin/env python3
"""print the kernel timeline of the LAST verify pass in a rocprofv3 --kernel-trace csv"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('ed::k_verify')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_verify_prepare' in r['Kernel_Name']]
last = rows[starts[-1]:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-34s start %8.3f  end %8.3f  dur %7.3f ms  queue %s' % (r['Kernel_Name'].split('(')[0], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r['Queue_Id']))
