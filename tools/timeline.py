"""Dev helper: print the kernel timeline of the last frame from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'unpack_kernel' in r['Kernel_Name']]
last = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    n = r['Kernel_Name']; n = n[n.find('::') + 2:][:30]
    if 'rocclr' in r['Kernel_Name']: continue
    print('%-32s start %8.3f dur %8.3f ms' % (n, (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
