"""steady-state timeline of a leg from a rocprofv3 kernel trace: python scratch/r6/timeline.py <leg_kernel_trace.csv> <anchor kernel substring> [periods]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'naive_conv' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
a, b = idx[-(per + 2)], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
print("period (anchor to anchor): %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3 / per))
for r in rows[a:b]:
    n = r['Kernel_Name'].replace('void mcrx::', '').replace('mcrx::', '')[:48]
    print("%9.1f us  +%8.1f  q%-3s %-48s grid %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], n, r['Grid_Size_X']))
