"""Timeline of the LAST call of a leg in a rocprofv3 --kernel-trace run: python3 tools/show_trace.py <dir> [first-kernel-substring]"""
import csv, glob, sys
d = sys.argv[1]; first = sys.argv[2] if len(sys.argv) > 2 else 'dil_mu'
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']][-1]
t0 = int(rows[idx]['Start_Timestamp']); prev = None
for r in rows[idx:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev) / 1e3 if prev else 0
    print(f"{(st-t0)/1e3:9.1f} +{(en-st)/1e3:8.1f} us gap {gap:6.1f}  {r['Kernel_Name'][:64]} grid {r.get('Grid_Size_X','')}")
    prev = en
print('total', (prev - t0) / 1e3, 'us')
