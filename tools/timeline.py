"""Kernel timeline of the LAST call of a scheme leg from a rocprofv3 --kernel-trace CSV (tools/profile_r04.sh: tl_<leg>/trace/.../*_kernel_trace.csv):
`python3 tools/timeline.py <kernel_trace.csv> <a kernel that opens a call | closes:<the kernel that closes one>> [its grid size]` prints start (us from the call's first kernel), duration, the
gap to the previous kernel, name and grid - the format of profiles/r0N_*_sign_timeline.txt.  The call ends at the first kernel that is not the engine's
(torch's element-wise kernels of the measuring script) or with the trace."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first, grid = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
if first.startswith('closes:'):                     # a call without a kernel of its own at its head: everything between the last two runs of the kernel that CLOSES a call
    ends = [i for i, r in enumerate(rows) if name(r).startswith(first[7:])]
    call = [r for r in rows[ends[-2] + 1:ends[-1] + 1] if not name(r).startswith('at::native')]
else:
    starts = [i for i, r in enumerate(rows) if name(r).startswith(first) and (grid is None or r['Grid_Size_X'] == grid)]
    b = starts[-1]
    call = []
    for r in rows[b:]:
        if name(r).startswith('at::native'):
            break
        call.append(r)
t0 = int(call[0]['Start_Timestamp'])
prev_end = t0
for r in call:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} + {(e - s) / 1e3:7.1f} us  gap {max(0, s - prev_end) / 1e3:6.1f}  {name(r)[:64]}  grid {r['Grid_Size_X']}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
