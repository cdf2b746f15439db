"""HBM bytes per launch of the headline kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_r02.sh (bytes = 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes:
the gfx950 correction of MI355X_MICROARCH.md).  The run contains launches at the BASELINE batch (65536 polynomials) and at the past-cache batch (2^20): the two
clusters are told apart by value.  usage: python3 tools/summarize_headline_pmc.py gpurun_out/prof_<tag>"""
import csv, glob, json, sys

def values(d, counter):
    out = []
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'ntt_kernel<8, 4, true, 2' in r['Kernel_Name'] and r['Counter_Name'] == counter:      # <.., false> plain form, <.., true> streaming form (round 4)
                out.append(float(r['Counter_Value']))
    return out

root = sys.argv[1]
f, w = values(root + '/pmc_fetch', 'FETCH_SIZE'), values(root + '/pmc_write', 'WRITE_SIZE')
res = {}
for name, lo_hi in (('polymul_ntt_n256_b65536', (0.0, 2.0)), ('polymul_ntt_n256_b1048576', (8.0, 32.0))):
    base_f, base_w = min(f), min(w)
    ff = [v for v in f if lo_hi[0] * base_f <= v < lo_hi[1] * base_f]
    ww = [v for v in w if lo_hi[0] * base_w <= v < lo_hi[1] * base_w]
    if ff and ww:
        res[name + '_bytes'] = (2 * sum(ff) / len(ff) + sum(ww) / len(ww)) * 1024
        res[name + '_launches'] = [len(ff), len(ww)]
res['algorithmic_bytes'] = {'b65536': 65536 * 3072, 'b1048576': 1048576 * 3072}
res['note'] = ('2*FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate rocprofv3 --pmc passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras`, '
               'kernel ntt_kernel<8,4,true,2,*> (signed small-modulus mode; the 2^20 batch runs the streaming I/O form), launches averaged per batch size')
print(json.dumps(res, indent=1))

# the same split for the kernel trace of the full bench command: average duration per batch size (the stats CSV averages both together)
tr = glob.glob(root + '/trace/*/*_kernel_trace.csv')
if tr and len(sys.argv) > 2 and sys.argv[2] == 'durations':
    d = []
    for r in csv.DictReader(open(tr[0])):
        if 'ntt_kernel<8, 4, true, 2' in r['Kernel_Name']:
            d.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    small, big = [x for x in d if x < 4 * min(d)], [x for x in d if x >= 4 * min(d)]
    print('ntt_kernel<8, 4, true, 2, *> in the kernel trace of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc`:', file=sys.stderr)
    print(f'  batch 65536 (BASELINE): {len(small)} launches, average {sum(small) / len(small) / 1e3:.2f} us, min {min(small) / 1e3:.2f} us', file=sys.stderr)
    if big:
        print(f'  batch 1048576 (past the Infinity Cache): {len(big)} launches, average {sum(big) / len(big) / 1e3:.2f} us', file=sys.stderr)
