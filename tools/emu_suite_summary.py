#!/usr/bin/env python3
"""Summary of a tools/emu_gpu_suite.sh run for profiles/: tools/emu_suite_summary.py gpurun_out/emu_suite/<kind>.txt [more runs ...]"""
import re
import sys


def summarize(path):
    text = open(path, errors='replace').read()
    failed = sorted(set(re.sub(r' - .*', '', l[len('FAILED '):]) for l in text.splitlines() if l.startswith('FAILED ')))
    timeouts = set(re.findall(r'^(tests/\S+) \+{5,} Timeout', text, re.M))
    tail = [l for l in text.splitlines() if re.search(r'\d+ passed', l)]
    print(f'== {path}')
    print(tail[-1].strip('= ') if tail else '(run did not finish)')
    # a test the per-test limit ended: pytest-timeout (thread method) prints its stack and takes the worker down
    limit = [t for t in failed if 'Timeout' in text and re.search(re.escape(t) + r'.{0,40}Timeout|Timeout.{0,2000}' + re.escape(t.split('::')[-1].split('[')[0]), text, re.S)]
    print(f'{len(failed)} did not pass:')
    for t in failed:
        print('   ', t)


for p in sys.argv[1:]:
    summarize(p)
