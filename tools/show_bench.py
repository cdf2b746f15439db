"""prints a bench.py JSON line in short form"""
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps({k: v for k, v in d.items() if k != 'extra'}, indent=1)[:3500])
for k, v in d['extra'].items():
    print(k, json.dumps(v)[:700])
