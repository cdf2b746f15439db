import sys, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_falcon as bf
for s, b in ((2, 8192), (2, 32768), (2, 131072), (0, 8192), (0, 131072)):
    print(json.dumps(bf.run(s, b)), flush=True)
