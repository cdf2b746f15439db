"""CPU: bench.py's contract line can never outgrow what the driver parses (round 3's line was 27.7 KB and was not parsed; the line is now built by
bench.contract_line, which drops optional blocks - largest first - until it fits and never gives up the required keys).  Exercised on the last line a
driver-shaped run produced on hardware (profiles/r04_bench.json) and on inflated copies of it."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_line_fits_whatever_the_legs_produce():
    b = _bench()
    res = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')).read().strip().split('\n')[-1])
    line = b.contract_line(res)
    assert json.loads(line) == res and len(line) < b.LINE_LIMIT               # a normal line goes out as it is
    for inflate in (lambda r: r.update(multi_gpu={'x': 'y' * 20000}),
                    lambda r: r.update(north_star={k: dict(v, blob={'a': 'z' * 5000}) for k, v in r['north_star'].items()}),
                    lambda r: r.update(config=dict(r['config'], huge='q' * 30000), surprise=['w' * 100] * 400)):
        big = json.loads(json.dumps(res))
        inflate(big)
        before = json.dumps(big)
        line = b.contract_line(big)
        out = json.loads(line)
        assert len(line) < b.LINE_LIMIT and out.get('line_trimmed') is True
        assert all(k in out for k in b.REQUIRED_KEYS) and out['value'] == res['value'] and out['metric'] == res['metric']
        assert out['roofline']['frac'] == res['roofline']['frac'] and out['cpu_baseline']['value'] == res['cpu_baseline']['value'] and 'workload' in out['config']
        assert json.dumps(big) == before                                        # the caller's dict is untouched (it has gone to bench_extra.json already)
