python bench.py 2>gpurun_out/bench_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
e = d.pop('extra', {})
print(json.dumps(d)[:1800])
for k, v in e.items(): print(k, json.dumps(v)[:1500])
"
tail -3 gpurun_out/bench_err.log
