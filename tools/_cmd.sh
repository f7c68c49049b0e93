for n in 1 2 4 8; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --emulate-shards $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('shards $n: ms_per_step %.2f  iters %d' % (d['ms_per_step'], d['wavefront_iterations']))"
python bench.py --scene materialtest --steps 4 --warmup 1 --no-cpu-baseline --no-extra --emulate-shards $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  materialtest shards $n: ms_per_step %.2f  iters %d' % (d['ms_per_step'], d['wavefront_iterations']))"
done
