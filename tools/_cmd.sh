python bench.py --scene materialtest --steps 8 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single: %.1f Msamples/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
for i in 1 2; do
python bench.py --scene materialtest --steps 8 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('concurrent $i: %.1f Msamples/s %.2f ms/step' % (d['value'], d['ms_per_step']))" &
done
wait
for i in 1 2; do
python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cornell concurrent $i: %.1f Msamples/s %.2f ms/step' % (d['value'], d['ms_per_step']))" &
done
wait
