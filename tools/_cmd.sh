for cfg in "1 4" "1 2" "1 1"; do
set -- $cfg
echo "## trav_cost=$1 max_leaf=$2"
TGH_BVH_TRAV_COST=$1 TGH_BVH_MAX_LEAF=$2 python bench.py --scene mesh1m --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  %.1f Msamples/s  nodes/ray %.2f prims/ray %.2f bvh %s  %s setup %s' % (d['value'], d['nodes_per_ray'], d['prims_per_ray'], d['bvh'], {k: v['avg_us'] for k, v in d['kernels'].items()}, d['setup_s']))"
done
