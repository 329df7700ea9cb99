for cfg in "224 32" "232 24" "240 24" "240 12" "244 12" "236 20"; do set -- $cfg; echo "grid $1 gens $2:"; SGDNET_LDS_GRID=$1 SGDNET_RNG_GENERATORS=$2 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d.get('convergence',{})
print('  epochs/s %.1f ms %.3f gather %.1f us frac %.3f conv %s epochs %.4f s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], c.get('epochs'), c.get('seconds',0)))"; done
