cd $GRAFT_REPO_ROOT
for s in 0 448 512 0 512; do
  TSD_ICP_SHAPE=$s python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-second-pass --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shape $s: %.0f scans/s, ms_icp %.4f spread %s' % (d['value'], d['ms_icp_iterate'], d['ms_icp_iterate_spread']['max_over_mean']))"
done
