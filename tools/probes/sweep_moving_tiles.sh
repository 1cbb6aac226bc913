R=$GRAFT_REPO_ROOT
for tj in 0 7 10 13 19; do for tt in 0 4 16 32; do
  tun=""; [ $tj -ne 0 ] && tun="tile_j=$tj"; [ $tt -ne 0 ] && tun="${tun:+$tun,}tile_t=$tt"
  python3 $R/bench.py --moving --timesteps 512 --cpu-baseline none --steps 10 --warmup 3 ${tun:+--tuning $tun} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile_j=$tj tile_t=$tt  launch %.4f ms  frac %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
