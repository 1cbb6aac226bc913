for tj in "4 8" "4 16" "8 8" "8 16" "2 16" "4 12" "6 8" "8 4" "16 4" "4 24" "2 32"; do set -- $tj; python3 bench.py --cpu-baseline none --steps 8 --warmup 2 --storage f32 --tuning kernel=row_sweep,order=xcd_tiled,tile_t=$1,tile_j=$2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tile_t $1 tile_j $2: launch ms %8.3f  frac %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; done
