for tj in "2 4" "2 8" "2 16" "1 8" "1 16" "4 4" "4 8" "2 6" "3 4" "2 12"; do set -- $tj; python3 bench.py --cpu-baseline none --steps 6 --warmup 2 --tuning kernel=row_block,tile_t=$1,tile_j=$2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('row_block tile_t $1 tile_j $2 (blocks of 2 steps x 2 lats): launch ms %8.3f  frac %.4f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; done
