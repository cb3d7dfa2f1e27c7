for wh in "640 480" "512 512" "768 512" "640 416" "1024 512"; do set -- $wh
python bench.py --steps 20 --warmup 5 --only-main --repeats 2 --width $1 --height $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1 x $2 tiles', (($1+15)//16)*(($2+15)//16), 'k_fuse us', r['avg_launch_us'], 'frac', r['frac'], 'n_upd', d['config']['n_upd_per_frame'], 'fps', d['value'], 'fused_only', d['config']['fused_only_fps'])
"
done
