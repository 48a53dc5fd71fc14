A="--no-cpu-baseline --sim-steps 0 --steps 200 --warmup 20"
run() { python3 bench.py $A "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', round(d['ms_per_step']*1000,1),'us')"; }
for f in 8 10 12 14 16; do
 run --size 61 --dim-y 81 --iters 20 --fuse $f
 run --size 61 --dim-y 81 --iters 10 --fuse $f
 run --size 256 --dim-y 192 --iters 40 --fuse $f
 run --size 512 --dim-y 512 --iters 40 --fuse $f
done
