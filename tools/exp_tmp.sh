A="--no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5"
run() { python3 bench.py $A "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', round(d['ms_per_step'],4))"; }
for r in 0 14 18 22 27 33 40 52 64 80; do run --dim-y 1024 --fuse 12 --sor-rows $r; done
for r in 0 20 27 33 40 48 56 66 80 100; do run --dim-y 2048 --fuse 16 --sor-rows $r; done
for r in 0 120 160 200 235 280 340; do run --fuse 16 --sor-rows $r; done
for r in 0 8 10 12 16 20 26 32; do run --size 2048 --iters 40 --fuse 8 --sor-rows $r; done
