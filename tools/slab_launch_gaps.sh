export TMPDIR=/tmp
mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gaps/t -o trace -- python3 bench.py --dim-y 1024 --fuse 8 --steps 10 --warmup 2 --no-cpu-baseline --sim-steps 0 > gpurun_out/gaps/run.log 2>&1
python3 bench.py --dim-y 1024 --fuse 8 --steps 10 --warmup 2 --no-cpu-baseline --sim-steps 0 > gpurun_out/gaps/plain.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/gaps/t/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'sor_fused' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
g=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
g2=[x for x in g if x<100000]
print('launches',len(rows),'avg dur us',sum(d)/len(d)/1e3,'avg gap us (within solve)',sum(g2)/len(g2)/1e3,'min gap',min(g)/1e3, 'median gap', sorted(g2)[len(g2)//2]/1e3)
PY
