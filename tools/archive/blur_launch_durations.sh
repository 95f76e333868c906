# per-launch blur durations of one serial step (kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/tile
mkdir -p $D
cd $R
SARA_HIP_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $D -o tile -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 > $D/tile.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/tile/tile_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'descriptor_kernel' in r['Kernel_Name']]
a=idx[-2]+1; b=idx[-1]+1
for r in rows[a:b]:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('sara_hip::','')
    if 'blur' in n or 'scale' in n:
        print(f"{n[:42]:42s} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
