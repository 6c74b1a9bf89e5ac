cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/r02n
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/r02n/kt2 -o out --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02n/kt2.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$R/gpurun_out/r02n/kt2/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if 'k_sqp_pool' in r['Kernel_Name'] or 'k_prepare' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:12],r.get('Queue_Id','?'),r.get('Stream_Id','?')))
rows.sort()
t0=rows[0][0]
for a,b,n,q,s in rows: print(f"{n:12s} q{q} s{s} start {(a-t0)/1e6:9.2f} ms  end {(b-t0)/1e6:9.2f} ms  dur {(b-a)/1e6:8.2f}")
PY
