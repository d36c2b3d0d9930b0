cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r13_scope_prof -o p -- python3 -c "
import sys; sys.path.insert(0,'tools')
import bench_meters as b
b.scope_stereo()
" > gpurun_out/r13_scope_prof.log 2>&1
python3 - <<'PY' > gpurun_out/r13_scope_prof_stats.txt
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/r13_scope_prof/*results.db')[0])
cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit 12"
for r in cur.execute(q): print(r[0][:90], r[1], round(r[2],1), round(r[3],1))
PY
