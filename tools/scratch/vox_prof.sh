mkdir -p gpurun_out/s2 && export TMPDIR=/tmp
rm -rf gpurun_out/s2/prof_vox
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/s2/prof_vox -o vox -- python3 tools/kbench_voxelize.py --rounds 2 "$@" > gpurun_out/s2/prof_vox.log 2>&1
python3 - <<'PY'
import sqlite3, glob
c=sqlite3.connect(glob.glob('gpurun_out/s2/prof_vox/*.db')[0])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if 'kernel_symbol' in t][0]
for r in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {sym} s on d.kernel_id=s.id where s.kernel_name like '%vox%' group by s.kernel_name order by 3 desc"):
    print(r[0][:60], r[1], round(r[2]/1000,2), round(r[3]/1000,2))
PY
