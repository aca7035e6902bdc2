export TMPDIR=/tmp; mkdir -p gpurun_out/s2; rm -rf gpurun_out/s2/bn1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2/bn1 -o bn -- python3 tools/scratch/bn_one.py "$@" > gpurun_out/s2/bn1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob("gpurun_out/s2/bn1/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "bn_" in r["Kernel_Name"]]
# group consecutive launches: per (kernel, grid) median duration
acc=collections.defaultdict(list)
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:40]
    acc[(k, r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in acc.items():
    v.sort(); print(k, len(v), "median us", round(v[len(v)//2]/1e3,2))
PY
