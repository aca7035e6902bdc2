mkdir -p gpurun_out/s2 && export TMPDIR=/tmp
rm -rf gpurun_out/s2/tax
MASTER_PORT=29733 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2/tax -o tax -- python3 tools/scratch/ddp_tax.py native 6 > gpurun_out/s2/tax.log 2>&1
tail -2 gpurun_out/s2/tax.log | cut -c1-200
f=$(find gpurun_out/s2/tax -name "*kernel_trace.csv" | head -1)
python3 tools/reducer_timeline.py $f > gpurun_out/s2/reducer_timeline.txt 2>&1; cat gpurun_out/s2/reducer_timeline.txt
head -1 $f
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
c=collections.Counter(r["Kernel_Name"][:80] for r in rows if "multi_tensor" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower())
for k,v in c.most_common(12): print(v,k)
PY
rm -f $f.keep; ls -la $f
