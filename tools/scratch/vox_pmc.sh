mkdir -p gpurun_out/s2 && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf gpurun_out/s2/pmc_vox_$c
timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/s2/pmc_vox_$c -o pmc -- python3 tools/kbench_voxelize.py --rounds 1 "$@" > gpurun_out/s2/pmc_vox_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/s2/pmc_vox_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "vox" in r["Kernel_Name"]:
            acc[re.search(r"vox_\w+", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        h = v[len(v) // 2:]
        print(c, k, len(v), "avg KiB later half", round(sum(h) / len(h), 1))
PY
