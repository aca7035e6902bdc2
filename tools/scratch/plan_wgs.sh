mkdir -p gpurun_out/s2
for w in ${WLIST:-768 1024 1280 1536 2048}; do MMT_PLAN_WGS=$w timeout -k 10 120 python tools/kbench_camera.py --shape ${SHAPE:-cfg5} --dtype ${DT:-bf16} --cases plan --rounds 3 > gpurun_out/s2/plan_wgs_${SHAPE:-cfg5}_$w.json 2>/dev/null; done
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/s2/plan_wgs_*_*.json"), key=lambda f: (f.split("_")[-2], int(f.split("_")[-1][:-5]))):
    t=open(f).read(); dec=json.JSONDecoder()
    try:
        o1,j=dec.raw_decode(t,t.index("{")); o2,_=dec.raw_decode(t,t.index("{",j))
    except Exception as e:
        print(f, "unparsed"); continue
    us=o2["us"]
    print(os.path.basename(f), {k.replace("fwd_plan_prepared",""):list(v.values())[0] for k,v in us.items() if k.startswith("fwd_plan_prepared")})
PY
