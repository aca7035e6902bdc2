mkdir -p gpurun_out/s2
for sh in "cfg4 f32" "cfg5 bf16"; do set -- $sh
timeout -k 10 200 python tools/kbench_camera.py --shape $1 --dtype $2 --cases plan_prepared --rounds 3 mm_training_amd/libmmt_hip.so "$VARIANT" > gpurun_out/s2/ab_plan_$1.json 2> gpurun_out/s2/ab_plan_$1.err; tail -2 gpurun_out/s2/ab_plan_$1.err | cut -c1-300
python3 - <<PY
import json
t=open("gpurun_out/s2/ab_plan_$1.json").read(); dec=json.JSONDecoder()
try:
    o1,j=dec.raw_decode(t,t.index("{")); o2,_=dec.raw_decode(t,t.index("{",j)); print("$1", {k:v for k,v in o2["us"].items() if "plan_prepared" in k})
except Exception as e: print("unparsed", e, t[-300:])
PY
done
