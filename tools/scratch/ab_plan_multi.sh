# A/B of several variant libraries of the plan forward (kbench_camera interleaves them): VARIANTS="name ..." [SHAPES="cfg4:f32 cfg5:bf16"] [MMT_PLAN_WGS=n]
mkdir -p gpurun_out/s2
LIBS="mm_training_amd/libmmt_hip.so"; for v in $VARIANTS; do LIBS="$LIBS mm_training_amd/variants/libmmt_$v.so"; done
for sh in ${SHAPES:-cfg4:f32 cfg5:bf16}; do s=${sh%%:*}; dt=${sh##*:}
timeout -k 10 300 python tools/kbench_camera.py --shape $s --dtype $dt --cases plan_prepared --rounds 3 $LIBS > gpurun_out/s2/abm_$s.json 2> gpurun_out/s2/abm_$s.err || tail -3 gpurun_out/s2/abm_$s.err
python3 - <<PY
import json
t=open("gpurun_out/s2/abm_$s.json").read(); dec=json.JSONDecoder(); i=0; objs=[]
while True:
    try: j=t.index("{",i)
    except ValueError: break
    try: o,e=dec.raw_decode(t,j); objs.append(o); i=e
    except Exception: i=j+1
for o in objs:
    if "us" in o: print("$s wgs=${MMT_PLAN_WGS:-default}", {k:v for k,v in o["us"].items() if "plan_prepared" in k})
PY
done
