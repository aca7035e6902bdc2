# A/B of variant libraries of the column backward: VARIANTS="name ..." [SHAPES="cfg4:f32 cfg5:bf16"]
LIBS="mm_training_amd/libmmt_hip.so"; for v in $VARIANTS; do LIBS="$LIBS mm_training_amd/variants/libmmt_$v.so"; done
for sh in ${SHAPES:-cfg4:f32 cfg5:bf16}; do s=${sh%%:*}; dt=${sh##*:}
python tools/kbench_camera.py --shape $s --dtype $dt --cases "col_bwd_cam_summary" --rounds 3 $LIBS 2>/dev/null | python3 -c "
import sys, json
t=sys.stdin.read(); dec=json.JSONDecoder(); i=0
while True:
    try: j=t.index('{',i)
    except ValueError: break
    try:
        o,e=dec.raw_decode(t,j); i=e
        if 'us' in o: print('$s', {k:v for k,v in o['us'].items() if 'col_bwd' in k})
    except Exception: i=j+1
"
done
