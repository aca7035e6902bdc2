import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], r["value"], r["ms_per_step"])
for k in ("roofline","roofline_backward"):
    v=r.get(k); print(" ",k, v and (round(v["avg_ms"]*1e3,1), round(v["frac"],3)))
