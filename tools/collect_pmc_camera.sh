#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_pmc_camera.sh'): the three PMC passes of tools/collect_profiles.sh on the fused
# lift-splat kernels ALONE (tools/kbench_camera.py: the plan forward and the column backward at BASELINE configs[3]'s shape in fp32,
# then configs[4]'s in bf16; warm launches only).  The training bench does not survive `rocprofv3 --pmc` on this image (DESIGN
# section 5); the operator's own launches do.  -> gpurun_out/profiles_new/pmc_camera_cfg4.json / pmc_camera_cfg5.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new; mkdir -p $out
raw=/tmp/mmt_prof_cam; rm -rf $raw
step() { local log=$1; shift; timeout -k 10 200 "$@" > "$log" 2>&1; local rc=$?; echo "[profiles] $(basename $log) rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then tail -5 "$log"; exit $rc; fi; }
for sh in cfg4:f32 cfg5:bf16; do
  shape=${sh%%:*}; dt=${sh##*:}
  args="--shape $shape --dtype $dt --rounds 1 --cases ^fwd_plan_prepared.kernel.\$|^col_bwd_cam_summary\$"
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum"; do
    i=$((i+1))
    step $out/pmc_camera_${shape}_$i.log rocprofv3 --pmc $ctrs --output-format csv -d $raw/${shape}_$i -o pmc -- python3 tools/kbench_camera.py $args
  done
  python3 tools/aggregate_pmc.py "python tools/kbench_camera.py $args" $raw/${shape}_1 $raw/${shape}_2 $raw/${shape}_3 > $out/pmc_camera_$shape.json
done
ls -la $out | grep pmc_camera
