#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_pmc_dcn.sh'): the three PMC passes of tools/collect_profiles.sh on the deformable
# convolution ALONE (tools/scratch/dcn_prof.py: six forward + backward calls of the operator at the DepthNet shape of BASELINE
# configs[3], then configs[4]'s).  The training bench itself does not survive `rocprofv3 --pmc` on this image (the profiler's queue
# interception: DESIGN section 5); the operator's own launches do.  -> gpurun_out/profiles_new/pmc_dcn_cfg4.json / pmc_dcn_cfg5.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new; mkdir -p $out
raw=/tmp/mmt_prof_dcn; rm -rf $raw
step() { local log=$1; shift; timeout -k 10 200 "$@" > "$log" 2>&1; local rc=$?; echo "[profiles] $(basename $log) rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then tail -5 "$log"; exit $rc; fi; }
for shape in cfg4 cfg5; do
  arg=""; [ $shape = cfg5 ] && arg="--cfg5"
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum"; do
    i=$((i+1))
    step $out/pmc_dcn_${shape}_$i.log rocprofv3 --pmc $ctrs --output-format csv -d $raw/${shape}_$i -o pmc -- python3 tools/scratch/dcn_prof.py $arg
  done
  python3 tools/aggregate_pmc.py "python tools/scratch/dcn_prof.py $arg" $raw/${shape}_1 $raw/${shape}_2 $raw/${shape}_3 > $out/pmc_dcn_$shape.json
done
ls -la $out | grep pmc_dcn
