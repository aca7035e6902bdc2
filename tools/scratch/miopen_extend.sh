# Run ON THE GPU BOX: extend the shipped MIOpen find DB by the shapes new code paths added (one ordinary find per new problem, the
# shipped files seed the directory so known shapes are answered from it).  Result: gpurun_out/miopen_db_ext/*.txt (copy over
# mm_training_amd/miopen_db/).   usage: bash tools/scratch/miopen_extend.sh cfg4 cfg5 cfg3
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/miopen_db_ext; rm -rf $out; mkdir -p $out
cp mm_training_amd/miopen_db/*.txt $out/
wc -l $out/*.txt
for cfg in "$@"; do
  t0=$(date +%s)
  env MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0 \
      MIOPEN_USER_DB_PATH=$out MIOPEN_FIND_MODE=1 MIOPEN_FIND_ENFORCE=1 \
      timeout -k 10 400 python3 bench.py --config $cfg --miopen-tune --steps 4 --warmup 2 --no-cpu-baseline --no-hotpath-leg > $out/tune_$cfg.log 2>&1
  rc=$?; t1=$(date +%s); echo "$cfg rc=$rc wall $((t1-t0)) s"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
wc -l $out/*.txt
