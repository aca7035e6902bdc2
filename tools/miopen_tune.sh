#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/miopen_tune.sh cfg4'): extend the shipped MIOpen user find/perf DB
# (mm_training_amd/miopen_db) by the convolution shapes of another bench configuration.  The shipped files seed
# the search directory, so shapes that already have a record are answered from it; the result lands in
# gpurun_out/miopen_db_<cfg>/ (copy the two *.txt files over mm_training_amd/miopen_db/ to ship them).
cfg=${1:-cfg4}
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/miopen_db_$cfg; rm -rf $out; mkdir -p $out
cp mm_training_amd/miopen_db/*.txt $out/
NONAIVE="MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0"
t0=$(date +%s)
env $NONAIVE MIOPEN_USER_DB_PATH=$out MIOPEN_FIND_MODE=1 MIOPEN_FIND_ENFORCE=${MMT_FIND_ENFORCE:-3} timeout -k 10 900 python3 bench.py --config $cfg --miopen-tune --steps 10 --warmup 4 --no-cpu-baseline > $out/tune.log 2>&1
rc=$?; t1=$(date +%s); echo "tune rc=$rc wall $((t1-t0)) s"; grep '^{' $out/tune.log | cut -c1-200
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
t0=$(date +%s)
env $NONAIVE MIOPEN_USER_DB_PATH=$out MIOPEN_FIND_MODE=3 timeout -k 10 400 python3 bench.py --config $cfg --miopen-tune --steps 20 --warmup 8 --no-cpu-baseline > $out/mode3.log 2>&1
rc=$?; t1=$(date +%s); echo "mode3 rc=$rc wall $((t1-t0)) s"; grep '^{' $out/mode3.log | cut -c1-200
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
t0=$(date +%s)
MIOPEN_USER_DB_PATH=/tmp/mmt_nodb_$$ timeout -k 10 400 python3 bench.py --config $cfg --steps 20 --warmup 8 --no-cpu-baseline > $out/nodb.log 2>&1
rc=$?; t1=$(date +%s); echo "nodb rc=$rc wall $((t1-t0)) s"; grep '^{' $out/nodb.log | cut -c1-200
ls -la $out; wc -l $out/*.txt
