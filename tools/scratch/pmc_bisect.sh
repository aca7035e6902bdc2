# one PMC pass (FETCH_SIZE) of the default bench under two settings, stopping at the first failure
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmcb
one() { # tag, env...
  local tag=$1; shift
  rm -rf /tmp/pmcb_$tag
  env "$@" timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcb_$tag -o pmc -- python3 bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-hotpath-leg > gpurun_out/pmcb/$tag.log 2>&1
  local rc=$?; echo "$tag rc=$rc"; return $rc
}
one event_fresh MMT_REUSE_FORK_EVENT=0 && one event_reuse MMT_REUSE_FORK_EVENT=1
