#!/bin/bash
# usage (on the GPU box): bash tools/scratch/pmc_dcn.sh <tag> <counters...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dcn_pmc_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/tools/scratch/dcn_prof.py > $GRAFT_REPO_ROOT/gpurun_out/dcn_pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"
