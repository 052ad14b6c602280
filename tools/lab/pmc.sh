#!/bin/bash
# per-kernel average of PMC counters for one command (one rocprofv3 pass per counter group):
#   tools/lab/pmc.sh TAG "CTR1 CTR2" cmd args...
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $d; mkdir -p $d
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $d -o p -- "$@" > $d/stdout.log 2>&1)
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py $d | sed 's/(anonymous namespace):://g; s/void //' 
