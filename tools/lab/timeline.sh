#!/bin/bash
# kernel timeline of one Keyless-shape proof: tools/lab/timeline.sh TAG [env assignments are inherited]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/tl_$tag
rm -rf $d; mkdir -p $d
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/bench_proof.py --proofs 8 > $d/stdout.log 2>&1)
tail -1 $d/stdout.log
python3 $GRAFT_REPO_ROOT/tools/proof_timeline.py $d 3 > $d/timeline.txt
cat $d/timeline.txt
