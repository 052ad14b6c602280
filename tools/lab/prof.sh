#!/bin/bash
# per-kernel averages of one command under rocprofv3: tools/lab/prof.sh TAG cmd args...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $d; mkdir -p $d
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- "$@" > $d/stdout.log 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    import re; name=re.sub(r"\(.*","",r["Name"].replace("(anonymous namespace)::","").replace("void ",""))[:70]
    print("%-70s calls %5s avg %9.1f us total %6.1f%%" % (name, r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
