#!/bin/bash
# VALU wave-instructions per proof, by kernel: tools/lab/valu_per_proof.sh TAG   (environment inherited)
tag=$1
cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/valu_$tag
rm -rf $d; mkdir -p $d
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $d -o p -- python3 tools/bench_proof.py --proofs 6 --no-stats > $d/stdout.log 2>&1)
python3 - $d <<'PY'
import csv,glob,sys,re,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(f)):
    n=re.sub(r"\(.*","",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ",""))[:60]
    acc[n][0]+=float(r["Counter_Value"]); acc[n][1]+=1
# proofs in the run: 1 create warm-up + 1 bench warm-up + 6 = 8 ; k_hscalars launches = proofs
proofs=acc["k_hscalars"][1]
tot=0
for n,(v,c) in sorted(acc.items(), key=lambda kv:-kv[1][0]):
    if c < proofs: continue
    print("%-62s %8.1f M wave-instr/proof  (%4.1f launches/proof)" % (n, v/proofs/1e6, c/proofs)); tot+=v/proofs
print("total %.1f M wave-instr per proof" % (tot/1e6))
PY
