#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/k16_tp; rocprofv3 --kernel-trace --output-format csv -d /tmp/k16_tp -- python3 tools/bench_proof.py --proofs 60 --concurrent 2 --no-stats > /tmp/tp.json 2>/tmp/tp.err
tail -c 200 /tmp/tp.json; echo
python3 tools/two_prover_phases.py /tmp/k16_tp 80 | head -12
mkdir -p gpurun_out/r05; cd tools && python3 -c 'import two_prover_phases as t; t.window("/tmp/k16_tp", 20, 45)' > ../gpurun_out/r05/two_prover_window.txt
