#!/bin/bash
REPS=${REPS:-6} tools/lab/run_ab_tailfit.sh K16_WITNESS_C=12 - 2>&1 | sed -e "s/stages.*proof/proof/" | cut -c1-20,50-200
