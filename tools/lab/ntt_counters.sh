#!/bin/bash
# SQ counters of the NTT pass kernels (one rocprofv3 pass per group): where do the ~35 % of non-VALU time go?
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=n$(echo $grp | md5sum | cut -c1-6)
  tools/lab/pmc.sh $tag "$grp" python3 tools/ntt_timing.py 21 6 2>/dev/null | grep "k_ntt_pass9" | sed 's/k_ntt_pass9<\(.*\)>(NttPtrs.*unsigned/pass9<\1>/' | cut -c1-150
done
