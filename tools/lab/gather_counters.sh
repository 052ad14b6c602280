#!/bin/bash
# TLB / L2 / fabric counters of the bucket accumulation over fixed-base tables of three sizes:
#   2^21 points (13 tables, 1.7 GB), 2^19 (15 tables, 0.5 GB), 2^17 (17 tables, 0.14 GB: Infinity-Cache resident)
# one rocprofv3 pass per counter group; prints the k_accumulate rows
for ln in 21 19 17; do
  echo "=== fixed-base MSM, n = 2^$ln"
  python3 tools/fixed_base_timing.py $ln 6 2>/dev/null | tail -1
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY SQ_WAVES SQ_INSTS_VALU" \
             "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
    tag=g$(echo $grp | md5sum | cut -c1-6)
    tools/lab/pmc.sh ${tag}_$ln "$grp" python3 tools/fixed_base_timing.py $ln 4 2>/dev/null | grep "k_accumulate" | sed 's/k_accumulate<Eng9>(Eng9::Row const\*, unsigned int const\*, unsi//'
  done
done
