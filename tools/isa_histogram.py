#!/usr/bin/env python3
"""Instruction-class histogram of one kernel from hipcc's device assembly.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -S csrc/msm_g1.hip -o /tmp/msm_g1.s
  python3 tools/isa_histogram.py /tmp/msm_g1.s k_accumulate.*Eng9 [--loop]

Counts STATIC instructions of the kernel body (or, with --loop, of its hottest basic-block range: the largest
backward-branch loop) by class.  Static counts of the main loop are what a lane executes per mixed addition when the
loop body is straight-line code, which it is for the accumulation (the exceptional branches are outlined blocks)."""
import re, sys, collections

def classify(op):
    if op.startswith('v_mad_u64_u32') or op.startswith('v_mad_co_u64_u32'): return 'mad_u64_u32 (multiply-add)'
    if op.startswith('v_mul_lo_u32') or op.startswith('v_mul_hi_u32') or op.startswith('v_mul_u32_u24'): return 'mul_lo/hi (m_k)'
    if op.startswith('v_lshl_add_u64') or op.startswith('v_add_co') or op.startswith('v_addc') or op.startswith('v_add_u64'): return 'add64 / add-with-carry'
    if op.startswith('v_sub_co') or op.startswith('v_subb') or op.startswith('v_subrev_co') or op.startswith('v_subbrev'): return 'sub-with-borrow'
    if re.match(r'v_(lshrrev|lshlrev|ashrrev)_b64', op) or op.startswith('v_alignbit') or op.startswith('v_alignbyte'): return 'shift64 / alignbit'
    if re.match(r'v_(lshrrev|lshlrev|ashrrev)_[bi]32', op) or op.startswith('v_bfe') or op.startswith('v_lshl_or') or op.startswith('v_lshl_add_u32') or op.startswith('v_and_or') : return 'shift32 / bfe / lshl_or'
    if op.startswith('v_and_b32') or op.startswith('v_or_b32') or op.startswith('v_or3') or op.startswith('v_xor') or op.startswith('v_not') or op.startswith('v_bfi'): return 'and / or / xor'
    if op.startswith('v_add3') or op.startswith('v_add_u32') or op.startswith('v_sub_u32') or op.startswith('v_subrev_u32') or op.startswith('v_add_nc') or op.startswith('v_sub_nc') or op.startswith('v_add_lshl') or op.startswith('v_sub_i32') or op.startswith('v_add_i32'): return 'add32 / sub32 / add3'
    if op.startswith('v_cmp') or op.startswith('v_cmpx'): return 'compare'
    if op.startswith('v_cndmask'): return 'cndmask (select)'
    if op.startswith('v_mov') or op.startswith('v_accvgpr') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane') or op.startswith('v_swap') or op.startswith('v_pk_mov'): return 'move / accvgpr / lane'
    if op.startswith('v_'): return 'other VALU: ' + op.split('_e')[0]
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'VMEM'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier') or op.startswith('s_sleep') or op.startswith('s_setprio'): return 'wait / nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'SALU'
    return 'other: ' + op

def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    loop_only = '--loop' in sys.argv
    lines = open(path, errors='replace').read().split('\n')
    # kernel bodies: "<mangled>:" ... "s_endpgm"
    i, n = 0, len(lines)
    found = 0
    while i < n:
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', lines[i])
        if m and pat.search(m.group(1)):
            name = m.group(1); j = i + 1; body = []
            while j < n and not lines[j].startswith('.Lfunc_end'):
                body.append(lines[j]); j += 1
            report(name, body, loop_only); found += 1
            i = j
        i += 1
    if not found: print('no kernel matches', sys.argv[2])

def report(name, body, loop_only):
    # (label, op) stream
    ins = []
    labels = {}
    for ln in body:
        s = ln.strip()
        if not s or s.startswith(';') or s.startswith('.') and not s.endswith(':'): continue
        m = re.match(r'^(\.?\w+):', s)
        if m: labels[m.group(1)] = len(ins); continue
        op = s.split()[0]
        tgt = None
        if op.startswith('s_cbranch') or op.startswith('s_branch'):
            t = s.split()[1] if len(s.split()) > 1 else ''
            tgt = t
        ins.append((op, tgt))
    lo, hi = 0, len(ins)
    title = 'whole kernel'
    if loop_only:
        best = (0, 0, 0)
        for k, (op, tgt) in enumerate(ins):
            if tgt in labels and labels[tgt] <= k:
                span = k - labels[tgt] + 1
                if span > best[0]: best = (span, labels[tgt], k + 1)
        if best[0]:
            lo, hi = best[1], best[2]; title = 'largest loop (%d instructions)' % best[0]
    h = collections.Counter(classify(op) for op, _ in ins[lo:hi])
    tot = sum(h.values())
    valu = sum(v for k, v in h.items() if k.split()[0] not in ('VMEM', 'LDS', 'wait', 'branch', 'SALU', 'other:'))
    print('== %s\n   %s: %d instructions, %d VALU' % (name, title, tot, valu))
    for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
        print('   %6d  %5.1f%%  %s' % (v, 100.0 * v / tot, k))

main()
