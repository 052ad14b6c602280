#!/usr/bin/env python3
"""Extracts the raw Montgomery known-answer VECTORS (inputs + expected outputs, data only)
that the reference's own test file declares for Fr_Rw_* / Fq_Rw_* into field_kats.json.

Run in the build container only (reads /root/reference):
    python tests/golden/make_field_kats.py

Source of the vectors: rust-rapidsnark/rapidsnark/src/test_prover.cpp
  Fr_Rw_{Neg,add,sub,mul,Msquare,ToMontgomery,FromMontgomery}_unit_test  (lines 152-690)
  Fq_Rw_{...}_unit_test                                                (lines 13164-13700)
As SURVEY.md section 4 notes, some compare_Result() calls in that file are wired to the
wrong expected vector; the DECLARED pRawResultN constants are what is extracted.
"""
import json
import os
import re

SRC = "/root/reference/rust-rapidsnark/rapidsnark/src/test_prover.cpp"
OPS = {"Neg": "neg", "add": "add", "sub": "sub", "mul": "mul", "Msquare": "sqr",
       "ToMontgomery": "tomont", "FromMontgomery": "frommont"}

text = open(SRC).read()
out = []
for field in ("Fr", "Fq"):
    for cname, op in OPS.items():
        m = re.search(r"void %s_Rw_%s_unit_test\(\)\s*\{(.*?)\n\}" % (field, cname), text, re.S)
        if not m:
            continue
        body = m.group(1)
        decl = dict()
        for d in re.finditer(r"%sRawElement\s+(pRaw\w+?)(\d+)\s*=\s*\{([^}]*)\}" % field, body):
            name, idx, vals = d.group(1), int(d.group(2)), d.group(3)
            limbs = [int(v.strip(), 16) for v in vals.split(",") if v.strip()]
            if len(limbs) == 4:
                decl[(name, idx)] = limbs
        idxs = sorted({i for (n, i) in decl if n == "pRawResult"})
        for i in idxs:
            a = decl.get(("pRawA", i))
            b = decl.get(("pRawB", i))
            r = decl.get(("pRawResult", i))
            if a is None or r is None:
                continue
            out.append({"field": field, "op": op, "case": i,
                        "a": ["0x%016x" % v for v in a],
                        "b": ["0x%016x" % v for v in b] if b else None,
                        "r": ["0x%016x" % v for v in r]})
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "field_kats.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", len(out), "vectors to", dst)
