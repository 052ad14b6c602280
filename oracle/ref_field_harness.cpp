// oracle/ref_field_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// extern "C" wrapper around the REFERENCE's own field implementation
// (RawFq / RawFr / F2Field<RawFq>), compiled from the sources where they lie
// under /root/reference by build_ref.sh into oracle/_ref/libref_field.so.
// Used only in this container to validate the restatement in field.h /
// bn254_ref.c on random inputs.  Nothing from the reference is copied here.
#include <cstdint>
#include <cstring>
#include "fq.hpp"
#include "fr.hpp"
#include "f2field.hpp"

static RawFq          g_fq;
static RawFr          g_fr;
static F2Field<RawFq> g_f2("-1");

extern "C" {
// op: 0 add 1 sub 2 neg 3 mul 4 sqr 5 toMont 6 fromMont 7 inv   (matches ORA_OP_*)
void ref_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r)
{
    if (field == 0) {
        RawFq::Element x, y, z;
        memcpy(x.v, a, 32);
        if (b) memcpy(y.v, b, 32);
        switch (op) {
        case 0: g_fq.add(z, x, y); break;
        case 1: g_fq.sub(z, x, y); break;
        case 2: g_fq.neg(z, x); break;
        case 3: g_fq.mul(z, x, y); break;
        case 4: g_fq.square(z, x); break;
        case 5: g_fq.toMontgomery(z, x); break;
        case 6: g_fq.fromMontgomery(z, x); break;
        case 7: g_fq.inv(z, x); break;
        }
        memcpy(r, z.v, 32);
    } else {
        RawFr::Element x, y, z;
        memcpy(x.v, a, 32);
        if (b) memcpy(y.v, b, 32);
        switch (op) {
        case 0: g_fr.add(z, x, y); break;
        case 1: g_fr.sub(z, x, y); break;
        case 2: g_fr.neg(z, x); break;
        case 3: g_fr.mul(z, x, y); break;
        case 4: g_fr.square(z, x); break;
        case 5: g_fr.toMontgomery(z, x); break;
        case 6: g_fr.fromMontgomery(z, x); break;
        case 7: g_fr.inv(z, x); break;
        }
        memcpy(r, z.v, 32);
    }
}
void ref_field_op_vec(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) ref_field_op(field, op, a + 4 * i, b ? b + 4 * i : nullptr, r + 4 * i);
}
void ref_fq2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* r)
{
    F2Field<RawFq>::Element x, y, z;
    memcpy(&x, a, 64);
    if (b) memcpy(&y, b, 64);
    switch (op) {
    case 0: g_f2.add(z, x, y); break;
    case 1: g_f2.sub(z, x, y); break;
    case 2: g_f2.neg(z, x); break;
    case 3: g_f2.mul(z, x, y); break;
    case 4: g_f2.square(z, x); break;
    case 7: g_f2.inv(z, x); break;
    }
    memcpy(r, &z, 64);
}
int ref_fe_to_dec(int field, const uint64_t* a, char* out)
{
    std::string s;
    if (field == 0) {
        RawFq::Element x;
        memcpy(x.v, a, 32);
        s = g_fq.toString(x, 10);
    } else {
        RawFr::Element x;
        memcpy(x.v, a, 32);
        s = g_fr.toString(x, 10);
    }
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
}
