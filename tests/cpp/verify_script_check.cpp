// tests/cpp/verify_script_check.cpp -- the wave-cooperative verifier's PROGRAM (csrc/verify_script.h) executed on the host
// with the concrete field, against the straight-line code the batched kernels run (miller_loop x 3, final_exponentiation):
// the GT value of e(A,B) e(vk_x,-gamma) e(C,-delta) must be identical for random points, and the program's shape is printed
// (steps per class, operations, slot file) -- what the device kernel's cost is made of.
//   hipcc -O2 -std=c++17 -I keyless-zk-proofs_amd/csrc tests/cpp/verify_script_check.cpp -o /tmp/vsc   (host code only)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "verify_script.h"

using namespace k16;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rand_scalar(uint8_t k[32])
{
    for (int i = 0; i < 4; i++) {
        uint64_t v = rnd();
        memcpy(k + 8 * i, &v, 8);
    }
    k[31] &= 0x1f;
}
static G1Aff g1_gen()
{
    Fq one = Fq::one(), two = fadd(one, one);
    return G1Aff{one, two};
}
static G2Aff g2_gen()
{
    return G2Aff{Fq2{fq_from_dec("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
                     fq_from_dec("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
                 Fq2{fq_from_dec("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
                     fq_from_dec("4082367875863433681332203403145435568316851327593401208105741076214120093531")}};
}

int main()
{
    {   // the binary-GCD inversion against Fermat's, random and edge values
        int badinv = 0;
        for (int i = 0; i < 300; i++) {
            uint8_t k[32];
            rand_scalar(k);
            Fq x;
            memcpy(x.v, k, 32);
            if (i == 0) x = Fq::zero();
            if (i == 1) x = Fq::one();
            if (i == 2) x = fneg(Fq::one());
            if (i == 3) { x = Fq::zero(); x.v[0] = 1; }
            if (i == 4) { x = Fq::zero(); x.v[0] = 2; }
            if (!(finv_bgcd(x) == finv(x))) badinv++;
        }
        printf("finv_bgcd vs finv: %s\n", badinv ? "MISMATCH" : "300 values identical");
        if (badinv) return 1;
    }
    PairConsts K;
    pairing_consts_init(&K);
    CoopProgram P;
    coop_build_program(K, &P);
    size_t nm = 0, nl = 0, ni = 0;
    for (uint8_t c : P.step_class) (c == CS_MUL ? nm : c == CS_LIN ? nl : ni)++;
    printf("program: %zu steps (mul %zu, lin %zu, inv %zu); ops mul %u lin %u inv %u; terms %zu; slots %u (constants %u); lines %u; depth %u levels, %u multiplications on the longest chain\n",
           P.step_class.size(), nm, nl, ni, P.n_mul_ops, P.n_lin_ops, P.n_inv_ops, P.terms.size(), P.n_slots, P.n_const, P.n_lines, P.depth, P.mul_depth);
    {   // a crude cost model of the device interpreter, in wave instructions: what the flattening limits trade
        double instr = 0;
        for (size_t st = 0; st < P.step_class.size(); st++) {
            if (P.step_class[st] == CS_MUL) instr += 760;
            else if (P.step_class[st] == CS_INV) instr += 25000;
            else {
                unsigned mx = 0;
                for (int l = 0; l < 64; l++) {
                    const uint64_t w = P.words[st * 64 + l];
                    if (w >> 63) mx = std::max<unsigned>(mx, (w >> 14) & 0x3f);
                }
                instr += 170 + 22.0 * mx;
            }
        }
        printf("cost model: %.0f k wave instructions\n", instr / 1e3);
    }
    int bad = 0;
    for (int trial = 0; trial < 3; trial++) {
        uint8_t k[5][32];
        for (auto& x : k) rand_scalar(x);
        G1Aff a = to_affine(pmul_scalar(G1Xyzz::from_aff(g1_gen()), k[0]));
        G1Aff c = to_affine(pmul_scalar(G1Xyzz::from_aff(g1_gen()), k[1]));
        G1Aff v = to_affine(pmul_scalar(G1Xyzz::from_aff(g1_gen()), k[2]));
        G2Aff b = to_affine(pmul_scalar(G2Xyzz::from_aff(g2_gen()), k[3]));
        G2Aff g = to_affine(pmul_scalar(G2Xyzz::from_aff(g2_gen()), k[4]));
        G2Aff d = to_affine(pmul_scalar(G2Xyzz::from_aff(g2_gen()), k[0]));
        // straight-line reference
        Fp12 f0, f1, f2, prod, want;
        miller_loop(&f0, &a, &b, &K);
        miller_loop(&f1, &v, &g, &K);
        miller_loop(&f2, &c, &d, &K);
        f12_mul(&prod, &f0, &f1);
        f12_mul(&prod, &prod, &f2);
        final_exponentiation(&want, &prod, &K);
        // the program
        std::vector<Ell> l1, l2;
        coop_prepare_lines(g, K, &l1);
        coop_prepare_lines(d, K, &l2);
        if (l1.size() != P.n_lines) {
            printf("line count %zu != %u\n", l1.size(), P.n_lines);
            return 1;
        }
        std::vector<Fq> slots;
        coop_const_table(K, want /* any target */, l1, l2, &slots);
        if (slots.size() != P.n_const) {
            printf("constant table %zu != %u\n", slots.size(), P.n_const);
            return 1;
        }
        slots.resize(P.n_slots, Fq::zero());
        // vk_x in projective form with a random (non-trivial) Z: X = x zz, Y = y zzz, zz = z^2, zzz = z^3
        uint8_t kz[32];
        rand_scalar(kz);
        Fq z;
        memcpy(z.v, kz, 32);
        const Fq zz = fsqr(z), zzz = fmul(zz, z), X = fmul(v.x, zz), Y = fmul(v.y, zzz);
        const Fq in[COOP_N_INPUTS] = {a.x, a.y, b.x.a, b.x.b, b.y.a, b.y.b, c.x, c.y, fmul(X, zzz), fmul(Y, zz), fmul(zz, zzz)};
        for (uint32_t i = 0; i < COOP_N_INPUTS; i++) slots[P.in_base + i] = in[i];
        coop_run_host(P, slots);
        const Fq2* w = &want.c0.c0;
        for (int i = 0; i < 6; i++)
            if (!(slots[P.out_slot[2 * i]] == w[i].a) || !(slots[P.out_slot[2 * i + 1]] == w[i].b)) bad++;
        printf("trial %d: %s\n", trial, bad ? "MISMATCH" : "GT value identical");
    }
    return bad ? 1 : 0;
}
