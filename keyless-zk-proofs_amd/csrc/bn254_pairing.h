// bn254_pairing.h -- optimal-ate pairing on BN254 and the Groth16 pairing check, for the batched verifier
// (include/k16.h: k16_verify_batch).  __host__ __device__: the kernels of verify.hip and the host-side cross-check
// (tests/cpp/pairing_check.cpp) compile the same code.
//
// What it replaces.  The service re-verifies every proof on the CPU before releasing it
// (prover-service/src/request_handler/prover_handler.rs:329-336 -> aptos-types Groth16Proof::verify_proof ->
// ark-groth16 0.4.0 verify_proof_with_prepared_inputs -> ark-ec 0.4.2 Bn::multi_miller_loop / final_exponentiation on
// ark-bn254 0.4.0 -- third-party crates pinned in Cargo.lock:501-639, not vendored under /root/reference).  The pairing
// VALUE is fixed by the exponent of the final exponentiation, which follows ark-ec's (easy part, then the
// Fuentes-Castaneda hard part, i.e. (p^12 - 1)/r times 2x(6x^2 + 3x + 1)); how the tower arithmetic is organised does not
// change any value (field elements are canonical), so it is organised for the GPU here:
//   Fp6 by Karatsuba (6 Fq2 products), Fp12 squaring by the complex method, sparse line multiplication (13 Fq2 products),
//   Granger-Scott squaring in the cyclotomic subgroup for the three x-powers of the hard part.
// One lane computes one Miller loop or one final exponentiation: a batch of proofs is 3n independent Miller loops and n
// final exponentiations.  The large functions are deliberately NOT inlined (one copy of each in the code object).
#pragma once
#include "bn254_curve.h"

#define K16_HDN __host__ __device__ __attribute__((noinline))

namespace k16 {

#include "bn254_pairing_body.inc"

// ---------------------------------------------------------------- host: constants
// decimal (standard form) -> Montgomery, on the host
inline Fq fq_from_dec(const char* s)
{
    Fq ten = Fq::zero();
    ten.v[0] = 10;
    ten      = to_mont(ten);
    Fq acc   = Fq::zero();
    for (const char* p = s; *p; p++) {
        Fq d   = Fq::zero();
        d.v[0] = (uint32_t)(*p - '0');
        acc    = fadd(fmul(acc, ten), to_mont(d));
    }
    return acc;
}
inline void pairing_consts_init(PairConsts* K)
{
    auto f2 = [](const char* a, const char* b) { return Fq2{fq_from_dec(a), fq_from_dec(b)}; };
    // values: xi^((p^k-1)/3), xi^(2(p^k-1)/3), xi^((p^k-1)/6) for xi = 9 + u (ark-bn254 fq6.rs / fq12.rs FROBENIUS_COEFF tables),
    // 3/xi (g2.rs COEFF_B), xi^((p-1)/3), xi^((p-1)/2) (TWIST_MUL_BY_Q_X / _Y); recomputed by tests/test_oracle_pairing.py
    K->twist_b = f2("19485874751759354771024239261021720505790618469301721065564631296452457478373",
                    "266929791119991161246907387137283842545076965332900288569378510910307636690");
    K->twqx = f2("21575463638280843010398324269430826099269044274347216827212613867836435027261",
                 "10307601595873709700152284273816112264069230130616436755625194854815875713954");
    K->twqy = f2("2821565182194536844548159561693502659359617185244120367078079554186484126554",
                 "3505843767911556378687030309984248845540243509899259641013678093033130930403");
    K->frob6_c1[0] = K->frob6_c2[0] = K->frob12_c1[0] = Fq2::one();
    K->frob6_c1[1] = K->twqx;
    K->frob6_c2[1] = f2("2581911344467009335267311115468803099551665605076196740867805258568234346338",
                        "19937756971775647987995932169929341994314640652964949448313374472400716661030");
    K->frob12_c1[1] = f2("8376118865763821496583973867626364092589906065868298776909617916018768340080",
                         "16469823323077808223889137241176536799009286646108169935659301613961712198316");
    K->frob6_c1[2] = f2("21888242871839275220042445260109153167277707414472061641714758635765020556616", "0");
    K->frob6_c2[2] = f2("2203960485148121921418603742825762020974279258880205651966", "0");
    K->frob12_c1[2] = f2("21888242871839275220042445260109153167277707414472061641714758635765020556617", "0");
    K->frob6_c1[3] = f2("3772000881919853776433695186713858239009073593817195771773381919316419345261",
                        "2236595495967245188281701248203181795121068902605861227855261137820944008926");
    K->frob6_c2[3] = f2("5324479202449903542726783395506214481928257762400643279780343368557297135718",
                        "16208900380737693084919495127334387981393726419856888799917914180988844123039");
    K->frob12_c1[3] = f2("11697423496358154304825782922584725312912383441159505038794027105778954184319",
                         "303847389135065887422783454877609941456349188919719272345083954437860409601");
    K->two_inv = fq_from_dec("10944121435919637611123202872628637544348155578648911831344518947322613104292");
}

} // namespace k16
