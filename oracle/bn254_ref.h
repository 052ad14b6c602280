/*
 * oracle/bn254_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * C API of the CPU oracle (liboracle_bn254.so), loaded through ctypes by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 */
#ifndef ORACLE_BN254_REF_H
#define ORACLE_BN254_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_ERR_IO = -1, ORA_ERR_FORMAT = -2, ORA_ERR_CURVE = -3, ORA_ERR_BUFFER = -4 };
enum { ORA_OP_ADD = 0, ORA_OP_SUB, ORA_OP_NEG, ORA_OP_MUL, ORA_OP_SQR, ORA_OP_TOMONT, ORA_OP_FROMMONT, ORA_OP_INV };
enum { ORA_PT_ADD = 0, ORA_PT_MADD, ORA_PT_DBL, ORA_PT_NEG };

/* field: 0 = Fq, 1 = Fr.  Elements are 4 x u64 little-endian limbs. */
void ora_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r);
void ora_field_op_vec(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, uint64_t n);
int  ora_fe_to_dec(int field, const uint64_t* a, char* out);
void ora_fe_from_dec(int field, const char* s, uint64_t* r);
void ora_fq2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* r);

/* group: 0 = G1 (affine 64 B, XYZZ 128 B), 1 = G2 (affine 128 B, XYZZ 256 B) */
void ora_pt_op(int group, int op, const void* p1, const void* p2, void* r);
void ora_pt_to_affine(int group, const void* p, void* out_aff);
int  ora_pt_eq(int group, const void* p1, const void* p2);
void ora_generator(int group, void* out_aff);
void ora_mul_scalar(int group, const void* base_aff, const uint8_t* scalar, unsigned scalar_size, void* out_xyzz);
void ora_gen_points(int group, uint64_t start, uint64_t n, void* out_aff);
void ora_msm(int group, const void* bases_aff, const uint8_t* scalars, uint64_t scalar_size, uint64_t n, int nthreads,
             void* out_xyzz, void* out_aff);

/* pairing / Groth16 verification (oracle/pairing_ref.h): Fq12 values are 12 x 32 B (c0.c0.a, c0.c0.b, c0.c1.a, ...), Montgomery */
void ora_miller(const void* g1_aff, const void* g2_aff, void* out_f);
int  ora_pairing(const void* g1_aff, const void* g2_aff, void* out_gt);
int  ora_final_exp(const void* f_in, void* out_gt);
void ora_gt_mul(const void* a, const void* b, void* out);
int  ora_groth16_verify(const void* alpha1, const void* beta2, const void* gamma2, const void* delta2, const void* ic,
                        uint32_t n_ic, const void* proof, const uint8_t* inputs);

int ora_ntt(uint64_t* a, uint64_t n, uint64_t max_domain, int inverse);
int ora_ntt_root(uint64_t max_domain, unsigned domain_pow, uint64_t idx, uint64_t* out);

int ora_zkey_info(const char* zkey_path, uint32_t* n_vars, uint32_t* n_public, uint32_t* domain_size,
                  uint64_t* n_coefs);
int ora_prove_files(const char* zkey_path, const char* wtns_path, const uint8_t r_std[32], const uint8_t s_std[32],
                    int nthreads, char* out_json, size_t cap, uint64_t* h_scalars_out);

#ifdef __cplusplus
}
#endif
#endif
