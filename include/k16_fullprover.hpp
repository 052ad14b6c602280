// k16_fullprover.hpp -- the C++ prover facade the Aptos prover service binds to, backed by the
// MI355X HIP path (libk16.so).
//
// Drop-in for rust-rapidsnark/rapidsnark/src/fullprover.hpp of aptos-labs/keyless-zk-proofs: the
// same global enum / struct / class names, the same member order and types (so the object layouts
// bindgen derives in rust-rapidsnark/build.rs:142-146 are unchanged: FullProver = { impl pointer,
// state }, ProverResponse = { type, raw_json, error, metrics }), the same Itanium-mangled entry
// points:
//      FullProver::FullProver(const char* zkey_path)            fullprover.cpp:80-101
//      FullProver::~FullProver()                                fullprover.cpp:103-109
//      ProverResponse FullProver::prove(const char* wtns) const fullprover.cpp:114-125 -> :204-250
//      ProverResponse::ProverResponse(ProverError)              fullprover.cpp:183-189
//      ProverResponse::ProverResponse(const char*, ProverResponseMetrics)   :191-198
//      ProverResponse::~ProverResponse()                        fullprover.cpp:252-260
//
// Behavioural contract kept from the reference:
//   * no exception ever crosses this boundary; constructor failures land in `state`
//     (open/mmap failure -> ZKEY_FILE_LOAD_ERROR, wrong container / protocol / curve ->
//     UNSUPPORTED_ZKEY_CURVE), and prove() on a prover that is not OK returns PROVER_NOT_READY
//   * raw_json of a SUCCESS response is a malloc'ed, NUL-terminated compact JSON string
//     (`{"pi_a":[..],"pi_b":[..],"pi_c":[..],"protocol":"groth16"}`) freed by ~ProverResponse;
//     error responses point at a static empty string
//   * metrics.prover_time is wall-clock milliseconds around the proof computation
//   * prove() is never called concurrently on one object (the service serialises with a mutex,
//     prover-service/src/request_handler/prover_state.rs:21)
// Deliberate differences:
//   * a malformed .wtns file yields ProverError::INVALID_INPUT instead of an exception escaping
//     (the reference lets std::range_error propagate across the FFI, binfile_utils.cpp:25-37)
//   * a witness with fewer values than the zkey has wires is INVALID_INPUT (the reference reads
//     past the mapping)
//   * no GPU / HIP failure: the constructor reports ZKEY_FILE_LOAD_ERROR and logs the reason on
//     stderr; there is no CPU fallback
//   * the reference's stdout log lines are emitted only when K16_LOG=1
//   * K16_DEVICES=<comma-separated ordinals> puts one prover per listed device (repeats allowed) behind this one
//     object; prove() then takes a free one and MAY be called concurrently.  Without it the object
//     owns a single prover on K16_DEVICE and concurrent callers simply queue.
#pragma once

class FullProverImpl;

enum ProverResponseType
{
    SUCCESS,
    ERROR
};

enum FullProverState
{
    OK,
    ZKEY_FILE_LOAD_ERROR,
    UNSUPPORTED_ZKEY_CURVE
};

enum ProverError
{
    NONE,
    PROVER_NOT_READY,
    INVALID_INPUT,
    WITNESS_GENERATION_INVALID_CURVE
};

struct ProverResponseMetrics
{
    int prover_time; // milliseconds
};

struct ProverResponse
{
    ProverResponseType    type;
    char const*           raw_json;
    ProverError           error;
    ProverResponseMetrics metrics;

private:
    static char const* const empty_string;

public:
    ProverResponse(ProverError _error);
    ProverResponse(const char* _raw_json, ProverResponseMetrics _metrics);

    ProverResponse()                                 = delete;
    ProverResponse(ProverResponse const&)            = delete;
    ProverResponse& operator=(ProverResponse const&) = delete;

    ~ProverResponse();
};

class FullProver
{
    FullProverImpl* impl;
    FullProverState state;

public:
    FullProver() = delete;
    FullProver(const char* _zkeyFileName);
    ~FullProver();
    ProverResponse prove(const char* input) const;
};
