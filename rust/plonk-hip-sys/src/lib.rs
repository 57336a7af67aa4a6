//! Raw `extern "C"` declarations of `include/plonk_hip.h` — the entry points behind the seven reference
//! functions.  Elements are 4 little-endian `u64` Montgomery limbs (R = 2^256), the memory of ark-ff 0.3
//! `Fp256`; sigma columns are `u64` row indices; `sel[k]` is indexed by the `ASTKind` discriminant, null =
//! column absent.  Not compiled in this repository's environment (no Rust toolchain there).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_longlong, c_void};

#[repr(C)]
pub struct plk_ctx {
    _private: [u8; 0],
}

pub const PLK_OK: c_int = 0;
pub const PLK_ERR_LENGTH: c_int = -1;
pub const PLK_ERR_NOT_POW2: c_int = -2;
pub const PLK_ERR_TWO_ADICITY: c_int = -3;
pub const PLK_ERR_ZERO_DENOMINATOR: c_int = -4;
pub const PLK_ERR_SIGMA_RANGE: c_int = -5;
pub const PLK_ERR_HIP: c_int = -6;
pub const PLK_ERR_ARG: c_int = -7;
pub const PLK_ERR_NOMEM: c_int = -8;

pub const PLK_FIELD_BLS12_381_FR: c_int = 0;
pub const PLK_FIELD_TWEEDLE_FR: c_int = 1;
pub const PLK_NUM_FIELDS: usize = 2;

pub const PLK_NUM_WIRES: usize = 4;
pub const PLK_NUM_SELECTORS: usize = 9;

/// Opaque node (G ranks in one process), created by `plk_node_create`.
#[repr(C)]
pub struct plk_node {
    _private: [u8; 0],
}
pub const PLK_NODE_RCCL: c_int = 0;
pub const PLK_NODE_COPY: c_int = 1;

/// `plk_circuit_io` of plonk_hip.h: one circuit of `plk_node_round23_batch` (device addresses).
#[repr(C)]
pub struct plk_circuit_io {
    pub wires: [*const u64; 4],
    pub sel: [*const u64; 9],
    pub sigma: [*const u64; 4],
    pub beta: [u64; 4],
    pub gamma: [u64; 4],
    pub z_out: *mut u64,
    pub gate_out: *mut u64,
    pub perm_out: *mut u64,
}

/// `plk_builder_fn`: fills up to `capacity` elements, returns how many evaluations the builder produced.
pub type plk_builder_fn = unsafe extern "C" fn(user: *mut c_void, evals_out: *mut u64, capacity: usize) -> usize;

extern "C" {
    pub fn plk_ctx_create(device: c_int, out: *mut *mut plk_ctx) -> c_int;
    pub fn plk_ctx_destroy(ctx: *mut plk_ctx);
    pub fn plk_strerror(code: c_int) -> *const c_char;
    pub fn plk_last_error(ctx: *const plk_ctx) -> *const c_char;
    pub fn plk_last_error_row(ctx: *const plk_ctx) -> c_longlong;
    pub fn plk_field_constants(modulus: *mut u64, one: *mut u64, r2: *mut u64);
    pub fn plk_host_alloc(ctx: *mut plk_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn plk_host_free(ctx: *mut plk_ctx, p: *mut c_void);

    /// src/polynomial_utils.rs:18-33
    pub fn plk_intt(ctx: *mut plk_ctx, evals: *const u64, evals_len: usize, domain_size: usize, coeffs_out: *mut u64) -> c_int;
    /// src/polynomial_utils.rs:47-62
    pub fn plk_build_and_convert(ctx: *mut plk_ctx, domain_size: usize, builder: plk_builder_fn, user: *mut c_void,
                                 coeffs_out: *mut u64) -> c_int;
    /// DensePolynomial::evaluate_over_domain_by_ref as called at src/execution_trace.rs:362-370
    pub fn plk_ntt(ctx: *mut plk_ctx, coeffs: *const u64, n_coeffs: usize, domain_size: usize, evals_out: *mut u64) -> c_int;
    /// src/execution_trace.rs:318-344
    pub fn plk_gate_polys(ctx: *mut plk_ctx, wires: *const *const u64, sel: *const *const u64, n: usize, domain_size: usize,
                          wires_out: *const *mut u64, sel_out: *const *mut u64) -> c_int;
    /// src/execution_trace.rs:357-431
    pub fn plk_gate_constraints(ctx: *mut plk_ctx, wire_coeffs: *const *const u64, sel_coeffs: *const *const u64,
                                n_coeffs: usize, big_n: usize, out: *mut u64) -> c_int;
    /// src/permutation_polynomial.rs:217-261; z_out receives n + 1 elements
    pub fn plk_grand_product(ctx: *mut plk_ctx, wires: *const *const u64, sigma: *const *const u64, n: usize,
                             beta: *const u64, gamma: *const u64, z_out: *mut u64) -> c_int;
    /// src/permutation_polynomial.rs:272-298 (z[0..n) interpolated, z[n] through z_last_out)
    pub fn plk_perm_polys(ctx: *mut plk_ctx, wires: *const *const u64, sigma: *const *const u64, n: usize, domain_size: usize,
                          beta: *const u64, gamma: *const u64, sigma_coeffs_out: *const *mut u64, z_coeffs_out: *mut u64,
                          z_last_out: *mut u64) -> c_int;
    /// src/permutation_polynomial.rs:312-376; z_shift = 1 is the reference
    pub fn plk_perm_constraints(ctx: *mut plk_ctx, z_coeffs: *const u64, sigma_coeffs: *const *const u64,
                                wire_coeffs: *const *const u64, n_coeffs: usize, big_n: usize, beta: *const u64,
                                gamma: *const u64, z_shift: usize, out: *mut u64) -> c_int;
    /// rounds 2+3 in one device-resident call (no counterpart in the reference)
    pub fn plk_round23(ctx: *mut plk_ctx, wires: *const *const u64, sel: *const *const u64, sigma: *const *const u64,
                       n: usize, big_n: usize, beta: *const u64, gamma: *const u64, z_shift: usize,
                       z_out: *mut u64, gate_out: *mut u64, perm_out: *mut u64) -> c_int;

    // ---- fields: a context computes in the field it was created for (plk_ctx_create = field 0)
    pub fn plk_ctx_create_field(device: c_int, field_id: c_int, out: *mut *mut plk_ctx) -> c_int;
    pub fn plk_ctx_field(ctx: *const plk_ctx) -> c_int;
    pub fn plk_field_name(field_id: c_int) -> *const c_char;
    pub fn plk_field_constants_of(field_id: c_int, modulus: *mut u64, one: *mut u64, r2: *mut u64) -> c_int;
    pub fn plk_node_create_field(devices: *const c_int, ranks: usize, exchange: c_int, field_id: c_int, out: *mut *mut plk_node) -> c_int;

    // ---- device memory for a host without a HIP binding (no counterpart in the reference) ----
    pub fn plk_dev_alloc(ctx: *mut plk_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn plk_dev_free(ctx: *mut plk_ctx, p: *mut c_void);
    pub fn plk_dev_upload(ctx: *mut plk_ctx, stream: *mut c_void, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn plk_dev_download(ctx: *mut plk_ctx, stream: *mut c_void, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;
    pub fn plk_ctx_stream(ctx: *mut plk_ctx) -> *mut c_void;
    /// rounds 2+3 on columns resident on the context's GPU, enqueued on `stream`
    pub fn plk_dev_round23(ctx: *mut plk_ctx, stream: *mut c_void, wires: *const *const u64, sel: *const *const u64,
                           sigma: *const *const u64, n: usize, big_n: usize, beta: *const u64, gamma: *const u64, z_shift: usize,
                           z_out: *mut u64, gate_out: *mut u64, perm_out: *mut u64) -> c_int;
    pub fn plk_dev_status(ctx: *mut plk_ctx, stream: *mut c_void) -> c_int;

    // ---- one node: G ranks in one process, one GPU each (no counterpart in the reference) ----
    /// `exchange`: PLK_NODE_RCCL (one GPU per rank) or PLK_NODE_COPY (device-to-device copies; ranks may share a GPU)
    pub fn plk_node_create(devices: *const c_int, ranks: usize, exchange: c_int, out: *mut *mut plk_node) -> c_int;
    pub fn plk_node_destroy(node: *mut plk_node);
    pub fn plk_node_ranks(node: *const plk_node) -> usize;
    pub fn plk_node_ctx(node: *mut plk_node, rank: usize) -> *mut plk_ctx;
    pub fn plk_node_last_error(node: *const plk_node) -> *const c_char;
    /// one transform of `big_n` points over the node's ranks (four-step form, three all-to-all exchanges);
    /// in_shards[g] / out_shards[g]: big_n / ranks elements on rank g's GPU
    pub fn plk_node_ntt(node: *mut plk_node, in_shards: *const *const u64, big_n: usize, inverse: c_int, natural_output: c_int,
                        out_shards: *const *mut u64) -> c_int;
    /// the same on host arrays: src/polynomial_utils.rs:29 / src/execution_trace.rs:362-370 at sizes sharded over the node
    pub fn plk_node_ntt_host(node: *mut plk_node, input: *const u64, big_n: usize, inverse: c_int, out: *mut u64) -> c_int;
    pub fn plk_node_sync(node: *mut plk_node) -> c_int;
    /// batched proofs: circuit i on rank i % ranks, all its pointers in that rank's GPU memory
    pub fn plk_node_round23_batch(node: *mut plk_node, circuits: *const plk_circuit_io, count: usize, n: usize, big_n: usize,
                                  z_shift: usize) -> c_int;
    pub fn plk_node_round23(node: *mut plk_node, per_rank: *const plk_circuit_io, n: usize, big_n: usize, z_shift: usize) -> c_int;
    pub fn plk_node_round23_host(node: *mut plk_node, wires: *const *const u64, sel: *const *const u64, sigma: *const *const u64, n: usize,
                                 big_n: usize, beta: *const u64, gamma: *const u64, z_shift: usize, z_out: *mut u64, gate_out: *mut u64,
                                 perm_out: *mut u64) -> c_int;
}
