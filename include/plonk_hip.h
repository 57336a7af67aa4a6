/* plonk_hip.h — C ABI of the MI355X (gfx950) implementation of the PLONK polynomial hot path.
 *
 * Every entry point replaces one generic Rust function of wu-s-john/PLONK (cited per
 * declaration, paths relative to the reference tree).  The reference has no FFI of its own;
 * a Rust caller binds these symbols from an `extern "C"` block and keeps its signatures.
 *
 * Conventions
 *  - Field: the field of the context (BLS12-381 scalar field unless created otherwise, see "fields").  An element is
 *    4 little-endian uint64_t limbs in Montgomery form with R = 2^256 — byte-for-byte the memory of ark-ff 0.3
 *    `Fp256<P>` (`x.0.0`).  Columns are contiguous arrays of elements (n * 4 uint64_t).
 *  - sigma columns are arrays of uint64_t row indices (`PermutationTable<usize>`,
 *    src/permutation_polynomial.rs:51-61).
 *  - `sel[k]` / `sel_out[k]` are indexed by `ASTKind` declaration order
 *    (Int, Bool, Add, Sub, Mult, Div, Eq, Not, If — src/ast.rs:56-66); NULL = column absent.
 *  - Domains are the power-of-two multiplicative subgroups ark-poly's `GeneralEvaluationDomain`
 *    builds (`Radix2EvaluationDomain`); only the size crosses the ABI.
 *  - Return value: PLK_OK or a negative code; each code stands for one reference panic.
 *  - Host entry points (`plk_*`) take caller-owned host memory, are synchronous and never keep a
 *    pointer.  Device entry points (`plk_dev_*`) take device memory of the context's GPU and
 *    enqueue on the given hipStream_t (passed as void*); data-dependent failures (zero
 *    denominator, sigma out of range) are reported by plk_dev_status().
 *  - One context per GPU; a context must not be used from two threads at once.  The context's workspace, status
 *    word and side stream are shared by all its calls, so the library orders them: every device call ends by
 *    recording an event on the stream it was given, and a call given another stream than the context's previous
 *    call first makes that stream wait (hipStreamWaitEvent) for that event.  Calls on one context therefore never
 *    overlap on the GPU; use one context per stream for overlap (except plk_dev_grand_product_noalloc, whose fixed
 *    array is shared per GPU).
 *  - Stream lifetime: a stream handle is used only inside the call that brings it; the library keeps no handle
 *    of the caller's.  A stream may be destroyed as soon as the work enqueued on it has finished (per-request
 *    streams are fine); destroying it with library work still pending is the caller's race, as with any HIP work.
 *  - HIP graph capture: a call on a stream that is being captured takes no part in the ordering above (an event
 *    recorded into a capture cannot be waited for from outside it): its launches are ordered by the graph, and the
 *    caller orders the graph's replays against other calls on the context.  Only entry points that neither
 *    allocate nor synchronise are valid under capture: plk_dev_grand_product_noalloc, and the other plk_dev_*
 *    calls once their workspace and tables exist from an earlier identical call (not plk_dev_status, not the msm
 *    calls, which wait for the stream).
 *  - Preconditions on data, not checked: every element of every column is reduced (< p), as ark-ff guarantees
 *    for `Fp256`; an unreduced limb pattern gives a wrong element, not an error.  beta, gamma, shift, alpha and
 *    default_val ARE checked (PLK_ERR_ARG).
 *  - sigma range: the reference indexes each wire column with its own sigma column
 *    (src/permutation_polynomial.rs:235-238), so with columns of unequal length an index is legal up to that
 *    column's length; here all four columns have the n rows of `input1` and any index >= n is
 *    PLK_ERR_SIGMA_RANGE.
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plk_ctx plk_ctx;

enum {
  PLK_OK = 0,
  PLK_ERR_LENGTH = -1,           /* assert_eq!(evals.len(), domain.size()) — src/polynomial_utils.rs:22-26, :56-60 */
  PLK_ERR_NOT_POW2 = -2,         /* domain size is not a power of two (Radix2EvaluationDomain only) */
  PLK_ERR_TWO_ADICITY = -3,      /* log2(domain size) exceeds the field's two-adicity (32 for BLS12-381 Fr) */
  PLK_ERR_ZERO_DENOMINATOR = -4, /* `denominator.inverse().unwrap()` — src/permutation_polynomial.rs:258 */
  PLK_ERR_SIGMA_RANGE = -5,      /* slice index out of bounds — src/permutation_polynomial.rs:235-238 */
  PLK_ERR_HIP = -6,              /* HIP runtime failure (see plk_last_error) */
  PLK_ERR_ARG = -7,              /* NULL where a pointer is required, z_shift >= N, batch too large ... */
  PLK_ERR_NOMEM = -8             /* device or host allocation failed */
};

#define PLK_NUM_WIRES 4
#define PLK_NUM_SELECTORS 9
#define PLK_MAX_BATCH 24
#define PLK_INPUT_FIELD 0u  /* column of field elements */
#define PLK_INPUT_U64 1u    /* column of uint64_t indices, converted with F::from(u64) on load */

/* ---- fields ------------------------------------------------------------------------------ */

/* The reference's functions are generic over `F: FftField`; the one field it names, `ark_tweedle::Fr`
 * (src/main.rs:3), comes from a crate its manifest does not list.  The library is instantiated for:
 *   PLK_FIELD_BLS12_381_FR  the BLS12-381 scalar field (two-adicity 32, generator 7) - the default, the field every
 *                           figure and every parity statement of this library is made for; its root of unity is
 *                           the constant arkworks publishes;
 *   PLK_FIELD_TWEEDLE_FR    the scalar field of Tweedledee, 2^254 + 4707489544292117082687961190295928833
 *                           (two-adicity 34).  Its generator (5) and root of unity are SELF-DERIVED
 *                           (tools/derive_field_constants.py): no arkworks source was available to compare with, so
 *                           transforms agree with ark-poly only if ark-ff's Tweedle parameters use the same generator.
 * A context is created for one field; every call on it computes in that field.  The commitment entry points
 * (plk_msm_*, plk_dev_msm_*, plk_dev_g1_multiples) exist for BLS12-381 only and return PLK_ERR_ARG otherwise. */
enum { PLK_FIELD_BLS12_381_FR = 0, PLK_FIELD_TWEEDLE_FR = 1, PLK_NUM_FIELDS = 2 };

/* ---- context ---------------------------------------------------------------------------- */

/* plk_ctx_create = plk_ctx_create_field with PLK_FIELD_BLS12_381_FR */
int plk_ctx_create_field(int device, int field_id, plk_ctx** out);
/* the field a context was created for; its name ("BLS12-381 Fr", "Tweedle Fr"; NULL for an unknown id) */
int plk_ctx_field(const plk_ctx* ctx);
const char* plk_field_name(int field_id);
/* plk_field_constants for any field: modulus, R mod p, R^2 mod p (R = 2^256), any pointer may be NULL;
 * returns PLK_ERR_ARG for an unknown id */
int plk_field_constants_of(int field_id, uint64_t modulus[4], uint64_t one[4], uint64_t r2[4]);
int plk_ctx_create(int device, plk_ctx** out);
void plk_ctx_destroy(plk_ctx* ctx);
const char* plk_strerror(int code);
/* text of the last failure on this context ("" if none) */
const char* plk_last_error(const plk_ctx* ctx);
/* row reported by the last PLK_ERR_ZERO_DENOMINATOR / PLK_ERR_SIGMA_RANGE, or -1 */
long long plk_last_error_row(const plk_ctx* ctx);
/* field constants, 4 limbs each: modulus (plain integer), one (= R mod p), R^2 mod p */
void plk_field_constants(uint64_t modulus[4], uint64_t one[4], uint64_t r2[4]);

/* Page-locked host memory for the columns passed to the host entry points.  They accept any host
 * pointer; buffers from plk_host_alloc are copied to and from the GPU by DMA at PCIe speed.  Ordinary
 * (pageable) buffers of 8 MiB and more are copied by a pool of up to ten threads that the context starts
 * at the first such copy and plk_ctx_destroy joins (each stages 4 MiB chunks through page-locked slots of
 * its own): about 1.3x the time of the page-locked path for a whole plk_round23 call, where the runtime's
 * own staging made it 2.3x.  No counterpart in the reference (a Rust caller would back its Vec<F> with
 * plk_host_alloc through a custom allocator). */
int plk_host_alloc(plk_ctx* ctx, size_t bytes, void** out);
void plk_host_free(plk_ctx* ctx, void* p);

/* ---- host entry points: the reference's seven functions ---------------------------------- */

/* evaluations_to_dense_polynomial — src/polynomial_utils.rs:18-33.
 * Inverse DFT over the size-`domain_size` subgroup, natural order, 1/n included.
 * Writes all `domain_size` coefficients (DensePolynomial::from_coefficients_vec would only
 * drop trailing zero coefficients). PLK_ERR_LENGTH if evals_len != domain_size. */
int plk_intt(plk_ctx* ctx, const uint64_t* evals, size_t evals_len, size_t domain_size, uint64_t* coeffs_out);

/* build_and_convert — src/polynomial_utils.rs:47-62.  `builder` fills up to `capacity`
 * elements and returns how many evaluations it produced (may exceed capacity to report an
 * over-long vector); the count is checked against domain_size as at :56-60. */
typedef size_t (*plk_builder_fn)(void* user, uint64_t* evals_out, size_t capacity);
int plk_build_and_convert(plk_ctx* ctx, size_t domain_size, plk_builder_fn builder, void* user, uint64_t* coeffs_out);

/* DensePolynomial::evaluate_over_domain_by_ref as called at src/execution_trace.rs:362-370 and
 * src/permutation_polynomial.rs:323-334: zero-pad to domain_size, forward DFT. */
int plk_ntt(plk_ctx* ctx, const uint64_t* coeffs, size_t n_coeffs, size_t domain_size, uint64_t* evals_out);

/* gate_evaluation_table_to_gate_polynomial — src/execution_trace.rs:318-344. */
int plk_gate_polys(plk_ctx* ctx, const uint64_t* const wires[PLK_NUM_WIRES], const uint64_t* const sel[PLK_NUM_SELECTORS],
                   size_t n, size_t domain_size,
                   uint64_t* const wires_out[PLK_NUM_WIRES], uint64_t* const sel_out[PLK_NUM_SELECTORS]);

/* build_gate_constraint_evaluation_polynomial — src/execution_trace.rs:357-431.
 * Coefficient vectors of n_coeffs elements each, evaluated on the size-N domain. */
int plk_gate_constraints(plk_ctx* ctx, const uint64_t* const wire_coeffs[PLK_NUM_WIRES],
                         const uint64_t* const sel_coeffs[PLK_NUM_SELECTORS], size_t n_coeffs, size_t N, uint64_t* out);

/* build_grand_product_evaluation_polynomial — src/permutation_polynomial.rs:217-261.
 * z_out receives n + 1 elements. */
int plk_grand_product(plk_ctx* ctx, const uint64_t* const wires[PLK_NUM_WIRES], const uint64_t* const sigma[PLK_NUM_WIRES],
                      size_t n, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out);

/* build_permutation_polynomials — src/permutation_polynomial.rs:272-298.
 * The reference interpolates the n+1 grand-product values on a size-n domain and therefore
 * panics on its own length assert whenever n == domain_size; this entry point interpolates
 * z[0..n) and returns z[n] through z_last_out (may be NULL). Requires n == domain_size. */
int plk_perm_polys(plk_ctx* ctx, const uint64_t* const wires[PLK_NUM_WIRES], const uint64_t* const sigma[PLK_NUM_WIRES],
                   size_t n, size_t domain_size, const uint64_t beta[4], const uint64_t gamma[4],
                   uint64_t* const sigma_coeffs_out[PLK_NUM_WIRES], uint64_t* z_coeffs_out, uint64_t* z_last_out);

/* build_permutation_constraint_evaluation_polynomial — src/permutation_polynomial.rs:312-376.
 * z_shift = 1 reproduces the reference (`z_evals[(i + 1) % n]`, :347). */
int plk_perm_constraints(plk_ctx* ctx, const uint64_t* z_coeffs, const uint64_t* const sigma_coeffs[PLK_NUM_WIRES],
                         const uint64_t* const wire_coeffs[PLK_NUM_WIRES], size_t n_coeffs, size_t N,
                         const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* out);

/* Rounds 2+3 in one call, device-resident between the steps: the composition
 * gate polys (:318) -> grand product (:217) -> sigma/Z interpolation (:272) ->
 * gate constraints (:357) -> permutation constraints (:312), wires evaluated once.
 * z_out: n+1 elements; gate_out, perm_out: N elements.  Any output may be NULL. */
int plk_round23(plk_ctx* ctx, const uint64_t* const wires[PLK_NUM_WIRES], const uint64_t* const sel[PLK_NUM_SELECTORS],
                const uint64_t* const sigma[PLK_NUM_WIRES], size_t n, size_t N,
                const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift,
                uint64_t* z_out, uint64_t* gate_out, uint64_t* perm_out);

/* ---- the step before the path: the permutation table ------------------------------------- */

/* build_permutation_trace + build_permutation_table — src/permutation_polynomial.rs:18-39, :79-202,
 * over group_wire_positions_by_node_id (src/execution_trace.rs:191-197) and OffsetTable::to_index
 * (src/offset_table.rs:61-64), with the trace flattened to one node id per wire and global row:
 * node_ids[w][j] = `node_id` of the value on wire w (0, 1, 2 = inputs, 3 = output) of global row j
 * (rows numbered as the offset table numbers them), or PLK_NO_WIRE where the row's operation does
 * not use that wire.  All positions carrying one node id are linked into a cycle in ascending
 * (row, wire) order — the order collect_wire_positions (src/execution_trace.rs:109-187) produces —
 * and sigma_out[w][j] = the global ROW of the position after (j, w) in its cycle (the wire of that
 * next position is dropped, as to_index drops it); a node id seen once, and an unused wire, map to
 * their own row (:24-27, :121-122, :157).  Any n >= 0, not only powers of two. */
#define PLK_NO_WIRE UINT64_MAX
int plk_sigma_table(plk_ctx* ctx, const uint64_t* const node_ids[PLK_NUM_WIRES], size_t n,
                    uint64_t* const sigma_out[PLK_NUM_WIRES]);

/* build_gate_evaluation_table — src/execution_trace.rs:248-302, completed: the reference handles
 * the single-input tables only (:268-296, the rest "omitted", :298-299) and writes into vectors of
 * length zero (:238-241), so it has no output to compare with; this follows the pattern it shows.
 * The trace is flattened as for plk_sigma_table, plus
 *   row_ops[j]  = PLK_ROW_OP(kind, inputs): the `ASTKind` index of row j's operation and its number
 *                 of inputs (1, 2 or 3 = the table map it is filed in, src/execution_trace.rs:38-42);
 *   node_values = the evaluated value of every node (NodeMeta::evaluated_value, src/ast.rs:6-9),
 *                 n_values elements; this is the wire map: wire_map[(j, w)] = node_values[node_ids[w][j]],
 *                 and a position whose id is PLK_NO_WIRE or >= n_values is absent from it.
 * wires_out[w][j] = the mapped value if wire w is an input the operation has (w < inputs) or the
 * output (w = 3) and the position is in the map, else default_val (:260-265, :287-288);
 * sel_out[k][j] = one if kind == k else zero, for every k with sel_out[k] != NULL (:292-294). */
#define PLK_ROW_OP(kind, inputs) ((uint8_t)((kind) | ((inputs) << 4)))
int plk_gate_table_fill(plk_ctx* ctx, const uint8_t* row_ops, const uint64_t* const node_ids[PLK_NUM_WIRES], size_t n,
                        const uint64_t* node_values, size_t n_values, const uint64_t default_val[4],
                        uint64_t* const wires_out[PLK_NUM_WIRES], uint64_t* const sel_out[PLK_NUM_SELECTORS]);

/* ---- the step after the path: coset transforms and the quotient --------------------------- */

/* The reference stops at the two numerator vectors (README.md:12-14 done, :15-17 next) and has no
 * code for what follows; these entry points are ark-poly 0.3's coset transform pair and the textbook
 * division by the vanishing polynomial, for the caller that goes on to the quotient polynomial.
 *
 * plk_coset_ntt: inverse == 0: EvaluationDomain::coset_fft - the n_in coefficients (zero-padded) are
 * evaluated at shift * omega^i, i < domain_size; inverse != 0: coset_ifft - domain_size evaluations
 * at those points back to domain_size coefficients (n_in must equal domain_size).  ark-poly 0.3
 * fixes shift = F::multiplicative_generator() (7 for this field); here it is a parameter.
 *
 * plk_quotient: numerators[i], i < count <= 8, are evaluation vectors (N elements each) on the coset
 * shift * <omega_N>; writes the N coefficients of (sum_i alpha^i numerators[i]) / Z_H with
 * Z_H(x) = x^n - 1, N / n <= 16.  PLK_ERR_ZERO_DENOMINATOR if Z_H vanishes on the coset (e.g.
 * shift = 1, the reference's own evaluation domain). */
int plk_coset_ntt(plk_ctx* ctx, const uint64_t* in, size_t n_in, size_t domain_size, const uint64_t shift[4], int inverse,
                  uint64_t* out);
int plk_quotient(plk_ctx* ctx, const uint64_t* const* numerators, size_t count, const uint64_t alpha[4], size_t n, size_t N,
                 const uint64_t shift[4], uint64_t* coeffs_out);

/* ---- commitments: multi-scalar multiplication in BLS12-381 G1 ------------------------------ */

/* The reference imports `ark_poly_commit::kzg10` (src/main.rs:9) and stops there (README.md:15-17:
 * "Implement commitment scheme" unchecked).  A KZG10 commitment of a polynomial is
 * sum_i coeff_i * powers_of_g[i] (`KZG10::commit` -> `VariableBaseMSM::multi_scalar_mul`); these entry
 * points compute that sum.
 *  - A G1 point crosses the ABI in affine form as 12 uint64_t: x then y, each the memory of ark-ff
 *    `Fp384` (6 little-endian Montgomery limbs, R = 2^384); (0, 0) stands for the point at infinity.
 *  - plk_msm_set_bases takes n affine bases (the powers of tau) once and builds from them the table
 *    2^(c w) * base_i (c ~ log2 n bits per window, ceil(255 / c) windows, 128 bytes per entry: 2.1 GB for 2^20
 *    bases) that stays with the context; plk_dev_msm_set_bases does the same from a device array, which the
 *    caller may free afterwards.  Both wait for the device.
 *  - plk_msm_g1 / plk_dev_msm_g1: scalars are n field elements in the usual form (Montgomery Fr), on the
 *    host / on the device; n must not exceed the number of bases.  The affine result is written to host
 *    memory; the call waits for the stream (the last ~300 point operations are serial and run on the host).
 *  - plk_dev_g1_multiples writes (first + i) * base, i < count, as affine points to device memory
 *    (test and benchmark set-up: a table whose discrete logarithms are known).
 * The window width follows the number of BASES, so a call with far fewer scalars than bases is correct but
 * slower than after a set_bases of that size.  Every bucket gets lanes in proportion to the number of digits that
 * fall into it, so scalars that crowd into a few buckets (all equal, all small) cost about the same as random ones. */
int plk_msm_set_bases(plk_ctx* ctx, const uint64_t* bases_host, size_t n);
int plk_msm_g1(plk_ctx* ctx, const uint64_t* scalars_host, size_t n, uint64_t out_xy[12]);
/* `count` sums over the same bases, scalars[k] holding n elements each: the commitments of one prover round (the
 * four wire polynomials, the parts of the quotient); results at out_xy[12 k ...].  The sums alternate between two
 * streams, so that one sum's latency-bound bucket reduction runs beside the next one's additions. */
int plk_msm_g1_batch(plk_ctx* ctx, const uint64_t* const scalars_host[], size_t count, size_t n, uint64_t* out_xy);
/* The sum of `count` affine points (host memory, host arithmetic, no GPU and no context needed): what adds up the
 * partial sums when the terms of one multi-scalar multiplication are dealt to several GPUs.  Returns PLK_OK or
 * PLK_ERR_ARG (null pointer). */
int plk_g1_sum(const uint64_t* points_xy, size_t count, uint64_t out_xy[12]);

/* ---- device entry points (same arithmetic, device pointers, asynchronous) ---------------- */

/* `batch` transforms of size N in one set of launches.  in[b] holds n_in valid elements
 * (zero-padded to N), of kind kinds[b] (PLK_INPUT_*; kinds == NULL means all field).
 * inverse != 0: inverse DFT including 1/N.  out[b] may alias in[b]. */
int plk_dev_ntt_batch(plk_ctx* ctx, void* stream, const uint64_t* const* in, const uint32_t* kinds, size_t batch,
                      size_t n_in, size_t N, int inverse, uint64_t* const* out);

/* Interpolation + evaluation on the 4n-point domain in one call, for `batch` columns of n rows:
 * evaluations_to_dense_polynomial (src/polynomial_utils.rs:18-33) followed by
 * evaluate_over_domain_by_ref on the domain of size 4n (src/execution_trace.rs:362-370), natural
 * order.  The 4n-point domain is four cosets of the trace domain: the coset of the trace domain
 * itself repeats the input column, the other three are n-point transforms of prescaled
 * coefficients - the same values as plk_dev_ntt_batch(coeffs, n, 4n) for about 3/4 of the work.
 * coef_out may be NULL, or hold NULL entries, where the coefficients are not wanted.
 * Output layout (for splitting the 4n points over GPUs): with chunk != 0, point i is written at
 * eval_out[b] + (i / chunk) * chunk_stride + i % chunk, and every chunk is followed by copies of the
 * first `halo` points of the next chunk, cyclically (chunk divides 4n, chunk_stride >= chunk + halo);
 * chunk = 0 writes the plain 4n-element vector. */
int plk_dev_extend4(plk_ctx* ctx, void* stream, const uint64_t* const* cols, const uint32_t* kinds, size_t batch, size_t n,
                    uint64_t* const* coef_out, uint64_t* const* eval_out, size_t chunk, size_t chunk_stride, size_t halo);

/* Same transform over `count` equally spaced columns: column b is read at in + b * in_stride and
 * written at out + b * out_stride (strides in elements, >= n_in and >= N).  This is the shape of
 * the row / column phases of a four-step transform on a matrix held in HBM. */
int plk_dev_ntt_strided(plk_ctx* ctx, void* stream, const uint64_t* in, size_t in_stride, uint32_t kind, size_t count,
                        size_t n_in, size_t N, int inverse, uint64_t* out, size_t out_stride);

/* The two local phases of ONE transform of N points sharded over `ranks` GPUs (a power of two), rank `rank` holding the
 * inputs [rank * N / ranks, (rank + 1) * N / ranks) (no counterpart in the reference: ark-poly runs one shared-memory
 * FFT).  N = N1 * N2 with N1 = 2^ceil(log2(N) / 2), n = n1 * N2 + n2, k = k1 + N1 * k2, r1 = N1 / ranks, c2 = N2 / ranks.
 * Both phases read the buffer exactly as an all-to-all delivers it and write exactly what the next all-to-all sends:
 *   exchange 1 (caller): rank g sends its rows' columns [h c2, (h + 1) c2) to rank h; the receive buffer, blocks in
 *       source order, IS the matrix [N1][c2] of this rank's columns;
 *   plk_dev_fourstep_cols: the N1-point column transforms, the twiddles omega_N^(k1 n2) (and 1 / N if inverse) ->
 *       send2_out = [k1][c2] with k1 in natural order, so destination h's block is rows [h r1, (h + 1) r1);
 *   exchange 2 (caller): the receive buffer is [source h][k1 local][c2];
 *   plk_dev_fourstep_rows: the N2-point row transforms on that buffer -> out, canonical values, either
 *       natural_layout == 0: [k1 local][k2] = X[k1 + N1 k2] (N / ranks elements, no further exchange), or
 *       natural_layout != 0: [h][k2 local][k1 local] with k2 = h c2 + k2 local, the send buffer of exchange 3, after
 *       which rank h holds [source g][k2 local][k1 local]; k = k1 + N1 k2 with k1 = g r1 + k1 local, so one
 *       transposition of the (g, k2 local) axes gives its contiguous range of X.
 * Buffers hold N / ranks elements; input and output of a call must not alias.  PLK_ERR_ARG if `ranks` does not divide
 * both N1 and N2 or leaves fewer than 2^floor(log2(N2) / 2) columns per rank (sizes above 2^7 per side). */
int plk_dev_fourstep_cols(plk_ctx* ctx, void* stream, const uint64_t* recv1, size_t N, size_t ranks, size_t rank, int inverse,
                          uint64_t* send2_out);
int plk_dev_fourstep_rows(plk_ctx* ctx, void* stream, const uint64_t* recv2, size_t N, size_t ranks, size_t rank, int inverse,
                          int natural_layout, uint64_t* out);

/* out[i] = a[i] * b[i] (field product), n elements; out may alias a or b */
int plk_dev_mul(plk_ctx* ctx, void* stream, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);

/* out[r * cols + c] = omega_N^((row0 + r) * (col0 + c)) (omega_N^-1 if inverse), optionally times 1/N:
 * the inter-phase twiddle block of a four-step transform of size N owned by one rank.
 * No counterpart in the reference (ark-poly runs one shared-memory FFT). */
int plk_dev_twiddle_block(plk_ctx* ctx, void* stream, size_t N, int inverse, int scale_by_inverse_n, size_t rows, size_t cols,
                          size_t row0, size_t col0, uint64_t* out);

int plk_dev_msm_set_bases(plk_ctx* ctx, const uint64_t* bases_dev, size_t n);
int plk_dev_msm_g1(plk_ctx* ctx, void* stream, const uint64_t* scalars_dev, size_t n, uint64_t out_xy[12]);
int plk_dev_msm_g1_batch(plk_ctx* ctx, void* stream, const uint64_t* const scalars_dev[], size_t count, size_t n, uint64_t* out_xy);
int plk_dev_g1_multiples(plk_ctx* ctx, void* stream, const uint64_t base_xy[12], uint64_t first, size_t count, uint64_t* out_dev);

/* plk_coset_ntt for `batch` device columns, plk_quotient on device vectors (alpha, shift stay host pointers) */
int plk_dev_coset_ntt(plk_ctx* ctx, void* stream, const uint64_t* const* in, size_t batch, size_t n_in, size_t N,
                      const uint64_t shift[4], int inverse, uint64_t* const* out);
int plk_dev_quotient(plk_ctx* ctx, void* stream, const uint64_t* const* numerators, size_t count, const uint64_t alpha[4],
                     size_t n, size_t N, const uint64_t shift[4], uint64_t* coeffs_out);

/* plk_gate_table_fill on device memory (default_val stays a host pointer) */
int plk_dev_gate_table_fill(plk_ctx* ctx, void* stream, const uint8_t* row_ops, const uint64_t* const node_ids[PLK_NUM_WIRES],
                            size_t n, const uint64_t* node_values, size_t n_values, const uint64_t default_val[4],
                            uint64_t* const wires_out[PLK_NUM_WIRES], uint64_t* const sel_out[PLK_NUM_SELECTORS]);

/* plk_sigma_table on device columns (n uint64_t each) */
int plk_dev_sigma_table(plk_ctx* ctx, void* stream, const uint64_t* const node_ids[PLK_NUM_WIRES], size_t n,
                        uint64_t* const sigma_out[PLK_NUM_WIRES]);

int plk_dev_grand_product(plk_ctx* ctx, void* stream, const uint64_t* const wires[PLK_NUM_WIRES],
                          const uint64_t* const sigma[PLK_NUM_WIRES], size_t n,
                          const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out);

/* plk_dev_grand_product for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same z_out (n + 1 elements),
 * same deferred status through plk_dev_status, same kernels; the per-block products that the kernels hand to one
 * another go to a fixed array inside the library's code object (one element per 1024 rows, PLK_NOALLOC_MAX_ROWS
 * rows at most: 32 KiB per GPU, held until the process ends) instead of the context's grow-only workspace, so no
 * call allocates or frees anything.  n above PLK_NOALLOC_MAX_ROWS is PLK_ERR_ARG, and so is a call between
 * plk_profile_begin and plk_profile_end (per-launch events are allocated there).  All contexts on one GPU share
 * that GPU's array: calls that may overlap on one GPU (two contexts on two streams; two processes do NOT share
 * it) race and must be ordered by the caller.  Nothing is carried from one call to the next: every word the
 * kernels read from the array was written earlier in the same call. */
#define PLK_NOALLOC_MAX_ROWS ((size_t)1 << 20)
int plk_dev_grand_product_noalloc(plk_ctx* ctx, void* stream, const uint64_t* const wires[PLK_NUM_WIRES],
                                  const uint64_t* const sigma[PLK_NUM_WIRES], size_t n,
                                  const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out);

/* The same grand product with the rows split over several GPUs (no counterpart in the reference, which
 * runs one serial loop, src/permutation_polynomial.rs:227-259).  Every GPU holds the whole columns
 * (the numerator gathers reach any row) and owns the rows [first, first + rows):
 *   plk_dev_grand_product_local  - per-row terms and local scans, kept in the context; writes the
 *       range's numerator product and denominator product to totals_out[0], totals_out[1] (2 elements);
 *   (the caller gathers every range's totals, in row order, into totals_all: 2 * ranges elements)
 *   plk_dev_grand_product_finish - applies the other ranges' products and the one inversion, and writes
 *       z_after_out[i] = z[first + i + 1], i < rows (z[0] = 1 is not written).
 * The two calls must follow each other on one context and stream with the same `rows`. */
int plk_dev_grand_product_local(plk_ctx* ctx, void* stream, const uint64_t* const wires[PLK_NUM_WIRES],
                                const uint64_t* const sigma[PLK_NUM_WIRES], size_t n, size_t first, size_t rows,
                                const uint64_t beta[4], const uint64_t gamma[4], uint64_t* totals_out);
int plk_dev_grand_product_finish(plk_ctx* ctx, void* stream, size_t rows, const uint64_t* totals_all, size_t ranges,
                                 size_t mine, uint64_t* z_after_out);

/* loop of src/execution_trace.rs:379-428 on columns already evaluated on the N-point domain */
int plk_dev_gate_constraint_eval(plk_ctx* ctx, void* stream, const uint64_t* const wire_evals[PLK_NUM_WIRES],
                                 const uint64_t* const sel_evals[PLK_NUM_SELECTORS], size_t N, uint64_t* out);

/* loop of src/permutation_polynomial.rs:341-373 on columns already evaluated on the N-point domain */
int plk_dev_perm_constraint_eval(plk_ctx* ctx, void* stream, const uint64_t* z_evals,
                                 const uint64_t* const sigma_evals[PLK_NUM_WIRES],
                                 const uint64_t* const wire_evals[PLK_NUM_WIRES], size_t N,
                                 const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* out);

/* The same loop on `count` consecutive points of the domain starting at index0 (the identity term
 * uses index0 + i, src/permutation_polynomial.rs:351).  z_evals holds count + z_shift elements:
 * the range followed by its next z_shift points, so no wrap-around is taken.  Used when the
 * evaluation domain is split over GPUs. */
int plk_dev_perm_constraint_eval_range(plk_ctx* ctx, void* stream, const uint64_t* z_evals,
                                       const uint64_t* const sigma_evals[PLK_NUM_WIRES],
                                       const uint64_t* const wire_evals[PLK_NUM_WIRES], size_t count, size_t index0,
                                       const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* out);

int plk_dev_round23(plk_ctx* ctx, void* stream, const uint64_t* const wires[PLK_NUM_WIRES],
                    const uint64_t* const sel[PLK_NUM_SELECTORS], const uint64_t* const sigma[PLK_NUM_WIRES],
                    size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift,
                    uint64_t* z_out, uint64_t* gate_out, uint64_t* perm_out);

/* Waits for `stream` and returns the deferred data-dependent status of the device calls
 * enqueued since the last plk_dev_status (PLK_OK, PLK_ERR_ZERO_DENOMINATOR, PLK_ERR_SIGMA_RANGE). */
int plk_dev_status(plk_ctx* ctx, void* stream);

/* Per-launch timing.  Between plk_profile_begin and plk_profile_end every kernel the library
 * launches on this context is bracketed by HIP events on its own stream.  plk_profile_end waits
 * for the device and returns a JSON array (valid until the next call on this context) of
 * {"label", "launches", "total_ms", "alg_bytes"} per kernel label, alg_bytes being the
 * compulsory bytes of those launches (inputs read once + outputs written once).
 * No counterpart in the reference. */
void plk_profile_begin(plk_ctx* ctx);
const char* plk_profile_end(plk_ctx* ctx);

/* canonical integer limbs <-> Montgomery limbs, n elements, device pointers, out may alias in */
int plk_dev_to_montgomery(plk_ctx* ctx, void* stream, const uint64_t* in, size_t n, uint64_t* out);
int plk_dev_from_montgomery(plk_ctx* ctx, void* stream, const uint64_t* in, size_t n, uint64_t* out);

/* ---- device memory for callers without a HIP binding (a Rust or C host that uses the plk_dev_* calls) ---------- */

/* `bytes` of memory on the context's GPU (hipMalloc), to be released with plk_dev_free.  No counterpart in the
 * reference, whose vectors live in host memory. */
int plk_dev_alloc(plk_ctx* ctx, size_t bytes, void** out);
void plk_dev_free(plk_ctx* ctx, void* p);
/* Copies between host memory and the context's GPU, ordered on `stream` like every other device call; the call
 * returns when the host buffer may be reused (upload) or holds the data (download).  Large pageable buffers go
 * through the context's threaded staging copy. */
int plk_dev_upload(plk_ctx* ctx, void* stream, void* dst_dev, const void* src_host, size_t bytes);
int plk_dev_download(plk_ctx* ctx, void* stream, void* dst_host, const void* src_dev, size_t bytes);
/* the context's own stream (what the host entry points enqueue on), for callers that have no stream of their own */
void* plk_ctx_stream(plk_ctx* ctx);

/* ---- one node: G ranks in ONE process, one GPU each ------------------------------------------------------------
 *
 * No counterpart in the reference (ark-poly runs one shared-memory FFT; nothing in the reference is distributed).
 * A node owns one context, one stream and one host thread per rank.  The ranks exchange data either by RCCL
 * (ncclAllToAll on communicators from ncclCommInitAll: one GPU per rank, the production form on an 8 x MI355X
 * node, one message per peer = one per xGMI link) or by device-to-device copies on the ranks' streams ordered by
 * events (PLK_NODE_COPY: peer copies between GPUs with peer access, plain copies when several ranks are given the
 * SAME device - which is how the index bookkeeping of G = 2, 4, 8 ranks is rehearsed on one GPU).
 * RCCL is bound at run time (dlopen of librccl.so.1) when a node asks for it: a process that never creates an
 * RCCL node does not load it, and a process that already holds a copy (PyTorch ships its own) uses that one. */
typedef struct plk_node plk_node;
enum { PLK_NODE_RCCL = 0, PLK_NODE_COPY = 1 };

/* devices[g] = GPU of rank g (G a power of two, 1 <= G <= 64).  PLK_NODE_RCCL needs G different devices.
 * plk_node_create = plk_node_create_field with PLK_FIELD_BLS12_381_FR. */
int plk_node_create(const int* devices, size_t G, int exchange, plk_node** out);
int plk_node_create_field(const int* devices, size_t G, int exchange, int field_id, plk_node** out);
void plk_node_destroy(plk_node* node);
size_t plk_node_ranks(const plk_node* node);
/* rank g's context (owned by the node): for plk_dev_alloc / plk_dev_upload / plk_dev_* calls on that rank's GPU */
plk_ctx* plk_node_ctx(plk_node* node, size_t g);
/* text of the node's last failure ("" if none) */
const char* plk_node_last_error(const plk_node* node);

/* ONE transform of N points over the node's G ranks (four-step form: plk_dev_fourstep_cols / _rows on every rank
 * and three all-to-all exchanges; src/polynomial_utils.rs:29 and src/execution_trace.rs:362-370 at a size one GPU
 * is not asked to hold).  in_shards[g] / out_shards[g]: N / G elements on rank g's GPU.  Input sharded
 * contiguously by index: rank g holds elements [g N / G, (g + 1) N / G).  Output, natural_output != 0: the same
 * ranges of the result (natural order in and out); natural_output == 0: rank g holds the block
 * [k1 local][k2] = X[(g r1 + k1 local) + N1 k2] (r1 = N1 / G, N1 = 2^ceil(log2(N) / 2)), saving the third exchange.
 * inverse != 0: inverse transform including 1 / N.  out_shards[g] must not alias in_shards[g].
 * The call returns when the work is enqueued on the ranks' streams; plk_node_sync waits for it.  Constraints as
 * for plk_dev_fourstep_cols (G divides both sides of the N1 x N2 view, enough columns per rank). */
int plk_node_ntt(plk_node* node, const uint64_t* const* in_shards, size_t N, int inverse, int natural_output,
                 uint64_t* const* out_shards);
/* The same transform on host arrays (N elements each, natural order in and out): shards go up to the ranks, the
 * transform runs, the result comes back; returns when `out` is complete. */
int plk_node_ntt_host(plk_node* node, const uint64_t* in, size_t N, int inverse, uint64_t* out);
/* waits for everything the node has enqueued on every rank */
int plk_node_sync(plk_node* node);

/* Batched proofs (replicas only, no exchange): `count` independent circuits of n rows, circuit i on rank i % G with
 * ALL its pointers in that rank's GPU memory; every rank runs plk_dev_round23 on its circuits one after another.
 * Returns when every circuit is done: PLK_OK, or the code of the first circuit (lowest index) that failed, with
 * plk_node_last_error naming it. */
typedef struct {
  const uint64_t* wires[PLK_NUM_WIRES];
  const uint64_t* sel[PLK_NUM_SELECTORS];     /* NULL = selector absent */
  const uint64_t* sigma[PLK_NUM_WIRES];
  uint64_t beta[4], gamma[4];
  uint64_t *z_out, *gate_out, *perm_out;      /* n + 1, N, N elements; gate_out / perm_out may be NULL */
} plk_circuit_io;
int plk_node_round23_batch(plk_node* node, const plk_circuit_io* circuits, size_t count, size_t n, size_t N, size_t z_shift);

/* ONE proof over the node's G ranks (independent polynomials sharded over the GPUs): per_rank[g] holds the SAME
 * circuit's columns resident on rank g's GPU.  The 18 columns (4 wires, the selectors present, 4 sigma, Z) are dealt
 * to the ranks for interpolation and extension to N = 4 n points (N / G >= z_shift >= 1), one all-to-all turns "my
 * columns, all points" into "all columns, my points" (with a z_shift-point halo of Z), and every rank evaluates both
 * constraint vectors on its range: per_rank[g].z_out receives the whole Z (n + 1 elements, computed on every rank),
 * gate_out / perm_out the points [g N / G, (g + 1) N / G) of the vectors plk_round23 returns (N / G elements each).
 * Returns when every rank is done; data-dependent failures come back as for plk_round23 with the rank named in
 * plk_node_last_error.  Every rank keeps two exchange buffers of G * ceil(18 / G) * (N / G + z_shift) elements
 * (grow-only, released by plk_node_destroy) beside its context's workspace: 2 x 403 MB at n = 2^20, G = 8. */
int plk_node_round23(plk_node* node, const plk_circuit_io* per_rank, size_t n, size_t N, size_t z_shift);

/* plk_node_round23 on HOST arrays, the arguments of plk_round23: every rank's GPU receives the columns over its own
 * link, the ranks' pieces of the two constraint vectors come back into gate_out / perm_out (N elements each), Z from
 * rank 0 into z_out (n + 1).  N = 4 n only.  The columns stay in a per-rank buffer the node keeps (grow-only). */
int plk_node_round23_host(plk_node* node, const uint64_t* const wires[4], const uint64_t* const sel[PLK_NUM_SELECTORS],
                          const uint64_t* const sigma[4], size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4],
                          size_t z_shift, uint64_t* z_out, uint64_t* gate_out, uint64_t* perm_out);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
