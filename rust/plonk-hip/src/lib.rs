//! The polynomial hot path of wu-s-john/PLONK on an MI355X, behind the reference's own signatures.
//!
//! Every function below keeps the name, generic bound, argument order and panic behaviour of the function it
//! replaces (cited per item, paths relative to the reference tree) and forwards to `libplonk_hip.so` through
//! `plonk-hip-sys`.  NOT compiled in this repository's environment (no Rust toolchain, no arkworks crates
//! there); the C++ and Python mirrors of the same ABI are what the test suite runs.
//!
//! The library is instantiated for the BLS12-381 scalar field (the default; every measured and checked figure is
//! for it) and for the scalar field of Tweedledee (the reference names `ark_tweedle::Fr`, src/main.rs:3; that
//! instantiation's generator and root of unity are self-derived, not checked against ark-ff).  The functions stay
//! generic over `F` as in the reference; `check_field::<F>()` picks the instantiation by `F::characteristic()` and
//! panics for any other `F`.  Elements cross the boundary as the 4 x u64
//! Montgomery limbs that ARE the memory of ark-ff 0.3 `Fp256<P>` (`x.0 .0`), so slices are passed by pointer.

use std::collections::HashMap;
use std::ffi::CStr;
use std::mem::size_of;
use std::ptr;
use std::sync::Mutex;

use ark_ff::{FftField, Field};
use ark_poly::{univariate::DensePolynomial, EvaluationDomain, GeneralEvaluationDomain, UVPolynomial};
use plonk_hip_sys as sys;

// ------------------------------------------------------------------------------------------- data types

/// Selector key, declaration order = column index in the C ABI (src/ast.rs:55-66).
#[derive(Debug, Clone, PartialEq, Eq, Hash)]
pub enum ASTKind {
    Int,
    Bool,
    Add,
    Sub,
    Mult,
    Div,
    Eq,
    Not,
    If,
}

impl ASTKind {
    pub const ALL: [ASTKind; 9] = [
        ASTKind::Int, ASTKind::Bool, ASTKind::Add, ASTKind::Sub, ASTKind::Mult, ASTKind::Div, ASTKind::Eq, ASTKind::Not, ASTKind::If,
    ];
    fn index(&self) -> usize {
        ASTKind::ALL.iter().position(|k| k == self).unwrap()
    }
}

/// src/execution_trace.rs:204-214
#[derive(Debug, Clone)]
pub struct GateEvaluationTable<F> {
    pub input1: Vec<F>,
    pub input2: Vec<F>,
    pub input3: Vec<F>,
    pub output: Vec<F>,
    pub selectors: HashMap<ASTKind, Vec<F>>,
}

/// src/execution_trace.rs:307-314
#[derive(Debug, Clone)]
pub struct GatePolynomials<F: FftField> {
    pub input1: DensePolynomial<F>,
    pub input2: DensePolynomial<F>,
    pub input3: DensePolynomial<F>,
    pub output: DensePolynomial<F>,
    pub selectors: HashMap<ASTKind, DensePolynomial<F>>,
}

/// src/permutation_polynomial.rs:51-61
#[derive(Debug, Clone)]
pub struct PermutationTable<F> {
    pub sigma1: Vec<F>,
    pub sigma2: Vec<F>,
    pub sigma3: Vec<F>,
    pub sigma_output: Vec<F>,
}

impl<F> PermutationTable<F> {
    /// src/permutation_polynomial.rs:63-75
    pub fn new(size: usize, default: F) -> Self
    where
        F: Clone,
    {
        Self {
            sigma1: vec![default.clone(); size],
            sigma2: vec![default.clone(); size],
            sigma3: vec![default.clone(); size],
            sigma_output: vec![default.clone(); size],
        }
    }
}

/// src/permutation_polynomial.rs:263-269
pub struct PermutationPolynomials<F: Field> {
    pub sigma1: DensePolynomial<F>,
    pub sigma2: DensePolynomial<F>,
    pub sigma3: DensePolynomial<F>,
    pub sigma_output: DensePolynomial<F>,
    pub z: DensePolynomial<F>,
}

// ------------------------------------------------------------------------------------------- context

struct Ctx(*mut sys::plk_ctx);
unsafe impl Send for Ctx {}

/// One context per process on GPU `PLONK_HIP_DEVICE` (default 0), created at the first call.  A context must
/// not be used from two threads at once, so every call holds this lock; the reference's functions are
/// re-entrant and callers may use any thread.
/// (one per field the library is instantiated for, indexed by the `PLK_FIELD_*` number)
static CTX: Mutex<[Option<Ctx>; sys::PLK_NUM_FIELDS]> = Mutex::new([None, None]);

/// Runs `f` on the context under the lock.  `f` reports a failure as the text of the panic the reference would
/// raise (`check` builds it, reading the context's error string and row while the lock is still held); the panic
/// itself is raised here AFTER the guard has been dropped, so the reference's ordinary, recoverable panics (a
/// zero denominator, a sigma index, a length: `catch_unwind`, `#[should_panic]` tests sharing one binary) never
/// poison the lock.  A lock poisoned by a panic in the caller's own closure (`build_and_convert`'s builder) is
/// taken over all the same: the C context stays valid after any error return.
fn with_ctx(field: usize, f: impl FnOnce(*mut sys::plk_ctx) -> Result<(), String>) {
    let failure = {
        let mut guard = CTX.lock().unwrap_or_else(|e| e.into_inner());
        let mut create_failure = None;
        if guard[field].is_none() {
            let device = std::env::var("PLONK_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut p: *mut sys::plk_ctx = ptr::null_mut();
            let rc = unsafe { sys::plk_ctx_create_field(device, field as i32, &mut p) };
            if rc == sys::PLK_OK {
                guard[field] = Some(Ctx(p));
            } else {
                create_failure = Some(format!("plk_ctx_create_field(device = {}, field = {}) failed: {}", device, field, strerror(rc)));
            }
        }
        match create_failure {
            Some(msg) => Err(msg),
            None => f(guard[field].as_ref().unwrap().0),
        }
    }; // the guard is gone here
    if let Err(msg) = failure {
        panic!("{}", msg);
    }
}

fn strerror(rc: i32) -> String {
    unsafe { CStr::from_ptr(sys::plk_strerror(rc)) }.to_string_lossy().into_owned()
}

fn last_error(ctx: *mut sys::plk_ctx, rc: i32) -> String {
    let s = unsafe { CStr::from_ptr(sys::plk_last_error(ctx)) }.to_string_lossy().into_owned();
    if s.is_empty() { strerror(rc) } else { s }
}

/// Turns a status code into the text of the panic the reference raises at the cited line (`with_ctx` raises it
/// once the lock is released).
fn check(ctx: *mut sys::plk_ctx, rc: i32, length_message: &str) -> Result<(), String> {
    match rc {
        sys::PLK_OK => Ok(()),
        // assert_eq! on lengths, src/polynomial_utils.rs:22-26 and :56-60
        sys::PLK_ERR_LENGTH => Err(format!("assertion failed: `(left == right)`: {}", length_message)),
        // `denominator.inverse().unwrap()`, src/permutation_polynomial.rs:258
        sys::PLK_ERR_ZERO_DENOMINATOR => Err(format!(
            "called `Option::unwrap()` on a `None` value (zero denominator at row {})",
            unsafe { sys::plk_last_error_row(ctx) }
        )),
        // slice index, src/permutation_polynomial.rs:235-238
        sys::PLK_ERR_SIGMA_RANGE => Err(format!("index out of bounds (sigma at row {})", unsafe { sys::plk_last_error_row(ctx) })),
        _ => Err(format!("libplonk_hip: {}", last_error(ctx, rc))),
    }
}

/// The library's number (`PLK_FIELD_*`) for `F`: the instantiation whose modulus is `F::characteristic()`, for a
/// 32-byte prime field (4 x u64 Montgomery limbs, the memory of ark-ff 0.3 `Fp256`).  Panics for any other `F`.
/// Field 0 is the BLS12-381 scalar field; field 1 the scalar field of Tweedledee, whose root of unity the library
/// derives from the generator 5 - results agree with ark-poly's only if ark-ff's parameters for it use that generator.
fn check_field<F: Field>() -> usize {
    assert!(size_of::<F>() == 32 && F::extension_degree() == 1, "libplonk_hip implements 256-bit prime fields (4 x u64 Montgomery limbs) only");
    for id in 0..sys::PLK_NUM_FIELDS {
        let mut modulus = [0u64; 4];
        let rc = unsafe { sys::plk_field_constants_of(id as i32, modulus.as_mut_ptr(), ptr::null_mut(), ptr::null_mut()) };
        if rc == sys::PLK_OK && F::characteristic() == &modulus[..] {
            return id;
        }
    }
    panic!("libplonk_hip is not instantiated for this field (it has the BLS12-381 scalar field and the Tweedledee scalar field)");
}

fn limbs<F>(v: &[F]) -> *const u64 {
    v.as_ptr() as *const u64
}
fn limbs_mut<F>(v: &mut [F]) -> *mut u64 {
    v.as_mut_ptr() as *mut u64
}
fn one_elem<F>(x: &F) -> *const u64 {
    x as *const F as *const u64
}

/// `usize` index columns as the `u64` the ABI takes (`PermutationTable<usize>`; identical memory on a 64-bit host).
fn index_ptr(v: &[usize]) -> *const u64 {
    assert_eq!(size_of::<usize>(), 8, "a 64-bit host is assumed");
    v.as_ptr() as *const u64
}

/// Coefficients of `p` zero-padded to `len` (`DensePolynomial` drops trailing zero coefficients, so the
/// vectors of one call can differ in length; the ABI takes one length for all).
fn padded<F: Field>(p: &DensePolynomial<F>, len: usize) -> Vec<F> {
    let mut v = p.coeffs.clone();
    v.resize(len, F::zero());
    v
}

// ------------------------------------------------------------------------------------------- the node (several GPUs)

struct Node(*mut sys::plk_node);
unsafe impl Send for Node {}

/// `PLONK_HIP_NODE_DEVICES=0,1,2,3,4,5,6,7` makes transforms of at least `PLONK_HIP_NODE_MIN_SIZE` points (default
/// 2^22) run sharded over those GPUs (four-step form, RCCL all-to-all; `PLONK_HIP_NODE_EXCHANGE=copy` selects
/// device-to-device copies instead).  Created at the first such call; `None` when the variable is not set.
static NODE: Mutex<Option<Option<Node>>> = Mutex::new(None);

fn node_min_size() -> usize {
    std::env::var("PLONK_HIP_NODE_MIN_SIZE").ok().and_then(|s| s.parse().ok()).unwrap_or(1 << 22)
}

/// Runs `f` on the node if one is configured and returns true; false = no node, the caller takes the one-GPU path.
/// As in `with_ctx`, a failure is carried out of the locked region as text and raised after the guard is gone.
fn with_node(f: impl FnOnce(*mut sys::plk_node) -> Result<(), String>) -> bool {
    let outcome: Option<Result<(), String>> = {
        let mut guard = NODE.lock().unwrap_or_else(|e| e.into_inner());
        let mut create_failure: Option<String> = None;
        if guard.is_none() {
            let devices: Vec<i32> = std::env::var("PLONK_HIP_NODE_DEVICES")
                .map(|s| s.split(',').filter_map(|d| d.trim().parse().ok()).collect())
                .unwrap_or_default();
            if devices.is_empty() {
                *guard = Some(None);
            } else {
                let exchange = match std::env::var("PLONK_HIP_NODE_EXCHANGE").as_deref() {
                    Ok("copy") => sys::PLK_NODE_COPY,
                    _ => sys::PLK_NODE_RCCL,
                };
                let mut p: *mut sys::plk_node = ptr::null_mut();
                let rc = unsafe { sys::plk_node_create(devices.as_ptr(), devices.len(), exchange, &mut p) };
                if rc == sys::PLK_OK {
                    *guard = Some(Some(Node(p)));
                } else {
                    *guard = Some(None); // later calls take the one-GPU path
                    create_failure = Some(format!("plk_node_create({:?}) failed: {}", devices, strerror(rc)));
                }
            }
        }
        if let Some(msg) = create_failure {
            Some(Err(msg))
        } else {
            match guard.as_ref().unwrap() {
                Some(node) => Some(f(node.0)),
                None => None,
            }
        }
    }; // the guard is gone here
    match outcome {
        Some(Err(msg)) => panic!("{}", msg),
        Some(Ok(())) => true,
        None => false,
    }
}

// ------------------------------------------------------------------------------------------- src/polynomial_utils.rs

/// src/polynomial_utils.rs:18-33.
pub fn evaluations_to_dense_polynomial<F: FftField>(evals: &[F], domain: &GeneralEvaluationDomain<F>) -> DensePolynomial<F> {
    let field = check_field::<F>();
    assert_eq!(evals.len(), domain.size(), "Evaluation length must match the domain size");
    let mut coeffs = vec![F::zero(); domain.size()];
    // a large transform on a configured node (BLS12-381 Fr): sharded over its GPUs
    if field == 0 && domain.size() >= node_min_size()
        && with_node(|node| {
            let rc = unsafe { sys::plk_node_ntt_host(node, limbs(evals), evals.len(), 1, limbs_mut(&mut coeffs)) };
            if rc == sys::PLK_OK {
                Ok(())
            } else {
                Err(format!("libplonk_hip: {}", unsafe { CStr::from_ptr(sys::plk_node_last_error(node)) }.to_string_lossy()))
            }
        })
    {
        return DensePolynomial::from_coefficients_vec(coeffs);
    }
    with_ctx(field, |ctx| {
        let rc = unsafe { sys::plk_intt(ctx, limbs(evals), evals.len(), domain.size(), limbs_mut(&mut coeffs)) };
        check(ctx, rc, "Evaluation length must match the domain size")
    });
    DensePolynomial::from_coefficients_vec(coeffs) // drops trailing zero coefficients exactly as at :32
}

/// src/polynomial_utils.rs:47-62.
pub fn build_and_convert<F, Builder>(domain: &GeneralEvaluationDomain<F>, builder: Builder) -> DensePolynomial<F>
where
    F: FftField,
    Builder: FnOnce() -> Vec<F>,
{
    let evals = builder();
    assert_eq!(domain.size(), evals.len(), "Domain size and builder evaluations length should match");
    evaluations_to_dense_polynomial(&evals, domain)
}

// ------------------------------------------------------------------------------------------- src/execution_trace.rs

/// src/execution_trace.rs:318-344: 4 + S interpolations in one batched call.
pub fn gate_evaluation_table_to_gate_polynomial<F: FftField>(
    table: &GateEvaluationTable<F>,
    coset_domain: &GeneralEvaluationDomain<F>,
) -> GatePolynomials<F> {
    let field = check_field::<F>();
    let n = coset_domain.size();
    let cols = [&table.input1, &table.input2, &table.input3, &table.output];
    for c in cols.iter() {
        assert_eq!(c.len(), n, "Evaluation length must match the domain size");
    }
    let wires: Vec<*const u64> = cols.iter().map(|c| limbs(c)).collect();
    let mut wire_out: Vec<Vec<F>> = (0..4).map(|_| vec![F::zero(); n]).collect();
    // HashMap<ASTKind, Vec<F>> <-> the 9-entry pointer table of the ABI, null = kind absent
    let mut sel: [*const u64; 9] = [ptr::null(); 9];
    let mut sel_out: Vec<Option<Vec<F>>> = (0..9).map(|_| None).collect();
    for (kind, evals) in &table.selectors {
        assert_eq!(evals.len(), n, "Evaluation length must match the domain size");
        sel[kind.index()] = limbs(evals);
        sel_out[kind.index()] = Some(vec![F::zero(); n]);
    }
    {
        let wire_out_ptrs: Vec<*mut u64> = wire_out.iter_mut().map(|v| limbs_mut(v)).collect();
        let sel_out_ptrs: Vec<*mut u64> =
            sel_out.iter_mut().map(|o| o.as_mut().map_or(ptr::null_mut(), |v| limbs_mut(v))).collect();
        with_ctx(field, |ctx| {
            let rc = unsafe { sys::plk_gate_polys(ctx, wires.as_ptr(), sel.as_ptr(), n, n, wire_out_ptrs.as_ptr(), sel_out_ptrs.as_ptr()) };
            check(ctx, rc, "Evaluation length must match the domain size")
        });
    }
    let mut selectors = HashMap::new();
    for (k, out) in sel_out.into_iter().enumerate() {
        if let Some(coeffs) = out {
            selectors.insert(ASTKind::ALL[k].clone(), DensePolynomial::from_coefficients_vec(coeffs));
        }
    }
    let mut it = wire_out.into_iter().map(DensePolynomial::from_coefficients_vec);
    GatePolynomials {
        input1: it.next().unwrap(),
        input2: it.next().unwrap(),
        input3: it.next().unwrap(),
        output: it.next().unwrap(),
        selectors,
    }
}

/// src/execution_trace.rs:357-431.
pub fn build_gate_constraint_evaluation_polynomial<F: FftField>(gate_poly: &GatePolynomials<F>, domain: &GeneralEvaluationDomain<F>) -> Vec<F> {
    let field = check_field::<F>();
    let big_n = domain.size();
    let polys = [&gate_poly.input1, &gate_poly.input2, &gate_poly.input3, &gate_poly.output];
    let n_coeffs = polys.iter().map(|p| p.coeffs.len()).chain(gate_poly.selectors.values().map(|p| p.coeffs.len())).max().unwrap_or(0);
    let wire_bufs: Vec<Vec<F>> = polys.iter().map(|p| padded(p, n_coeffs)).collect();
    let wires: Vec<*const u64> = wire_bufs.iter().map(|v| limbs(v)).collect();
    let mut sel_bufs: Vec<Option<Vec<F>>> = (0..9).map(|_| None).collect();
    for (kind, p) in &gate_poly.selectors {
        sel_bufs[kind.index()] = Some(padded(p, n_coeffs));
    }
    let sel: Vec<*const u64> = sel_bufs.iter().map(|o| o.as_ref().map_or(ptr::null(), |v| limbs(v))).collect();
    let mut out = vec![F::zero(); big_n];
    with_ctx(field, |ctx| {
        // more coefficients than domain points is the length panic inside evaluate_over_domain_by_ref
        let rc = unsafe { sys::plk_gate_constraints(ctx, wires.as_ptr(), sel.as_ptr(), n_coeffs, big_n, limbs_mut(&mut out)) };
        check(ctx, rc, "polynomial degree exceeds the evaluation domain")
    });
    out
}

// ------------------------------------------------------------------------------------------- src/permutation_polynomial.rs

/// src/permutation_polynomial.rs:217-261; returns n + 1 values.
pub fn build_grand_product_evaluation_polynomial<F: Field>(
    permutation_table: &PermutationTable<usize>,
    evaluation_table: &GateEvaluationTable<F>,
    beta: F,
    gamma: F,
) -> Vec<F> {
    let field = check_field::<F>();
    let n = evaluation_table.input1.len();
    let t = evaluation_table;
    let p = permutation_table;
    // the reference indexes rows 0..n of every column (:229-238): shorter columns are its slice panic
    for len in [t.input2.len(), t.input3.len(), t.output.len(), p.sigma1.len(), p.sigma2.len(), p.sigma3.len(), p.sigma_output.len()].iter() {
        assert!(*len >= n, "index out of bounds: the len is {} but the index is {}", len, len);
    }
    let wires = [limbs(&t.input1), limbs(&t.input2), limbs(&t.input3), limbs(&t.output)];
    let sigma = [index_ptr(&p.sigma1), index_ptr(&p.sigma2), index_ptr(&p.sigma3), index_ptr(&p.sigma_output)];
    let mut z = vec![F::one(); n + 1];
    with_ctx(field, |ctx| {
        let rc = unsafe { sys::plk_grand_product(ctx, wires.as_ptr(), sigma.as_ptr(), n, one_elem(&beta), one_elem(&gamma), limbs_mut(&mut z)) };
        check(ctx, rc, "")
    });
    z
}

/// src/permutation_polynomial.rs:272-298.  The reference hands the n + 1 grand-product values to a length-n
/// assert (:296 with src/polynomial_utils.rs:22-26) and panics whenever n equals the domain size; here
/// z[0..n) is interpolated.  `build_permutation_polynomials_with_last` also returns z[n].
pub fn build_permutation_polynomials<F: FftField>(
    permutation_table: &PermutationTable<usize>,
    evaluation_table: &GateEvaluationTable<F>,
    beta: F,
    gamma: F,
    coset_domain: &GeneralEvaluationDomain<F>,
) -> PermutationPolynomials<F> {
    build_permutation_polynomials_with_last(permutation_table, evaluation_table, beta, gamma, coset_domain).0
}

pub fn build_permutation_polynomials_with_last<F: FftField>(
    permutation_table: &PermutationTable<usize>,
    evaluation_table: &GateEvaluationTable<F>,
    beta: F,
    gamma: F,
    coset_domain: &GeneralEvaluationDomain<F>,
) -> (PermutationPolynomials<F>, F) {
    let field = check_field::<F>();
    let n = coset_domain.size();
    let t = evaluation_table;
    let p = permutation_table;
    for len in [t.input1.len(), t.input2.len(), t.input3.len(), t.output.len(), p.sigma1.len(), p.sigma2.len(), p.sigma3.len(), p.sigma_output.len()].iter() {
        assert_eq!(*len, n, "Evaluation length must match the domain size");
    }
    let wires = [limbs(&t.input1), limbs(&t.input2), limbs(&t.input3), limbs(&t.output)];
    let sigma = [index_ptr(&p.sigma1), index_ptr(&p.sigma2), index_ptr(&p.sigma3), index_ptr(&p.sigma_output)];
    let mut sig_out: Vec<Vec<F>> = (0..4).map(|_| vec![F::zero(); n]).collect();
    let mut z_coeffs = vec![F::zero(); n];
    let mut z_last = F::zero();
    {
        let sig_ptrs: Vec<*mut u64> = sig_out.iter_mut().map(|v| limbs_mut(v)).collect();
        with_ctx(field, |ctx| {
            let rc = unsafe {
                sys::plk_perm_polys(ctx, wires.as_ptr(), sigma.as_ptr(), n, n, one_elem(&beta), one_elem(&gamma), sig_ptrs.as_ptr(),
                                    limbs_mut(&mut z_coeffs), &mut z_last as *mut F as *mut u64)
            };
            check(ctx, rc, "Evaluation length must match the domain size")
        });
    }
    let mut it = sig_out.into_iter().map(DensePolynomial::from_coefficients_vec);
    let polys = PermutationPolynomials {
        sigma1: it.next().unwrap(),
        sigma2: it.next().unwrap(),
        sigma3: it.next().unwrap(),
        sigma_output: it.next().unwrap(),
        z: DensePolynomial::from_coefficients_vec(z_coeffs),
    };
    (polys, z_last)
}

/// src/permutation_polynomial.rs:312-376 (`z_evals[(i + 1) % n]`, identity term `F::from(i)`: z_shift = 1).
pub fn build_permutation_constraint_evaluation_polynomial<F: FftField>(
    polynomials: &PermutationPolynomials<F>,
    gate_polynomials: &GatePolynomials<F>,
    beta: F,
    gamma: F,
    evaluation_domain: &GeneralEvaluationDomain<F>,
) -> Vec<F> {
    let field = check_field::<F>();
    let big_n = evaluation_domain.size();
    let sig = [&polynomials.sigma1, &polynomials.sigma2, &polynomials.sigma3, &polynomials.sigma_output];
    let wir = [&gate_polynomials.input1, &gate_polynomials.input2, &gate_polynomials.input3, &gate_polynomials.output];
    let n_coeffs = sig.iter().chain(wir.iter()).map(|p| p.coeffs.len()).chain(std::iter::once(polynomials.z.coeffs.len())).max().unwrap();
    let z = padded(&polynomials.z, n_coeffs);
    let sig_bufs: Vec<Vec<F>> = sig.iter().map(|p| padded(p, n_coeffs)).collect();
    let wir_bufs: Vec<Vec<F>> = wir.iter().map(|p| padded(p, n_coeffs)).collect();
    let sig_ptrs: Vec<*const u64> = sig_bufs.iter().map(|v| limbs(v)).collect();
    let wir_ptrs: Vec<*const u64> = wir_bufs.iter().map(|v| limbs(v)).collect();
    let mut out = vec![F::zero(); big_n];
    with_ctx(field, |ctx| {
        let rc = unsafe {
            sys::plk_perm_constraints(ctx, limbs(&z), sig_ptrs.as_ptr(), wir_ptrs.as_ptr(), n_coeffs, big_n, one_elem(&beta), one_elem(&gamma), 1,
                                      limbs_mut(&mut out))
        };
        check(ctx, rc, "polynomial degree exceeds the evaluation domain")
    });
    out
}

// ------------------------------------------------------------------------------------------- the composition

/// Rounds 2+3 in one device-resident call: the composition of the functions above (gate polynomials -> grand
/// product -> sigma / Z interpolation -> gate constraints -> permutation constraints, wires evaluated once)
/// without returning to the host between the steps.  Returns (z [n + 1], gate constraints [N], permutation
/// constraints [N]).  No counterpart in the reference.
pub fn prove_round23<F: FftField>(
    evaluation_table: &GateEvaluationTable<F>,
    permutation_table: &PermutationTable<usize>,
    beta: F,
    gamma: F,
    coset_domain: &GeneralEvaluationDomain<F>,
    evaluation_domain: &GeneralEvaluationDomain<F>,
) -> (Vec<F>, Vec<F>, Vec<F>) {
    let field = check_field::<F>();
    let (n, big_n) = (coset_domain.size(), evaluation_domain.size());
    let t = evaluation_table;
    let p = permutation_table;
    for len in [t.input1.len(), t.input2.len(), t.input3.len(), t.output.len(), p.sigma1.len(), p.sigma2.len(), p.sigma3.len(), p.sigma_output.len()].iter() {
        assert_eq!(*len, n, "Evaluation length must match the domain size");
    }
    let wires = [limbs(&t.input1), limbs(&t.input2), limbs(&t.input3), limbs(&t.output)];
    let sigma = [index_ptr(&p.sigma1), index_ptr(&p.sigma2), index_ptr(&p.sigma3), index_ptr(&p.sigma_output)];
    let mut sel: [*const u64; 9] = [ptr::null(); 9];
    for (kind, evals) in &t.selectors {
        assert_eq!(evals.len(), n, "Evaluation length must match the domain size");
        sel[kind.index()] = limbs(evals);
    }
    let mut z = vec![F::zero(); n + 1];
    let mut gate = vec![F::zero(); big_n];
    let mut perm = vec![F::zero(); big_n];
    // on a configured node (BLS12-381 Fr, the 4n domain, circuits of at least PLONK_HIP_NODE_MIN_SIZE points): the one
    // proof's columns are dealt to the node's GPUs
    if field == 0 && big_n == 4 * n && n >= 4 && big_n >= node_min_size()
        && with_node(|node| {
            let rc = unsafe {
                sys::plk_node_round23_host(node, wires.as_ptr(), sel.as_ptr(), sigma.as_ptr(), n, big_n, one_elem(&beta), one_elem(&gamma), 1,
                                           limbs_mut(&mut z), limbs_mut(&mut gate), limbs_mut(&mut perm))
            };
            if rc == sys::PLK_OK {
                Ok(())
            } else {
                Err(format!("libplonk_hip: {}", unsafe { CStr::from_ptr(sys::plk_node_last_error(node)) }.to_string_lossy()))
            }
        })
    {
        return (z, gate, perm);
    }
    with_ctx(field, |ctx| {
        let rc = unsafe {
            sys::plk_round23(ctx, wires.as_ptr(), sel.as_ptr(), sigma.as_ptr(), n, big_n, one_elem(&beta), one_elem(&gamma), 1,
                             limbs_mut(&mut z), limbs_mut(&mut gate), limbs_mut(&mut perm))
        };
        check(ctx, rc, "Evaluation length must match the domain size")
    });
    (z, gate, perm)
}
