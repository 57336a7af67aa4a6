"""MI355X-native PLONK polynomial hot path (NTT/iNTT, grand product Z, constraint evaluation).

The compute lives in libplonk_hip.so (hand-written HIP for gfx950, C ABI in include/plonk_hip.h);
this package is the host-side mirror of the reference's Rust interface on top of it.
"""
from .api import (AST_KINDS, NO_WIRE, Context, GateEvaluationTable, GatePolynomials, GeneralEvaluationDomain, LengthMismatch,
                  PermutationPolynomials, PermutationTable, PlonkError, Radix2EvaluationDomain, SigmaOutOfRange,
                  ZeroDenominator, build_and_convert, build_gate_constraint_evaluation_polynomial, build_gate_evaluation_table, row_ops,
                  build_grand_product_evaluation_polynomial, build_permutation_constraint_evaluation_polynomial,
                  build_permutation_polynomials, build_permutation_table, coset_fft, coset_ifft, degree, evaluate_over_domain, evaluations_to_dense_polynomial,
                  gate_evaluation_table_to_gate_polynomial, msm_g1, msm_g1_batch, msm_set_bases, prove_round23, quotient_polynomial)

__all__ = [n for n in dir() if not n.startswith("_")]
