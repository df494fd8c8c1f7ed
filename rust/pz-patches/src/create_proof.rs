//! Patch point D -- halo2-axiom `keygen_vk` / `keygen_pk` / `create_proof` for the reference's circuit (reached from
//! /root/reference/src/bench.rs:161-175) through the library's stepper: the transcript, the blinding randomness and the proof's byte
//! layout stay halo2's; everything between two transcript round trips is one call (INTEGRATION.md sections 5d, 5e).
use ff::PrimeField;
use halo2curves::bn256::{Fr, G1Affine};
use pz_rt::{DeviceKey, MessageShape, ProofSession};

fn mont(c: &Fr) -> [u64; 4] {
    unsafe { core::mem::transmute_copy::<Fr, [u64; 4]>(c) } // halo2curves keeps Fr as 4 x u64 Montgomery limbs: the ABI's layout
}
fn points(words: &[u64]) -> Vec<G1Affine> {
    words.chunks(8).map(|p| unsafe { core::mem::transmute_copy::<[u64; 8], G1Affine>(&<[u64; 8]>::try_from(p).unwrap()) }).collect()
}

/// What the patched prover needs from halo2's transcript (`TranscriptWrite<G1Affine, Challenge255<_>>` implements it in two lines each)
pub trait Rounds {
    fn write_points(&mut self, pts: &[G1Affine]);
    fn write_scalars(&mut self, words: &[u64]); // evaluations, 4 Montgomery words each, in pz.h's family order
    fn challenge(&mut self) -> Fr;
}

/// keygen for ONE message of the reference's circuit (its bits are structure: paillier.rs:50-55): structure on the device -> key
pub fn keygen_for_message(lagrange: *const pz_sys::pz_bases, monomial: *const pz_sys::pz_bases, k: u32, lookup_bits: u32, m: &[u64], n: &[u64]) -> (DeviceKey, MessageShape) {
    DeviceKey::for_message(lagrange, monomial, k, lookup_bits, m, n, /* calculate_params(Some(20)): bench_builder */ 20, usize::MAX)
}

/// `create_proof`: d_cols = the K4 columns on the device (pz_circuit_expand_cols_dev with `shape.d_col_starts`); -> whether the quotient's
/// degree is within bounds (false: the witness does not satisfy the circuit and the proof will not verify)
pub fn create_proof<T: Rounds>(key: &DeviceKey, shape: &MessageShape, d_cols: *mut u64, os_random: &[u64], tr: &mut T) -> bool {
    let (mut s, advice) = ProofSession::begin(key, d_cols, shape.n_adv, shape.n_lk, os_random);
    tr.write_points(&points(&advice));
    let theta = tr.challenge();
    let (a_perm, s_perm) = s.lookups(&mont(&theta));
    tr.write_points(&points(&a_perm));
    tr.write_points(&points(&s_perm));
    let (beta, gamma) = (tr.challenge(), tr.challenge());
    let (z, z_lookup, random) = s.products(&mont(&beta), &mont(&gamma));
    tr.write_points(&points(&z));
    tr.write_points(&points(&z_lookup));
    tr.write_points(&points(&random));
    let y = tr.challenge();
    let h = s.quotient(&mont(&y));
    tr.write_points(&points(&h));
    let x = tr.challenge();
    let evals = s.evaluate(&mont(&x));
    tr.write_scalars(&evals[..evals.len() - 4]); // the last element is h(x): the verifier computes it itself
    let (sh_y, sh_v) = (tr.challenge(), tr.challenge());
    let w1 = s.open_begin(&mont(&sh_y), &mont(&sh_v));
    tr.write_points(&points(&w1));
    let u = tr.challenge();
    let (w2, degree_ok) = s.open_finish(&mont(&u));
    tr.write_points(&points(&w2));
    let _ = Fr::NUM_BITS;
    degree_ok
}
