//! Patch point C -- the witness of `BigUintChip::{mul_mod, pow_mod_fixed_exp}` (biguint-halo2; called from
//! /root/reference/src/paillier.rs:51,55,57,81) from ONE device call instead of num-bigint's `a * b` / `div_rem` per step
//!.
use num_bigint::BigUint;

fn limbs64(x: &BigUint, n: usize) -> Vec<u64> {
    let mut v = x.to_u64_digits();
    assert!(v.len() <= n);
    v.resize(n, 0);
    v
}

/// Filled in `PaillierChip::encrypt` (paillier.rs:32-60) before the first `pow_mod_fixed_exp`; `BigUintChip::mul_mod` then POPS its
/// `(q, r)` instead of dividing, and asserts that the popped operands are the ones it was called with (get_biguint order: paillier.rs:22-30).
pub struct TraceQueue {
    trace: pz_rt::EncryptTrace,
    limbs: usize, // 2 * limbs of n: every operand of the chain is an n^2-sized integer
    next: usize,
}
impl TraceQueue {
    pub fn for_encrypt(n: &BigUint, g: &BigUint, m: &BigUint, r: &BigUint, limbs_n: usize) -> Self {
        let trace = pz_rt::encrypt_trace(&limbs64(n, limbs_n), &limbs64(g, limbs_n), &limbs64(m, limbs_n), &limbs64(r, limbs_n));
        TraceQueue { trace, limbs: 2 * limbs_n, next: 0 }
    }
    /// the next `mul_mod` of the chain: -> (q, r); panics where the reference would compute a different step (operand mismatch)
    pub fn pop(&mut self, a: &BigUint, b: &BigUint) -> (BigUint, BigUint) {
        let l = self.limbs;
        let rec = &self.trace.steps[self.next * 4 * l..(self.next + 1) * 4 * l];
        let big = |w: &[u64]| BigUint::from_slice(&w.iter().flat_map(|x| [*x as u32, (*x >> 32) as u32]).collect::<Vec<u32>>());
        assert_eq!(&big(&rec[..l]), a, "trace step {}: operand a", self.next);
        assert_eq!(&big(&rec[l..2 * l]), b, "trace step {}: operand b", self.next);
        self.next += 1;
        (big(&rec[2 * l..3 * l]), big(&rec[3 * l..]))
    }
    pub fn ciphertext(&self) -> BigUint {
        BigUint::from_slice(&self.trace.c.iter().flat_map(|x| [*x as u32, (*x >> 32) as u32]).collect::<Vec<u32>>())
    }
}

/// `paillier_enc_native` (paillier.rs:87-92) through the library: no trace kept
pub fn paillier_enc_native(n: &BigUint, g: &BigUint, m: &BigUint, r: &BigUint, limbs_n: usize) -> BigUint {
    let (n_, g_, m_, r_) = (limbs64(n, limbs_n), limbs64(g, limbs_n), limbs64(m, limbs_n), limbs64(r, limbs_n));
    let mut c = vec![0u64; 2 * limbs_n];
    let (mut sg, mut sr) = (0u32, 0u32);
    pz_rt::check(unsafe {
        pz_sys::pz_paillier_encrypt(pz_rt::ctx(), limbs_n as u32, 1, n_.as_ptr(), g_.as_ptr(), m_.as_ptr(), r_.as_ptr(), core::ptr::null_mut(), 0, &mut sg, &mut sr,
                                    c.as_mut_ptr())
    });
    BigUint::from_slice(&c.iter().flat_map(|x| [*x as u32, (*x >> 32) as u32]).collect::<Vec<u32>>())
}

/// `PaillierChip::add`'s single step (paillier.rs:81) / `paillier_add_native` (paillier.rs:94-97)
pub fn mul_mod(a: &BigUint, b: &BigUint, modulus: &BigUint, limbs: usize) -> (BigUint, BigUint) {
    let (a_, b_, m_) = (limbs64(a, limbs), limbs64(b, limbs), limbs64(modulus, limbs));
    let (mut q, mut r) = (vec![0u64; limbs], vec![0u64; limbs]);
    pz_rt::check(unsafe { pz_sys::pz_mul_mod(pz_rt::ctx(), limbs as u32, a_.as_ptr(), b_.as_ptr(), m_.as_ptr(), q.as_mut_ptr(), r.as_mut_ptr()) });
    let big = |w: &[u64]| BigUint::from_slice(&w.iter().flat_map(|x| [*x as u32, (*x >> 32) as u32]).collect::<Vec<u32>>());
    (big(&q), big(&r))
}
