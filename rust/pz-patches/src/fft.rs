//! Patch point B -- halo2curves `fft.rs::best_fft`: G == Scalar == bn256::Fr on every prover call site.
use halo2curves::bn256::Fr;

pub fn best_fft(a: &mut [Fr], omega: Fr, log_n: u32) {
    assert_eq!(a.len(), 1usize << log_n);
    pz_rt::check(unsafe { pz_sys::pz_ntt_fr(pz_rt::ctx(), a.as_mut_ptr() as *mut u64, &omega as *const Fr as *const u64, log_n) });
}

/// `EvaluationDomain::lagrange_to_coeff` over many columns in one call (in place)
pub fn ifft_columns(cols: &mut [&mut [Fr]], omega_inv: Fr, log_n: u32) {
    let ptrs: Vec<*mut u64> = cols.iter_mut().map(|c| c.as_mut_ptr() as *mut u64).collect();
    pz_rt::check(unsafe { pz_sys::pz_ntt_fr_batch(pz_rt::ctx(), ptrs.as_ptr(), ptrs.len(), &omega_inv as *const Fr as *const u64, log_n) });
    // (the 1 / n scaling of ifft stays with the caller's `ifft_divisor` loop, or rides on pz_ntt_fr_dev's post_scale in the device-resident flow)
}
