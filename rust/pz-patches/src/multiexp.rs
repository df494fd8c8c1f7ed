//! Patch point A -- halo2curves `msm.rs::best_multiexp`.
use halo2curves::bn256::{Fr, G1Affine, G1};

/// `ParamsKZG::commit_lagrange` / `commit` call this once per column: the bases are the SRS slice (cached as a window-shifted table on
/// the device by address and length), the scalars cross PCIe.  No CPU fallback: a non-zero status panics (pz_rt::check).
pub fn best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> G1 {
    let ctx = pz_rt::ctx();
    let tbl = pz_rt::bases_for(bases.as_ptr() as *const u64, bases.len());
    let mut out = [0u64; 12];
    pz_rt::check(unsafe { pz_sys::pz_msm_g1(ctx, tbl, coeffs.as_ptr() as *const u64, coeffs.len(), out.as_mut_ptr()) });
    unsafe { core::mem::transmute_copy::<[u64; 12], G1>(&out) } // G1 {x, y, z}: the ABI's 96-byte Jacobian layout
}

/// all columns of a prover phase against the same bases in ONE launch sequence (what the patched `create_proof` calls instead of a
/// loop over `commit_lagrange`): uploads / downloads on the library's copy streams beside the neighbouring group's kernels
pub fn multiexp_columns(cols: &[&[Fr]], bases: &[G1Affine]) -> Vec<G1> {
    let ctx = pz_rt::ctx();
    let tbl = pz_rt::bases_for(bases.as_ptr() as *const u64, bases.len());
    let ptrs: Vec<*const u64> = cols.iter().map(|c| c.as_ptr() as *const u64).collect();
    let mut out = vec![0u64; 12 * cols.len()];
    pz_rt::check(unsafe { pz_sys::pz_msm_g1_batch(ctx, tbl, ptrs.as_ptr(), cols.len(), cols[0].len(), out.as_mut_ptr()) });
    out.chunks(12).map(|p| unsafe { core::mem::transmute_copy::<[u64; 12], G1>(&<[u64; 12]>::try_from(p).unwrap()) }).collect()
}
