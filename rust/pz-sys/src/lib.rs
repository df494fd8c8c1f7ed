//! pz-sys -- raw bindings of the C ABI in `include/pz.h` (libpz_hip.so: the MI355X-native hot path of the Paillier-in-Halo2 prover).
//!
//! One `extern "C"` item per entry point, in the header's order.  This file is SOURCE ONLY in this repository: the build image has no
//! cargo / rustc, so it has never been compiled here; what IS checked mechanically
//! (tests/test_abi.py::test_bindings_match_the_header_parameter_by_parameter) is that every item below agrees with the header in
//! parameter count and in the pointer / c_int / u32 / u64 / usize / f64 class of every parameter and of the return value.
//! The reference crate (aerius-labs/paillier-halo2, `src/lib.rs:1-2`) has no FFI seam of its own; INTEGRATION.md shows the four
//! patch points that call into these bindings.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)] pub struct pz_ctx     { _p: [u8; 0] }
#[repr(C)] pub struct pz_bases   { _p: [u8; 0] }
#[repr(C)] pub struct pz_shplonk { _p: [u8; 0] }
#[repr(C)] pub struct pz_pk      { _p: [u8; 0] }
#[repr(C)] pub struct pz_structure { _p: [u8; 0] }
#[repr(C)] pub struct pz_proof   { _p: [u8; 0] }

extern "C" {
    pub fn pz_init(n_devices: c_int, device_ids: *const c_int, out: *mut *mut pz_ctx) -> c_int;
    pub fn pz_free(ctx: *mut pz_ctx) -> c_int;
    pub fn pz_strerror(status: c_int) -> *const c_char;
    pub fn pz_last_hip_error(ctx: *const pz_ctx) -> *const c_char;
    pub fn pz_set_stream(ctx: *mut pz_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn pz_sync(ctx: *mut pz_ctx) -> c_int;
    pub fn pz_abi_version() -> c_int;       // == PZ_ABI_VERSION (7); checked once in pz_rt::ctx()

    // device memory + context ordering: what lets the columns of a proof stay in HBM (patch points C / D, section 5a)
    pub fn pz_dev_alloc(ctx: *mut pz_ctx, bytes: usize, d_out: *mut *mut c_void) -> c_int;
    pub fn pz_dev_cache_limit(ctx: *mut pz_ctx, max_bytes: usize) -> c_int;
    pub fn pz_dev_arena(ctx: *mut pz_ctx, bytes: usize) -> c_int;
    pub fn pz_dev_arena_info(ctx: *mut pz_ctx, out: *mut u64) -> c_int;
    pub fn pz_dev_mem_info(ctx: *mut pz_ctx, free_bytes: *mut usize, total_bytes: *mut usize) -> c_int;
    pub fn pz_dev_free(ctx: *mut pz_ctx, d: *mut c_void) -> c_int;
    pub fn pz_upload(ctx: *mut pz_ctx, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn pz_download(ctx: *mut pz_ctx, dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn pz_dev_memset(ctx: *mut pz_ctx, d_dst: *mut c_void, byte_value: c_int, bytes: usize) -> c_int;
    pub fn pz_dev_copy(ctx: *mut pz_ctx, d_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn pz_dev_copy_2d(ctx: *mut pz_ctx, d_dst: *mut c_void, dst_pitch: usize, d_src: *const c_void, src_pitch: usize, width: usize,
                          rows: usize) -> c_int;
    pub fn pz_ctx_wait(waiter: *mut pz_ctx, producer: *mut pz_ctx) -> c_int;

    // K1  (replaces halo2curves::msm::best_multiexp)
    pub fn pz_srs_load_g1(ctx: *mut pz_ctx, k: u32, bases_affine: *const u64, lagrange: c_int,
                          out: *mut *mut pz_bases) -> c_int;
    pub fn pz_bases_load_g1(ctx: *mut pz_ctx, bases_affine: *const u64, n_points: usize, on_device: c_int,
                            window_bits: u32, out: *mut *mut pz_bases) -> c_int;
    pub fn pz_bases_free(ctx: *mut pz_ctx, bases: *mut pz_bases) -> c_int;
    pub fn pz_bases_info(bases: *const pz_bases, n_points: *mut usize, window_bits: *mut u32,
                         n_windows: *mut u32) -> c_int;
    pub fn pz_msm_g1(ctx: *mut pz_ctx, bases: *const pz_bases, scalars: *const u64, n: usize,
                     out_jac: *mut u64 /* [12] */) -> c_int;
    pub fn pz_msm_g1_batch(ctx: *mut pz_ctx, bases: *const pz_bases, scalar_cols: *const *const u64,
                           n_cols: usize, n: usize, out_jac: *mut u64) -> c_int;
    pub fn pz_msm_g1_dev(ctx: *mut pz_ctx, bases: *const pz_bases, d_scalars: *const u64, n_cols: usize,
                         n: usize, col_stride: usize, win_lo: u32, win_hi: u32, d_out_jac: *mut u64) -> c_int;
    pub fn pz_g1_sum(ctx: *mut pz_ctx, jac: *const u64, n: usize, out_jac: *mut u64) -> c_int;
    pub fn pz_g1_sum_dev(ctx: *mut pz_ctx, d_jac: *const u64, n: usize, d_out_jac: *mut u64) -> c_int;
    pub fn pz_msm_g1_multi(ctxs: *const *mut pz_ctx, bases: *const *const pz_bases, d_scalars: *const *const u64, n_per_ctx: *const usize,
                           n_ctx: usize, split_points: c_int, out_jac: *mut u64) -> c_int;   // one MSM over N contexts / GPUs
    pub fn pz_g1_normalize(ctx: *mut pz_ctx, jac: *const u64, n: usize, aff: *mut u64) -> c_int;
    pub fn pz_g1_fixed_base_mul(ctx: *mut pz_ctx, scalars: *const u64, n: usize, out_affine: *mut u64) -> c_int;
    pub fn pz_g1_fixed_base_mul_dev(ctx: *mut pz_ctx, d_scalars: *const u64, n: usize, d_out: *mut u64) -> c_int;

    // K2  (replaces halo2curves::fft::best_fft)
    pub fn pz_ntt_fr(ctx: *mut pz_ctx, a: *mut u64, omega: *const u64 /* [4] */, log_n: u32) -> c_int;
    pub fn pz_ntt_fr_batch(ctx: *mut pz_ctx, cols: *const *mut u64, n_cols: usize, omega: *const u64,
                           log_n: u32) -> c_int;
    pub fn pz_ntt_fr_dev(ctx: *mut pz_ctx, d_a: *mut u64, n_cols: usize, col_stride: usize, omega: *const u64,
                         log_n: u32, pre_coset_g: *const u64, post_scale: *const u64) -> c_int;
    pub fn pz_ntt_fr_to_dev(ctx: *mut pz_ctx, d_in: *const u64, in_stride: usize, d_out: *mut u64, out_stride: usize, n_cols: usize,
                            omega: *const u64, log_n: u32, pre_coset_g: *const u64, post_scale: *const u64) -> c_int;
    pub fn pz_ntt_fr_extend_dev(ctx: *mut pz_ctx, d_coeff: *const u64, n_cols: usize, in_stride: usize, d_ext: *mut u64,
                                out_stride: usize, log_n: u32, log_e: u32, omega_n: *const u64, coset_gens: *const u64,
                                scale: *const u64) -> c_int;
    pub fn pz_ntt_fr_coeff_extend_dev(ctx: *mut pz_ctx, d_values: *mut u64, n_cols: usize, col_stride: usize, d_ext: *mut u64,
                                      out_stride: usize, log_n: u32, log_e: u32, omega_n: *const u64, omega_n_inv: *const u64,
                                      n_inv: *const u64, coset_gens: *const u64) -> c_int;
    pub fn pz_fr_convert_dev(ctx: *mut pz_ctx, d_a: *mut u64, n: usize, to_mont: c_int) -> c_int;
    pub fn pz_fr_from_mask_dev(ctx: *mut pz_ctx, d_mask: *const u8, n: usize, d_out: *mut u64) -> c_int;

    // K3  (replaces num-bigint mul / div_rem inside BigUintChip::{mul_mod, pow_mod_fixed_exp})
    pub fn pz_mul_mod(ctx: *mut pz_ctx, limbs: u32, a: *const u64, b: *const u64, modulus: *const u64,
                      q: *mut u64, r: *mut u64) -> c_int;
    pub fn pz_paillier_trace(ctx: *mut pz_ctx, limbs_n2: u32, n2: *const u64, base: *const u64, exp: *const u64,
                             exp_limbs: u32, steps_out: *mut u64, n_steps: *mut usize, result: *mut u64) -> c_int;
    pub fn pz_paillier_encrypt(ctx: *mut pz_ctx, limbs_n: u32, batch: usize, n: *const u64, g: *const u64,
                               m: *const u64, r: *const u64, steps_out: *mut u64, steps_cap: usize,
                               n_steps_g: *mut u32, n_steps_r: *mut u32, c_out: *mut u64) -> c_int;
    pub fn pz_paillier_encrypt_dev(ctx: *mut pz_ctx, limbs_n: u32, batch: usize, n: *const u64, g: *const u64,
                                   m: *const u64, r: *const u64, d_steps_out: *mut u64, steps_cap: usize,
                                   n_steps_g: *mut u32, n_steps_r: *mut u32, c_out: *mut u64) -> c_int;

    // uniform-shape variant (NOT the reference's circuit) -- g^m over m_bits in-circuit exponent bits
    pub fn pz_paillier_encrypt_uniform(ctx: *mut pz_ctx, limbs_n: u32, batch: usize, m_bits: u32, n: *const u64, g: *const u64,
                                       m: *const u64, r: *const u64, steps_out: *mut u64, steps_cap: usize,
                                       n_steps_g: *mut u32, n_steps_r: *mut u32, c_out: *mut u64) -> c_int;
    pub fn pz_paillier_encrypt_uniform_dev(ctx: *mut pz_ctx, limbs_n: u32, batch: usize, m_bits: u32, n: *const u64,
                                           g: *const u64, m: *const u64, r: *const u64, d_steps_out: *mut u64, steps_cap: usize,
                                           n_steps_g: *mut u32, n_steps_r: *mut u32, c_out: *mut u64) -> c_int;

    // SRS / keygen (next rows, rank 2)
    pub fn pz_srs_lagrange_from_monomial_dev(ctx: *mut pz_ctx, k: u32, omega_inv: *const u64, n_inv: *const u64,
                                             d_g: *const u64, d_g_lagrange: *mut u64) -> c_int;
    pub fn pz_permutation_sigma_dev(ctx: *mut pz_ctx, d_map_col: *const u32, d_map_row: *const u32, m: usize, k: u32,
                                    omega: *const u64, delta: *const u64, d_sigma: *mut u64, sigma_stride: usize) -> c_int;
    pub fn pz_keygen_columns_dev(ctx: *mut pz_ctx, bases_lagrange: *const pz_bases, d_cols: *mut u64, n_cols: usize,
                                 col_stride: usize, k: u32, log_e: u32, omega_n: *const u64, omega_n_inv: *const u64,
                                 n_inv: *const u64, coset_gens: *const u64, d_commit_jac: *mut u64, d_ext: *mut u64,
                                 ext_stride: usize) -> c_int;

    // SHPLONK multi-point opening (next rows, rank 3)
    pub fn pz_shplonk_begin_dev(ctx: *mut pz_ctx, n: usize, n_sets: u32, set_n_polys: *const u32, d_polys: *const *const u64,
                                set_n_points: *const u32, point_idx: *const u32, n_points_total: u32, points: *const u64,
                                evals: *const u64, y: *const u64, v: *const u64, d_h: *mut u64,
                                state: *mut *mut pz_shplonk) -> c_int;
    pub fn pz_shplonk_finish_dev(ctx: *mut pz_ctx, state: *mut pz_shplonk, u: *const u64, d_h: *const u64, d_h2: *mut u64) -> c_int;
    pub fn pz_shplonk_free(ctx: *mut pz_ctx, state: *mut pz_shplonk) -> c_int;

    // patch point D as entry points: keygen + create_proof, one call per transcript round (section 5d)
    pub fn pz_circuit_structure_dev(ctx: *mut pz_ctx, kind: c_int, limbs_n: u32, limb_bits: u32, lookup_bits: u32, k: u32, exp_g: *const u64,
                                    exp_r: *const u64, minimum_rows: usize, blinding_factors: u32, out: *mut *mut pz_structure) -> c_int;
    pub fn pz_structure_info(st: *const pz_structure, n_adv: *mut usize, n_adv_filled: *mut usize, n_lk: *mut usize, max_rows: *mut usize,
                             n_constants: *mut usize, n_cells: *mut usize, n_lookups: *mut usize, n_steps_g: *mut usize,
                             n_steps_r: *mut usize) -> c_int;
    pub fn pz_structure_arrays(st: *const pz_structure, d_selectors: *mut *const u8, d_map_col: *mut *const u32, d_map_row: *mut *const u32,
                               d_col_starts: *mut *const u64, constants: *mut *const u64, col_starts_host: *mut *const u64) -> c_int;
    pub fn pz_structure_free(st: *mut pz_structure) -> c_int;
    pub fn pz_pk_create(ctx: *mut pz_ctx, bases_lagrange: *const pz_bases, bases_monomial: *const pz_bases, k: u32, lookup_bits: u32,
                        blinding_factors: u32, max_rows: usize, n_adv: usize, n_lk: usize, selectors: *const u8,
                        constants: *const u64, n_constants: usize, map_col: *const u32, map_row: *const u32, tile: usize,
                        ext_resident_cols: usize, out: *mut *mut pz_pk) -> c_int;
    pub fn pz_pk_create_dev(ctx: *mut pz_ctx, bases_lagrange: *const pz_bases, bases_monomial: *const pz_bases, k: u32, lookup_bits: u32,
                            blinding_factors: u32, max_rows: usize, n_adv: usize, n_lk: usize, d_selectors: *const u8,
                            constants: *const u64, n_constants: usize, d_map_col: *const u32, d_map_row: *const u32, tile: usize,
                            ext_resident_cols: usize, out: *mut *mut pz_pk) -> c_int;
    pub fn pz_pk_info(pk: *const pz_pk, n_fixed: *mut usize, n_perm_cols: *mut usize, n_sets: *mut usize, blinding_words: *mut usize,
                      evals_words: *mut usize) -> c_int;
    pub fn pz_pk_commitments(pk: *const pz_pk, fixed_affine: *mut u64, sigma_affine: *mut u64) -> c_int;
    pub fn pz_pk_free(pk: *mut pz_pk) -> c_int;
    pub fn pz_proof_begin(pk: *mut pz_pk, d_cols: *mut u64, seed: u64, blinding: *const u64, n_blinding: usize, out: *mut *mut pz_proof,
                          advice_affine: *mut u64) -> c_int;
    pub fn pz_proof_lookups(proof: *mut pz_proof, theta: *const u64, perm_inputs_affine: *mut u64, perm_tables_affine: *mut u64) -> c_int;
    pub fn pz_proof_products(proof: *mut pz_proof, beta: *const u64, gamma: *const u64, perm_z_affine: *mut u64,
                             lookup_z_affine: *mut u64, random_affine: *mut u64) -> c_int;
    pub fn pz_proof_quotient(proof: *mut pz_proof, y: *const u64, h_affine: *mut u64) -> c_int;
    pub fn pz_proof_evaluate(proof: *mut pz_proof, x: *const u64, evals: *mut u64) -> c_int;
    pub fn pz_proof_open_begin(proof: *mut pz_proof, y: *const u64, v: *const u64, w1_affine: *mut u64) -> c_int;
    pub fn pz_proof_open_finish(proof: *mut pz_proof, u: *const u64, w2_affine: *mut u64, quotient_degree_ok: *mut c_int) -> c_int;
    pub fn pz_proof_free(proof: *mut pz_proof) -> c_int;

    // next rows (SRS setup, evaluation at a point)
    pub fn pz_srs_setup_g1_dev(ctx: *mut pz_ctx, k: u32, s: *const u64, omega: *const u64, d_g: *mut u64,
                               d_g_lagrange: *mut u64) -> c_int;
    pub fn pz_g1_check_dev(ctx: *mut pz_ctx, d_points: *const u64, n: usize, n_bad: *mut u64) -> c_int;
    pub fn pz_poly_eval_dev(ctx: *mut pz_ctx, d_coeffs: *const u64, n_cols: usize, col_stride: usize, n: usize,
                            x: *const u64, d_out: *mut u64) -> c_int;
    pub fn pz_poly_eval_multi_dev(ctx: *mut pz_ctx, d_coeffs: *const u64, n_cols: usize, col_stride: usize, n: usize,
                                  xs: *const u64, n_points: u32, d_out: *mut u64) -> c_int;

    // next rows: grand products, custom-gate quotient, openings (create_proof internals)
    pub fn pz_fr_batch_invert_dev(ctx: *mut pz_ctx, d_a: *mut u64, n: usize) -> c_int;
    pub fn pz_fr_prefix_product_dev(ctx: *mut pz_ctx, d_a: *const u64, n: usize, z0: *const u64, d_z: *mut u64) -> c_int;
    pub fn pz_permutation_product_dev(ctx: *mut pz_ctx, d_cols: *const u64, col_stride: usize, d_sigma: *const u64,
                                      sigma_stride: usize, m: usize, log_n: u32, omega: *const u64, beta: *const u64,
                                      gamma: *const u64, delta_start: *const u64, delta: *const u64, z0: *const u64,
                                      d_z: *mut u64) -> c_int;
    pub fn pz_lookup_permute_dev(ctx: *mut pz_ctx, d_inputs: *const u64, n_cols: usize, col_stride: usize, d_table: *const u64,
                                 rows: usize, value_bits: u32, d_perm_inputs: *mut u64, d_perm_tables: *mut u64,
                                 out_stride: usize) -> c_int;
    pub fn pz_lookup_product_dev(ctx: *mut pz_ctx, d_inputs: *const u64, input_stride: usize, d_table: *const u64,
                                 d_perm_inputs: *const u64, perm_input_stride: usize, d_perm_tables: *const u64,
                                 perm_table_stride: usize, n_lookups: usize, n: usize, beta: *const u64, gamma: *const u64,
                                 z0: *const u64, d_z: *mut u64, z_stride: usize) -> c_int;
    pub fn pz_permutation_product_sets_dev(ctx: *mut pz_ctx, d_cols: *const u64, col_stride: usize, d_sigma: *const u64,
                                           sigma_stride: usize, m: usize, chunk_len: u32, log_n: u32, usable_rows: usize,
                                           omega: *const u64, beta: *const u64, gamma: *const u64, delta: *const u64,
                                           d_z: *mut u64, z_stride: usize) -> c_int;
    pub fn pz_quotient_gate_dev(ctx: *mut pz_ctx, d_adv_ext: *const u64, adv_stride: usize, d_sel_ext: *const u64,
                                sel_stride: usize, n_cols: usize, log_ext: u32, rot_step: u32, y: *const u64,
                                d_h: *mut u64) -> c_int;
    pub fn pz_quotient_permutation_dev(ctx: *mut pz_ctx, d_cols_ext: *const u64, col_stride: usize, d_sigma_ext: *const u64,
                                       sigma_stride: usize, d_z_ext: *const u64, z_stride: usize, n_sets: u32, chunk_len: u32,
                                       m_total: u32, log_ext: u32, rot_step: u32, last_rotation: u32, d_l0: *const u64,
                                       d_l_last: *const u64, d_l_active: *const u64, beta: *const u64, gamma: *const u64,
                                       delta: *const u64, coset_g: *const u64, omega_ext: *const u64, y: *const u64,
                                       d_h: *mut u64) -> c_int;
    pub fn pz_quotient_permutation_part_dev(ctx: *mut pz_ctx, d_cols_ext: *const u64, col_stride: usize, d_sigma_ext: *const u64,
                                            sigma_stride: usize, d_z_ext: *const u64, z_stride: usize, n_sets_total: u32,
                                            set_lo: u32, n_sets: u32, chunk_len: u32, m_cols: u32, head: c_int, log_ext: u32,
                                            rot_step: u32, last_rotation: u32, d_l0: *const u64, d_l_last: *const u64,
                                            d_l_active: *const u64, beta: *const u64, gamma: *const u64, delta: *const u64,
                                            coset_g: *const u64, omega_ext: *const u64, y: *const u64, d_h: *mut u64) -> c_int;
    pub fn pz_quotient_lookup_dev(ctx: *mut pz_ctx, d_input_ext: *const u64, input_stride: usize, d_table_ext: *const u64,
                                  d_perm_input_ext: *const u64, perm_input_stride: usize, d_perm_table_ext: *const u64,
                                  perm_table_stride: usize, d_z_ext: *const u64, z_stride: usize, n_lookups: u32, log_ext: u32,
                                  rot_step: u32, d_l0: *const u64, d_l_last: *const u64, d_l_active: *const u64,
                                  beta: *const u64, gamma: *const u64, y: *const u64, d_h: *mut u64) -> c_int;
    pub fn pz_quotient_finish_dev(ctx: *mut pz_ctx, d_h: *mut u64, log_n: u32, log_e: u32, coset_g: *const u64,
                                  omega_ext: *const u64) -> c_int;
    pub fn pz_fr_distribute_powers_dev(ctx: *mut pz_ctx, d_a: *mut u64, n_cols: usize, col_stride: usize, n: usize,
                                       g: *const u64, c: *const u64) -> c_int;
    pub fn pz_fr_lincomb_dev(ctx: *mut pz_ctx, d_polys: *const u64, n_cols: usize, col_stride: usize, n: usize, v: *const u64,
                             d_out: *mut u64, accumulate: c_int) -> c_int;
    pub fn pz_poly_div_linear_dev(ctx: *mut pz_ctx, d_coeffs: *const u64, n_cols: usize, col_stride: usize, n: usize,
                                  x: *const u64, d_q: *mut u64, q_stride: usize) -> c_int;

    // K4
    pub fn pz_witness_cells_per_step(limbs: u32, limb_bits: u32, lookup_bits: u32, advice_cells: *mut usize,
                                     lookup_cells: *mut usize) -> c_int;
    pub fn pz_witness_expand(ctx: *mut pz_ctx, limbs: u32, limb_bits: u32, lookup_bits: u32, steps: *const u64, n_steps: usize,
                             modulus: *const u64, advice_out: *mut u64, lookup_out: *mut u64) -> c_int;
    pub fn pz_witness_expand_dev(ctx: *mut pz_ctx, limbs: u32, limb_bits: u32, lookup_bits: u32,
                                 d_steps: *const u64, n_steps: usize, d_modulus: *const u64, d_advice: *mut u64,
                                 d_lookup: *mut u64) -> c_int;
    // K4, the WHOLE circuit of a driver (bench.rs:33-75 kind 0, :77-117 kind 1; kind 2 = the uniform-shape encrypt circuit)
    pub fn pz_circuit_cells(kind: c_int, limbs_n: u32, limb_bits: u32, lookup_bits: u32, n_steps_g: usize, n_steps_r: usize,
                            advice_cells: *mut usize, lookup_cells: *mut usize) -> c_int;
    pub fn pz_circuit_expand_dev(ctx: *mut pz_ctx, kind: c_int, limbs_n: u32, limb_bits: u32, lookup_bits: u32,
                                 inputs: *const u64, d_steps: *const u64, n_steps_g: usize, n_steps_r: usize,
                                 d_modulus: *const u64, d_advice: *mut u64, d_lookup: *mut u64, rows: usize,
                                 col_stride: usize) -> c_int;
    pub fn pz_circuit_break_points(gate_mask: *const u8, n_cells: usize, max_rows: usize, starts_out: *mut u64, capacity: usize,
                                   n_cols: *mut usize) -> c_int;
    pub fn pz_circuit_expand_cols_dev(ctx: *mut pz_ctx, kind: c_int, limbs_n: u32, limb_bits: u32, lookup_bits: u32,
                                      inputs: *const u64, d_steps: *const u64, n_steps_g: usize, n_steps_r: usize,
                                      d_modulus: *const u64, d_advice: *mut u64, d_lookup: *mut u64, d_col_starts: *const u64,
                                      n_adv_cols: usize, max_rows: usize, lookup_rows: usize, col_stride: usize) -> c_int;
    pub fn pz_refresh_aux(limb_bits: u32, num_limbs_l: u32, num_limbs_r: u32, increased_limbs: *mut u8, capacity: u32,
                          n_out: *mut u32) -> c_int;
    pub fn pz_op_cells(op: c_int, limbs: u32, limb_bits: u32, lookup_bits: u32, advice_cells: *mut usize,
                       lookup_cells: *mut usize) -> c_int;

    // measurement helpers (bench.py / profiles only; a prover does not need them)
    pub fn pz_timing_enable(ctx: *mut pz_ctx, on: c_int) -> c_int;
    pub fn pz_timing_reset(ctx: *mut pz_ctx) -> c_int;
    pub fn pz_timing_get(ctx: *mut pz_ctx, which: c_int, total_ms: *mut f64, launches: *mut u64) -> c_int;
    // (the issue-rate microbenchmarks are not part of this ABI: libpz_probe.so, paillier_halo2_amd/probe.py)
}

// layout guards: the ABI takes halo2curves' in-memory representation verbatim
const _: () = assert!(core::mem::size_of::<halo2curves::bn256::Fr>() == 32);
const _: () = assert!(core::mem::size_of::<halo2curves::bn256::Fq>() == 32);
const _: () = assert!(core::mem::size_of::<halo2curves::bn256::G1Affine>() == 64);
const _: () = assert!(core::mem::size_of::<halo2curves::bn256::G1>() == 96);
const _: () = assert!(core::mem::align_of::<halo2curves::bn256::Fr>() == 8);
