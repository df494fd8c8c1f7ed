/* pz.h -- C ABI of the MI355X-native (gfx950, hand-written HIP) hot path of the
 * Paillier-in-Halo2 prover.  This header is the whole drop-in boundary: plain pointers and sizes,
 * no C++/torch types.  Library: paillier_halo2_amd/csrc/libpz_hip.so.
 *
 * The reference (aerius-labs/paillier-halo2) is a plain Rust library with no FFI seam of its own
 * (src/lib.rs:1-2); every entry point below names the reference call site / dependency routine it
 * replaces.  INTEGRATION.md shows the Rust `extern "C"` stub a maintainer adds for each.
 *
 * Conventions (all functions):
 *   - return 0 (PZ_OK) on success, a negative pz_status on error; never throw, never abort, no
 *     callbacks; there is NO CPU fallback -- without a usable gfx950 device pz_init fails.
 *   - field elements: 4 x u64 little-endian limbs in MONTGOMERY form (R = 2^256), exactly the
 *     in-memory layout of halo2curves Fr / Fq.  G1Affine = {x, y} (64 B), identity = all-zero.
 *     G1 Jacobian = {x, y, z} (96 B), identity z = 0 (returned as (0, R, 0)).
 *   - big integers: little-endian u64 limb arrays (limb 0 least significant), i.e. the limb order
 *     of AssignedBigUint::limbs() at limb_bits = 64 (paillier.rs:22-30).
 *   - "host" pointers are ordinary process memory; "_dev" entry points take device pointers
 *     (hipMalloc / torch tensor data_ptr) valid on the context's device and run asynchronously on
 *     the context's stream (pz_set_stream); host-pointer entry points synchronise before returning.
 *   - every field element / coordinate handed to the library must be CANONICAL (an integer below the modulus in
 *     Montgomery form), as halo2curves guarantees for Fr / Fq values; limbs in [p, 2^256) are not reduced on entry
 *     and give undefined (silently wrong) results.  pz_g1_check_dev validates points on request.
 *   - pz_set_stream orders the new stream after the work already queued on the previous one (event wait), so cached
 *     tables / workspaces stay consistent; the caller still orders its OWN buffers between streams.
 *   - a pz_ctx is bound to ONE device (one process per GPU, ranks joined by RCCL above this ABI);
 *     distinct contexts may be used concurrently; one context is serialised INTERNALLY (a recursive mutex held for
 *     the duration of every entry point), and entry points call hipSetDevice themselves, so they may be called
 *     from any thread -- e.g. halo2's rayon workers reaching a patched best_multiexp with one shared context.
 */
#ifndef PZ_H
#define PZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pz_ctx pz_ctx;
typedef struct pz_bases pz_bases;

enum pz_status {
    PZ_OK = 0,
    PZ_ERR_INVALID = -1,      /* null pointer, bad size, log_n out of range ...            */
    PZ_ERR_HIP = -2,          /* a HIP runtime call failed (see pz_last_hip_error)          */
    PZ_ERR_NO_DEVICE = -3,    /* no gfx950 device visible                                   */
    PZ_ERR_OOM = -4,          /* device allocation failed                                   */
    PZ_ERR_ZERO_MODULUS = -5, /* mul_mod / pow_mod with modulus 0 (reference: BigUint % 0 panics, paillier.rs:91) */
    PZ_ERR_RANGE = -6,        /* quotient does not fit the limb count the circuit assigns it (unsatisfiable in the reference) */
    PZ_ERR_UNSUPPORTED = -7,  /* n_devices != 1, limb count not a supported size ...         */
    PZ_ERR_CAPACITY = -8,     /* caller-provided output capacity too small                  */
    PZ_ERR_MESSAGE_RANGE = -9,/* uniform-shape circuit: a message does not fit the m_bits the circuit decomposes */
    PZ_ERR_ASYNC = -10        /* reported by a synchronising entry point (pz_sync, pz_download, the host-pointer MSM calls): an
                                 earlier asynchronous pz_msm_g1* call found its scalars changed while it ran, or
                                 pz_permutation_sigma_dev found an image outside its m x n cells; the outputs of that call are
                                 invalid (no out-of-bounds access took place: positions are checked on the device) */
    ,
    PZ_ERR_INTERNAL = -11     /* a device-side wait ran into its bound.  Only source today: K3 (pz_paillier_trace / pz_paillier_encrypt*):
                                 a chain is worked by two workgroups, and a multiplier workgroup waits for each square its squarer
                                 publishes; that wait is bounded (2^27 polls, tens of seconds) so that every wave has an exit whatever
                                 happens to the other workgroup.  For the caller: the call has returned cleanly, c_out / result /
                                 n_steps of the affected chains were NOT written and their step records are incomplete -- discard all
                                 outputs of the call; the context and its device memory are intact and the same call may simply be
                                 repeated.  It does not occur in normal operation (roles are handed out by arrival, so a multiplier's
                                 squarer has always started before it; see csrc/pz_bigint.hip k_pow_mod_chain); it is the defined
                                 outcome if the device stops scheduling a started workgroup. */
};

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */
/* n_devices must be 1; device_ids[0] is the HIP device ordinal (NULL -> device 0). */
int pz_init(int n_devices, const int* device_ids, pz_ctx** out);
int pz_free(pz_ctx* ctx);
const char* pz_strerror(int status);
/* last HIP error string recorded by this context (empty string if none) */
const char* pz_last_hip_error(const pz_ctx* ctx);
/* stream used by the _dev entry points (NULL = the context's own stream). */
int pz_set_stream(pz_ctx* ctx, void* hip_stream);
int pz_sync(pz_ctx* ctx);
/* explicit ABI version, bumped whenever an entry point below is added, removed or changes meaning (measurement probes are
 * not part of this ABI: they live in libpz_probe.so).  A binding compares it with the PZ_ABI_VERSION it was built against. */
#define PZ_ABI_VERSION 7
int pz_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * device memory -- for a host that owns no HIP runtime of its own (the reference's Rust prover behind this FFI; the C++
 * driver in paillier_halo2_amd/host/prove_c2.cpp).  The `_dev` entry points take these pointers as they are, so a proof's
 * columns can stay in HBM from K4 through K1 and K2 (patch points C / D of INTEGRATION.md: bench.rs:161-171's create_proof).
 * ------------------------------------------------------------------------------------------- */
/* (a request that does not fit makes the library give back what it holds as caches -- cached blocks, then the grow-only sort / partial-sum
 * workspaces of pz_msm_g1*, which the next MSM sizes anew for what is free then -- before it returns PZ_ERR_OOM) */
int pz_dev_alloc(pz_ctx* ctx, size_t bytes, void** d_out);
/* block cache of pz_dev_alloc / pz_dev_free (off by default: max_bytes = 0): freed blocks of 32 MiB and more are KEPT by the context, up
 * to max_bytes in all, and handed to the next pz_dev_alloc of that size (or up to an eighth smaller) instead of going back to the driver.
 * For a caller that proves message after message of the reference's circuit -- a new 116-GB proving key each (paillier.rs:50-55 makes
 * every message its own circuit) -- the driver's allocation cost (seconds per key) is then paid once: pz_pk_create / pz_pk_free allocate
 * through these functions, with column counts rounded up to 64 so that keys of nearly equal shape ask for identical sizes.  Lowering the
 * limit releases what exceeds it; the library releases the cache itself when one of its own allocations runs out of memory. */
int pz_dev_cache_limit(pz_ctx* ctx, size_t max_bytes);
/* the ARENA: one block of `bytes` reserved from the driver NOW, out of which this context's later allocations are carved -- pz_dev_alloc and
 * every buffer the library allocates for itself on this context (workspaces, tables, pz_load_bases*, pz_circuit_structure_dev's arrays and
 * temporaries, a proving key).  Freed blocks go back to the arena and coalesce; a request the arena cannot serve falls through to the
 * driver (and the block cache above).  For the caller pz_dev_cache_limit was made for, at the sizes where the cache stops helping: at
 * config c5 (3072-bit n, k = 19; bench.rs:161-171's keygen + proof per run) a step holds 250 GB, the cache has to be released between a
 * key and the next structure's temporaries, and half of an 11-s step was driver allocation calls; from an arena the same step makes none.
 * Blocks are 4 KiB-granular and NOT zeroed.  A block may be freed through any context.  bytes = 0 releases the arena (also pz_free does);
 * one that still holds live blocks -- PZ_ERR_INVALID -- is detached from the context and stays reserved until the last of them is freed.
 * Size it from pz_dev_mem_info (leave a few GiB to the runtime).  PZ_DEV_ARENA_POISON=1 in the environment fills the arena, and every
 * block again when it is freed, with 0xA5 (tests). */
int pz_dev_arena(pz_ctx* ctx, size_t bytes);
/* out: arena bytes, bytes in live blocks, the peak of that, the largest free hole, requests served, requests it could not serve (all 0
 * without an arena) */
int pz_dev_arena_info(pz_ctx* ctx, uint64_t out[6]);
/* the driver's free / total device memory (hipMemGetInfo; waits for the device's queued work on ROCm).  Either pointer may be NULL */
int pz_dev_mem_info(pz_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* waits for the context's queued work, then frees */
int pz_dev_free(pz_ctx* ctx, void* d);
/* host -> device on the context's stream; returns when the host buffer may be reused */
int pz_upload(pz_ctx* ctx, void* d_dst, const void* src, size_t bytes);
/* device -> host on the context's stream; returns when the data has arrived (everything queued before it has run) */
int pz_download(pz_ctx* ctx, void* dst, const void* d_src, size_t bytes);
/* asynchronous fill on the context's stream */
int pz_dev_memset(pz_ctx* ctx, void* d_dst, int byte_value, size_t bytes);
/* asynchronous device -> device copy on the context's stream (e.g. a column's Lagrange values into the buffer its coefficient
 * form is computed in) */
int pz_dev_copy(pz_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* the same for `rows` runs of `width` bytes, the runs dst_pitch / src_pitch bytes apart (a prover fills the blinding rows at the end
 * of every column of a proof from one staged block of random elements) */
int pz_dev_copy_2d(pz_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width, size_t rows);
/* order two contexts of one device: everything queued on `producer`'s stream so far happens before whatever `waiter` queues
 * from now on (an event wait, no host synchronisation) */
int pz_ctx_wait(pz_ctx* waiter, pz_ctx* producer);

/* ---------------------------------------------------------------------------------------------
 * K1 -- G1 multi-scalar multiplication.  Replaces halo2curves `best_multiexp(coeffs, bases)`
 * reached from /root/reference/src/bench.rs:161-171 (bench_builder -> create_proof ->
 * ParamsKZG::commit_lagrange / commit).  The VALUE (the group element) equals best_multiexp's;
 * the Jacobian representative is not unique, compare after pz_g1_normalize (halo2 itself writes
 * commitments to the transcript in affine form after batch_normalize).
 * ------------------------------------------------------------------------------------------- */
/* Upload 2^k affine bases (host) and build the device-resident window-shifted table
 * T[w][i] = 2^(c*w) * P_i.  `lagrange` only labels the set (g vs g_lagrange of ParamsKZG).
 * window_bits = 0 picks the default for k. */
int pz_srs_load_g1(pz_ctx* ctx, uint32_t k, const uint64_t* bases_affine, int lagrange, pz_bases** out);
/* general form: n_points need not be a power of two; bases may already be on the device. */
int pz_bases_load_g1(pz_ctx* ctx, const uint64_t* bases_affine, size_t n_points, int on_device,
                     uint32_t window_bits, pz_bases** out);
int pz_bases_free(pz_ctx* ctx, pz_bases* bases);
/* window parameters of a loaded set: c = window bits, n_windows = floor(253/c)+1 */
int pz_bases_info(const pz_bases* bases, size_t* n_points, uint32_t* window_bits, uint32_t* n_windows);

/* == best_multiexp(scalars[0..n], bases[0..n]);  n <= n_points.  Host pointers. */
int pz_msm_g1(pz_ctx* ctx, const pz_bases* bases, const uint64_t* scalars, size_t n, uint64_t out_jac[12]);
/* n_cols column commitments against the same bases (commit_lagrange per advice column).  Host pointers (pageable or
 * page-locked); the columns are streamed through device staging buffers in groups, uploads on a copy stream of the
 * library's own beside the kernels of the previous group; the call returns when out_jac is complete. */
int pz_msm_g1_batch(pz_ctx* ctx, const pz_bases* bases, const uint64_t* const* scalar_cols, size_t n_cols,
                    size_t n, uint64_t* out_jac /* n_cols x 12 */);
/* device-resident form: scalars = n_cols columns, column j at d_scalars + j*col_stride (in u64
 * units, >= 4*n); result n_cols x 12 u64 on the device.  Only Pippenger windows
 * [win_lo, win_hi) are accumulated (pass 0, n_windows for the full MSM) -- the multi-GPU split of
 * one large MSM gives each rank a disjoint window range and sums the partial points. */
int pz_msm_g1_dev(pz_ctx* ctx, const pz_bases* bases, const uint64_t* d_scalars, size_t n_cols, size_t n,
                  size_t col_stride, uint32_t win_lo, uint32_t win_hi, uint64_t* d_out_jac);
/* sum of n Jacobian points (host in/out): the fixed-order fold of per-rank partial MSMs. */
int pz_g1_sum(pz_ctx* ctx, const uint64_t* jac /* n x 12 */, size_t n, uint64_t out_jac[12]);
/* device form: d_jac (n x 12) and d_out_jac (12) are device pointers; asynchronous on the context's stream.  d_out_jac
 * must not alias d_jac.  Used after the all-gather of the sharded MSM: every rank folds the same points in rank order. */
int pz_g1_sum_dev(pz_ctx* ctx, const uint64_t* d_jac, size_t n, uint64_t* d_out_jac);
/* ONE multi-scalar multiplication over n_ctx contexts -- one per GPU of a node (several on one device work too) -- in one call:
 * north_star's "disjoint Pippenger windows of one large MSM sharded across the GPUs, partial sums folded" for a single-process host.
 *   split_points == 0: context r accumulates windows [r W / n_ctx, (r + 1) W / n_ctx) of ALL n_per_ctx[r] (== n) scalars; it holds all
 *                      scalars (d_scalars[r]) and a table of all bases (bases[r], the same window width everywhere);
 *   split_points != 0: context r takes its own point range: d_scalars[r] holds its n_per_ctx[r] scalars, bases[r] its n_per_ctx[r]
 *                      bases (SURVEY 8e's alternative: 1 / n_ctx of the table memory and scalar reads per GPU).
 * Every context's share is queued on its own stream (the call does not serialise them), the 96-byte partial points are then read
 * back in rank order and folded on ctxs[0] in that fixed order.  Blocks until out_jac is there.  d_scalars[r] / bases[r] belong to
 * ctxs[r]'s device.  Across PROCESSES (one per GPU) the same is pz_msm_g1_dev + an exchange of the partials + pz_g1_sum[_dev]. */
int pz_msm_g1_multi(pz_ctx* const* ctxs, const pz_bases* const* bases, const uint64_t* const* d_scalars, const size_t* n_per_ctx,
                    size_t n_ctx, int split_points, uint64_t out_jac[12]);
/* Jacobian -> affine for n points (host in/out): `batch_normalize`. */
int pz_g1_normalize(pz_ctx* ctx, const uint64_t* jac /* n x 12 */, size_t n, uint64_t* aff /* n x 8 */);
/* out[i] = [scalars[i]] * G1 generator, affine; scalars are Fr Montgomery (host in/out).  This is
 * the fixed-base multiplication `ParamsKZG::setup` uses for g[i] = [s^i]G; also used to build
 * synthetic base sets with known discrete logs. */
int pz_g1_fixed_base_mul(pz_ctx* ctx, const uint64_t* scalars, size_t n, uint64_t* out_affine /* n x 8 */);
/* device form of the same; out_affine is a device pointer */
int pz_g1_fixed_base_mul_dev(pz_ctx* ctx, const uint64_t* d_scalars, size_t n, uint64_t* d_out_affine);

/* ---------------------------------------------------------------------------------------------
 * K2 -- radix-2 NTT over Fr.  Replaces halo2curves `best_fft(a, omega, log_n)` reached from the
 * same call site through halo2-axiom EvaluationDomain::{ifft, fft, coeff_to_extended,
 * extended_to_coeff}:  a[k] <- sum_j a[j] * omega^(j k), natural order in and out, in place.
 * ------------------------------------------------------------------------------------------- */
int pz_ntt_fr(pz_ctx* ctx, uint64_t* a /* 2^log_n x 4 */, const uint64_t omega[4], uint32_t log_n);
/* n_cols transforms in place in HOST memory: column groups are uploaded, transformed and downloaded concurrently (two copy
 * streams of the library's own + the context's stream); the call returns when every column is back. */
int pz_ntt_fr_batch(pz_ctx* ctx, uint64_t* const* cols, size_t n_cols, const uint64_t omega[4], uint32_t log_n);
/* device-resident, n_cols columns at d_a + j*col_stride (u64 units).  Optional fused steps
 * (either may be NULL):
 *   pre_coset_g : a[i] *= g^i before the transform (distribute_powers of coeff_to_extended)
 *   post_scale  : a[k] *= s   after  the transform (the 1/n `ifft_divisor`)            */
int pz_ntt_fr_dev(pz_ctx* ctx, uint64_t* d_a, size_t n_cols, size_t col_stride, const uint64_t omega[4],
                  uint32_t log_n, const uint64_t* pre_coset_g, const uint64_t* post_scale);
/* the same out of place: column j is read at d_in + j*in_stride and its transform written at d_out + j*out_stride (d_in is not
 * modified; d_out == d_in with equal strides is pz_ntt_fr_dev).  lagrange_to_coeff of a committed column: the Lagrange values
 * stay where the commitment kernel reads them, the coefficient form gets its own buffer. */
int pz_ntt_fr_to_dev(pz_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint64_t* d_out, size_t out_stride, size_t n_cols,
                     const uint64_t omega[4], uint32_t log_n, const uint64_t* pre_coset_g, const uint64_t* post_scale);

/* coeff_to_extended in one call (n = 2^log_n coefficients per column -> 2^log_e * n evaluations on the coset):
 *   d_ext[col][2^log_e * q + r] = sum_i d_coeff[col][i] * scale * coset_gens[r]^i * omega_n^(i q)
 * coset_gens (host, 2^log_e x 4 limbs, Montgomery) = g * omega_ext^r; omega_n = omega_ext^(2^log_e) generates the
 * 2^log_n domain; scale may be NULL (1) -- pass the 1/n ifft divisor here when d_coeff still carries it.
 * log_n <= 18, log_e <= 3.  Equals zero-extending to 2^(log_n+log_e), distribute_powers(g) and best_fft(omega_ext). */
int pz_ntt_fr_extend_dev(pz_ctx* ctx, const uint64_t* d_coeff, size_t n_cols, size_t in_stride, uint64_t* d_ext,
                         size_t out_stride, uint32_t log_n, uint32_t log_e, const uint64_t omega_n[4],
                         const uint64_t* coset_gens, const uint64_t* scale);
/* in-place representation change of n Fr elements on the device: to_mont != 0: canonical little-endian
 * integers (must be < r) -> Montgomery form (Fr::from_raw); else Montgomery -> canonical (to_repr). */
/* Lagrange values -> coefficients (in place: lagrange_to_coeff = best_fft(omega_n^-1), then * n_inv) AND -> the extended
 * coset values of pz_ntt_fr_extend_dev, in one call: the final pass of the inverse transform is fused with the first pass
 * of the 2^log_e forward transforms, so the coefficients are written once and never read back (10 <= log_n <= 18; other
 * sizes run the two transforms back to back).  Results are identical to pz_ntt_fr_dev + pz_ntt_fr_extend_dev. */
int pz_ntt_fr_coeff_extend_dev(pz_ctx* ctx, uint64_t* d_values, size_t n_cols, size_t col_stride, uint64_t* d_ext,
                               size_t out_stride, uint32_t log_n, uint32_t log_e, const uint64_t omega_n[4],
                               const uint64_t omega_n_inv[4], const uint64_t n_inv[4], const uint64_t* coset_gens);
int pz_fr_convert_dev(pz_ctx* ctx, uint64_t* d_a, size_t n, int to_mont);
/* a byte mask -> field elements: d_out[i] = d_mask[i] != 0 ? 1 : 0 (Montgomery form), n elements, both on the device.  How selectors --
 * 0 / 1 bytes in halo2's Assembly -- become Lagrange-form fixed columns without crossing PCIe as 32-byte elements (keygen). */
int pz_fr_from_mask_dev(pz_ctx* ctx, const uint8_t* d_mask, size_t n, uint64_t* d_out);

/* ---------------------------------------------------------------------------------------------
 * K3 -- big-integer witness generation for g^m * r^n mod n^2.  Replaces the native
 * (num-bigint) part of biguint-halo2's BigUintChip::{mul_mod, pow_mod_fixed_exp}
 * (call sites /root/reference/src/paillier.rs:51,55,57,81) and paillier_enc_native /
 * paillier_add_native (paillier.rs:87-97).
 * A "step" is one mul_mod: (a, b, q, r) with a*b = q*modulus + r, 0 <= r < modulus, each `limbs`
 * u64 limbs, laid out a|b|q|r (4*limbs u64 per step).
 * ------------------------------------------------------------------------------------------- */
/* one mul_mod witness (PaillierChip::add's single step, paillier.rs:81) */
int pz_mul_mod(pz_ctx* ctx, uint32_t limbs, const uint64_t* a, const uint64_t* b, const uint64_t* modulus,
               uint64_t* q, uint64_t* r);
/* pow_mod_fixed_exp step trace: acc = 1, sq = base; for each bit of exp LSB->MSB: emit
 * (sq,sq,q,sq') ; if bit: emit (acc,sq_old,q,acc').  steps_out has room for *n_steps steps on
 * entry (needs bits(exp) + popcount(exp)); on return *n_steps = steps written. result = acc. */
int pz_paillier_trace(pz_ctx* ctx, uint32_t limbs_n2, const uint64_t* n2, const uint64_t* base, const uint64_t* exp,
                      uint32_t exp_limbs, uint64_t* steps_out, size_t* n_steps, uint64_t* result);
/* whole PaillierChip::encrypt witness (paillier.rs:32-60) for `batch` independent instances:
 * per instance n, g, m, r are limbs_n limbs (limbs_n2 = 2*limbs_n); n2 = n*n is formed on the
 * device; the g^m and r^n chains of every instance run concurrently; then c = gm*rn mod n2.
 * steps_out: per instance `steps_cap` steps (layout as above): first the g^m trace, then the r^n
 * trace, then the final mul_mod.  n_steps_g / n_steps_r (per instance) receive the trace lengths.
 * c_out: batch x 2*limbs_n limbs.  steps_out may be NULL (values only: the paillier_enc_native use). */
int pz_paillier_encrypt(pz_ctx* ctx, uint32_t limbs_n, size_t batch, const uint64_t* n, const uint64_t* g,
                        const uint64_t* m, const uint64_t* r, uint64_t* steps_out, size_t steps_cap,
                        uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out);
/* device-resident output form used by the prover pipeline (steps stay in HBM for K4) */
int pz_paillier_encrypt_dev(pz_ctx* ctx, uint32_t limbs_n, size_t batch, const uint64_t* n, const uint64_t* g,
                            const uint64_t* m, const uint64_t* r, uint64_t* d_steps_out, size_t steps_cap,
                            uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out);

/* Uniform-shape variant (NOT the reference's circuit): g^m as BigUintChip::pow_mod over exactly
 * m_bits in-circuit exponent bits -- per bit the step (acc, sq) [kept only if the bit is set: select] then the step (sq, sq)
 * -- so n_steps_g = 2 * m_bits for every message and one vk / pk serves every proof of a key.  r^n and the final mul_mod as
 * above.  PZ_ERR_MESSAGE_RANGE if a message does not fit m_bits.  Same buffer conventions as pz_paillier_encrypt[_dev]. */
int pz_paillier_encrypt_uniform(pz_ctx* ctx, uint32_t limbs_n, size_t batch, uint32_t m_bits, const uint64_t* n, const uint64_t* g,
                                const uint64_t* m, const uint64_t* r, uint64_t* steps_out, size_t steps_cap, uint32_t* n_steps_g,
                                uint32_t* n_steps_r, uint64_t* c_out);
int pz_paillier_encrypt_uniform_dev(pz_ctx* ctx, uint32_t limbs_n, size_t batch, uint32_t m_bits, const uint64_t* n,
                                    const uint64_t* g, const uint64_t* m, const uint64_t* r, uint64_t* d_steps_out, size_t steps_cap,
                                    uint32_t* n_steps_g, uint32_t* n_steps_r, uint64_t* c_out);

/* ---------------------------------------------------------------------------------------------
 * K4 -- expansion of a step trace into advice-column cells (Fr Montgomery, 32 B each), i.e. the
 * values halo2-lib's Context would hold after BigUintChip emitted the constraints of every
 * mul_mod (limb convolutions, range-check digit splits, carry chains).  Replaces the cell pushes
 * behind the dependency call sites paillier.rs:39-57, bench.rs:44-74.  Layout: DESIGN.md section 4.
 * ------------------------------------------------------------------------------------------- */
/* `limbs` = the circuit's limbs per big integer, each `limb_bits` wide (16..90: the reference uses 64 in
 * bench.rs:140 / paillier.rs:116 and 88 in its add test, paillier.rs:186-187).  Step records stay what K3 emits:
 * little-endian 64-bit words, words = ceil(limbs * limb_bits / 64) per big integer (== limbs at limb_bits 64);
 * K4 re-cuts them into limb_bits-wide limbs.  PZ_ERR_UNSUPPORTED outside 2*limb_bits + log2(limbs) + 3 <= 192. */
/* cells one mul_mod step expands to, for the given shape */
int pz_witness_cells_per_step(uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits, size_t* advice_cells,
                              size_t* lookup_cells);
/* d_steps: n_steps steps on the device (a|b|q|r, `words` 64-bit words each), d_modulus: `words` words on
 * the device; writes n_steps*advice_cells Fr elements to d_advice and n_steps*lookup_cells to
 * d_lookup (either may be NULL to skip). */
int pz_witness_expand_dev(pz_ctx* ctx, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits,
                          const uint64_t* d_steps, size_t n_steps, const uint64_t* d_modulus, uint64_t* d_advice,
                          uint64_t* d_lookup);

/* The WHOLE circuit's cell stream, operation by operation in the drivers' call order (bench.rs:33-75 paillier_enc_test,
 * kind = 0; bench.rs:77-117 paillier_enc_add_test, kind = 1): assign_integer of n, g, m|c1, r|c2; square + refresh of n
 * (paillier.rs:39-45 / 69-75); load_zero; for encrypt both pow_mod_fixed_exp chains (assign_constant(1), load_zero, their
 * mul_mod steps); the final mul_mod; assign_integer(res, 2*enc_bits); assert_equal_fresh.  Layout and per-operation cell
 * patterns: paillier_halo2_amd/layout.py::circuit_cells, DESIGN.md section 4 (dependency-derived, SURVEY tag [D]).
 * kind = 2: the uniform-shape encrypt circuit of pz_paillier_encrypt_uniform (g^m as pow_mod: num_to_bits of m's limbs, per
 * bit mul_mod + limb-wise select + square_mod; n_steps_g = 2 * limbs_n * limb_bits). */
int pz_circuit_cells(int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits, size_t n_steps_g,
                     size_t n_steps_r, size_t* advice_cells, size_t* lookup_cells);
/* inputs (HOST): n | g | x | y as ceil(limbs_n*limb_bits/64) 64-bit words each, then res as ceil(2*limbs_n*limb_bits/64)
 * words.  d_steps (device): the n_steps_g + n_steps_r + 1 step records K3 produced (add: one record); the circuit's
 * result c is the last record's remainder.  d_modulus: n^2 (device).  d_lookup may be NULL.
 * rows / col_stride cut the streams into the circuit's columns: cell c goes to column c / rows, row c % rows, columns
 * col_stride elements apart (rows = 2^k - blinding rows, col_stride = 2^k: every column is then a 2^k-row Lagrange vector
 * the commitment, product and NTT entry points take as it stands; rows above `rows` are not written).  0, 0 = dense. */
int pz_circuit_expand_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits,
                          const uint64_t* inputs, const uint64_t* d_steps, size_t n_steps_g, size_t n_steps_r,
                          const uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup, size_t rows, size_t col_stride);

/* The same stream with the ADVICE cells in halo2-lib's break-point column layout -- what the dependency's assign_with_constraints
 * [D] does with the Context's cell list when it fills the basic-gate columns (reached from bench.rs:161-171 through keygen /
 * create_proof -> synthesize): a 4-cell gate never straddles two columns; the column ends where one would, and the cell it ends
 * with is written AGAIN as row 0 of the next column (the two copies are tied by an equality constraint of the circuit).
 *   pz_circuit_break_points (host, no device work): from the selector mask of the stream (gate_mask[i] != 0 where a gate window
 *   starts at cell i) and max_rows (usable rows of a column) -> starts[j] = the stream index of column j's row 0, j < *n_cols, and
 *   starts[*n_cols] = n_cells.  starts_out may be NULL (count only); PZ_ERR_CAPACITY if capacity < *n_cols + 1; PZ_ERR_UNSUPPORTED if
 *   two gates overlap by other than three cells at a break (the dependency asserts the same).
 *   pz_circuit_expand_cols_dev: as pz_circuit_expand_dev, the advice stream placed by d_col_starts (DEVICE array of n_adv_cols + 1
 *   entries as above), columns col_stride elements apart; the lookup stream is cut plainly into columns of lookup_rows cells.
 *   Column j's rows above starts[j + 1] - starts[j] are not written. */
int pz_circuit_break_points(const uint8_t* gate_mask, size_t n_cells, size_t max_rows, uint64_t* starts_out, size_t capacity,
                            size_t* n_cols);
int pz_circuit_expand_cols_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits,
                               const uint64_t* inputs, const uint64_t* d_steps, size_t n_steps_g, size_t n_steps_r,
                               const uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup, const uint64_t* d_col_starts,
                               size_t n_adv_cols, size_t max_rows, size_t lookup_rows, size_t col_stride);

/* RefreshAux::new(limb_bits, l, r).increased_limbs_vec (paillier.rs:40-44): the number of further limbs each product limb's
 * maximal value spills into; *n_out entries (= the refreshed integer's limb count) are written. */
int pz_refresh_aux(uint32_t limb_bits, uint32_t num_limbs_l, uint32_t num_limbs_r, uint8_t* increased_limbs,
                   uint32_t capacity, uint32_t* n_out);
/* cells ONE chip operation pushes -- the terms pz_circuit_cells sums; op: 0 assign_integer(limbs), 1 square(limbs),
 * 2 refresh (limbs x limbs product), 3 load_zero / load_constant, 4 mul_mod(limbs), 5 assert_equal_fresh(limbs). */
int pz_op_cells(int op, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits, size_t* advice_cells, size_t* lookup_cells);

/* host-pointer form: steps / modulus / outputs in process memory (n_steps*advice_cells and n_steps*lookup_cells Fr
 * elements; either output may be NULL); synchronises before returning. */
int pz_witness_expand(pz_ctx* ctx, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits, const uint64_t* steps,
                      size_t n_steps, const uint64_t* modulus, uint64_t* advice_out, uint64_t* lookup_out);

/* ---------------------------------------------------------------------------------------------
 * "next" rows, built on the kernels above
 * ------------------------------------------------------------------------------------------- */
/* ParamsKZG::setup with a known toxic scalar s (what halo2-lib's gen_srs does with its seeded rng; reached from
 * /root/reference/src/bench.rs:161-171): d_g[i] = [s^i] G, d_g_lagrange[i] = [L_i(s)] G over the 2^k domain
 * generated by omega (L_i(s) = (s^n - 1) omega^i / (n (s - omega^i))).  Either output may be NULL.  s, omega: Fr
 * Montgomery (host); outputs: device, 2^k affine points each.  s must not lie in the domain (PZ_ERR_INVALID). */
int pz_srs_setup_g1_dev(pz_ctx* ctx, uint32_t k, const uint64_t s[4], const uint64_t omega[4], uint64_t* d_g,
                        uint64_t* d_g_lagrange);
/* The Lagrange-basis SRS from the monomial one WITHOUT the toxic scalar (what ParamsKZG does with a params file's `g`):
 * d_g_lagrange[i] = (1/n) sum_j omega^(-i j) d_g[j], a radix-2 inverse FFT over G1 (the twiddle product is a scalar
 * multiplication).  d_g, d_g_lagrange: 2^k affine points on the device (may not alias); omega_inv, n_inv: Fr Montgomery
 * (host).  One-time cost per SRS, (k + 1) 2^(k-1) scalar multiplications. */
int pz_srs_lagrange_from_monomial_dev(pz_ctx* ctx, uint32_t k, const uint64_t omega_inv[4], const uint64_t n_inv[4],
                                      const uint64_t* d_g, uint64_t* d_g_lagrange);
/* keygen, permutation polynomials (halo2 permutation::keygen::Assembly::build_pk): d_sigma[j][i] = delta^(d_map_col[j*n+i]) *
 * omega^(d_map_row[j*n+i]) over the 2^k domain, n = 2^k, for m columns -- (map_col, map_row) is the cell the copy-constraint
 * cycle maps (j, i) to (circuit structure, supplied by the caller: device arrays of m * n u32 each).  ONE call covers the whole
 * permutation: every image must lie inside the m x n cells of the call (an image outside is clamped on the device and the next
 * synchronising entry point returns PZ_ERR_ASYNC). */
int pz_permutation_sigma_dev(pz_ctx* ctx, const uint32_t* d_map_col, const uint32_t* d_map_row, size_t m, uint32_t k,
                             const uint64_t omega[4], const uint64_t delta[4], uint64_t* d_sigma, size_t sigma_stride);
/* keygen_vk + keygen_pk for n_cols Lagrange-form fixed columns (selectors, constants, table, sigma) on the device: their
 * commitments (commit_lagrange, n_cols x 12), then IN PLACE their coefficient form, and (d_ext != NULL) their values on the
 * extended coset (as pz_ntt_fr_extend_dev) -- the proving key's polynomials, resident in HBM for every later proof.
 * Reached in the reference through bench.rs:161-175 (keygen_vk / keygen_pk inside bench_builder). */
int pz_keygen_columns_dev(pz_ctx* ctx, const pz_bases* bases_lagrange, uint64_t* d_cols, size_t n_cols, size_t col_stride,
                          uint32_t k, uint32_t log_e, const uint64_t omega_n[4], const uint64_t omega_n_inv[4],
                          const uint64_t n_inv[4], const uint64_t* coset_gens, uint64_t* d_commit_jac, uint64_t* d_ext,
                          size_t ext_stride);
/* on-curve validation of n affine points on the device (y^2 = x^3 + 3, canonical coordinates, identity (0,0)
 * accepted): *n_bad = number of points that fail.  The check halo2curves' read_raw performs when ParamsKZG::read
 * loads a `params/kzg_bn254_{k}.srs` file (paillier_halo2_amd/srs.py reads that format). */
int pz_g1_check_dev(pz_ctx* ctx, const uint64_t* d_points, size_t n, uint64_t* n_bad);
/* evaluation of n_cols coefficient-form polynomials (n coefficients each, device) at the point x:
 * d_out[col] = sum_i d_coeffs[col][i] * x^i   (the evals phase of create_proof / eval_polynomial). */
int pz_poly_eval_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                     const uint64_t x[4], uint64_t* d_out);
/* the same at 1..4 points in one pass over the coefficients (a polynomial's rotation set x, omega x, ...):
 * xs = n_points x 4 words, d_out[col][point].  Identical values to n_points calls of pz_poly_eval_dev. */
int pz_poly_eval_multi_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                           const uint64_t* xs, uint32_t n_points, uint64_t* d_out);

/* The prover steps between the commitments and the NTTs (in the reference all of
 * them run inside create_proof, reached from /root/reference/src/bench.rs:161-171).  Device pointers throughout, field
 * elements Fr Montgomery, strides in uint64_t units; challenges / constants are host pointers to one element.
 * Precondition checked on entry: the challenges beta, gamma, delta, y handed to pz_permutation_product_sets_dev and the three
 * pz_quotient_*_dev functions must be canonical (below r) -- they are converted on the host into the limb form the kernels
 * keep in scalar registers; a non-canonical word returns PZ_ERR_INVALID (it is not reduced). */
/* halo2 BatchInvert: d_a[i] <- 1 / d_a[i] in place, zeros stay zero. */
int pz_fr_batch_invert_dev(pz_ctx* ctx, uint64_t* d_a, size_t n);
/* running product: d_z[0] = z0, d_z[i+1] = d_z[i] * d_a[i] for i < n-1 (n outputs; d_z may alias d_a). */
int pz_fr_prefix_product_dev(pz_ctx* ctx, const uint64_t* d_a, size_t n, const uint64_t z0[4], uint64_t* d_z);
/* permutation::Argument::commit for one chunk of m columns over the 2^log_n domain generated by omega:
 *   d_z[0] = z0,  d_z[i+1] = d_z[i] * prod_{j<m} (v_j[i] + beta*delta_start*delta^j*omega^i + gamma)
 *                                              / (v_j[i] + beta*sigma_j[i] + gamma)
 * v_j = d_cols + j*col_stride (Lagrange values), sigma_j = d_sigma + j*sigma_stride (permutation polynomial values).
 * delta_start = delta^(index of the chunk's first column).  Blinding rows are the caller's business. */
int pz_permutation_product_dev(pz_ctx* ctx, const uint64_t* d_cols, size_t col_stride, const uint64_t* d_sigma,
                               size_t sigma_stride, size_t m, uint32_t log_n, const uint64_t omega[4],
                               const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_start[4],
                               const uint64_t delta[4], const uint64_t z0[4], uint64_t* d_z);
/* all sets of the permutation argument in one call: m columns in chunks of chunk_len (set j = columns
 * [j*chunk_len, ...), delta powers continuing across sets); d_z[0][0] = 1 and d_z[j][0] = d_z[j-1][usable_rows], as
 * halo2 chains them.  The sets are computed independently (one batched inversion, one batched scan), then chained. */
int pz_permutation_product_sets_dev(pz_ctx* ctx, const uint64_t* d_cols, size_t col_stride, const uint64_t* d_sigma,
                                    size_t sigma_stride, size_t m, uint32_t chunk_len, uint32_t log_n, size_t usable_rows,
                                    const uint64_t omega[4], const uint64_t beta[4], const uint64_t gamma[4],
                                    const uint64_t delta[4], uint64_t* d_z, size_t z_stride);
/* lookup::prover permute_expression_pair for halo2-lib's range-check lookups: every input column (values in
 * [0, 2^value_bits), value_bits <= 24) against the shared table column, over the first `rows` (usable) rows:
 *   d_perm_inputs[col] = the input column sorted ascending;
 *   d_perm_tables[col][i] = d_perm_inputs[col][i] where a run of equal inputs starts, otherwise the left-over table
 *   values in ascending order (halo2's BTreeMap order).
 * PZ_ERR_RANGE if a value is not a canonical integer below 2^value_bits or an input value does not occur in the
 * table (the reference's prover fails there: the lookup is unsatisfiable).  Rows >= `rows` are not written. */
int pz_lookup_permute_dev(pz_ctx* ctx, const uint64_t* d_inputs, size_t n_cols, size_t col_stride,
                          const uint64_t* d_table, size_t rows, uint32_t value_bits, uint64_t* d_perm_inputs,
                          uint64_t* d_perm_tables, size_t out_stride);
/* lookup grand products, n_lookups of them against one table column (z0 the same for all, normally 1):
 *   d_z[k][0] = z0, d_z[k][i+1] = d_z[k][i] * (A_k[i]+beta)(S[i]+gamma) / ((A'_k[i]+beta)(S'_k[i]+gamma)).
 * One batched inversion and one batched scan for all of them. */
int pz_lookup_product_dev(pz_ctx* ctx, const uint64_t* d_inputs, size_t input_stride, const uint64_t* d_table,
                          const uint64_t* d_perm_inputs, size_t perm_input_stride, const uint64_t* d_perm_tables,
                          size_t perm_table_stride, size_t n_lookups, size_t n, const uint64_t beta[4],
                          const uint64_t gamma[4], const uint64_t z0[4], uint64_t* d_z, size_t z_stride);
/* evaluate_h, custom-gate part, for halo2-lib's vertical gate on the extended domain of 2^log_ext points:
 *   for each column j in order:  d_h[i] = d_h[i]*y + sel_j[i] * (a_j[i] + a_j[i+s]*a_j[i+2s] - a_j[i+3s]),
 * indices mod 2^log_ext, s = rot_step = 2^(log_ext - k) (one row of the 2^k domain).  d_h is read and written. */
int pz_quotient_gate_dev(pz_ctx* ctx, const uint64_t* d_adv_ext, size_t adv_stride, const uint64_t* d_sel_ext,
                         size_t sel_stride, size_t n_cols, uint32_t log_ext, uint32_t rot_step, const uint64_t y[4],
                         uint64_t* d_h);
/* evaluate_h, permutation argument (halo2 plonk/evaluation.rs "Permutations"): m_total columns in n_sets chunks of
 * chunk_len, all arrays on the extended domain of 2^log_ext points X_i = coset_g * omega_ext^i, one domain row =
 * rot_step indices, last_rotation = blinding_factors + 1 rows.  In halo2's order, each folded as h = h*y + term:
 *   l0 (1 - z_0);  l_last (z_last^2 - z_last);  for j > 0: l0 (z_j - z_{j-1}(omega^-last_rotation X));
 *   for every set: l_active ( z_j(omega X) prod_c (v_c + beta sigma_c + gamma) - z_j(X) prod_c (v_c + delta^c beta X + gamma) ).
 * omega_ext must be a PRIMITIVE 2^log_ext-th root of unity (log_ext >= 1): rows i and i + 2^(log_ext-1) are evaluated together and
 * share the identity term through X_{i + N/2} = -X_i.  All challenges canonical (below r), Montgomery form. */
int pz_quotient_permutation_dev(pz_ctx* ctx, const uint64_t* d_cols_ext, size_t col_stride, const uint64_t* d_sigma_ext,
                                size_t sigma_stride, const uint64_t* d_z_ext, size_t z_stride, uint32_t n_sets,
                                uint32_t chunk_len, uint32_t m_total, uint32_t log_ext, uint32_t rot_step,
                                uint32_t last_rotation, const uint64_t* d_l0, const uint64_t* d_l_last,
                                const uint64_t* d_l_active, const uint64_t beta[4], const uint64_t gamma[4],
                                const uint64_t delta[4], const uint64_t coset_g[4], const uint64_t omega_ext[4],
                                const uint64_t y[4], uint64_t* d_h);
/* the same for a RANGE of sets, so that the prover can stream tiles of extended columns through it (evaluate_h never needs all
 * columns on the extended domain at once): this call adds the product lines of sets [set_lo, set_lo + n_sets) of n_sets_total;
 * d_cols_ext / d_sigma_ext point at the call's first column (= column set_lo * chunk_len of the argument) and m_cols counts the call's
 * columns (n_sets * chunk_len, fewer only when the call ends with the argument's last set); d_z_ext holds ALL n_sets_total products.
 * head != 0: the boundary and chaining lines (l0 (1 - z_0), l_last (z_last^2 - z_last), l0 (z_j - z_{j-1}(omega^-last X))) are added
 * first -- pass it with the first range only.  Calls in increasing set order give the value of one pz_quotient_permutation_dev. */
int pz_quotient_permutation_part_dev(pz_ctx* ctx, const uint64_t* d_cols_ext, size_t col_stride, const uint64_t* d_sigma_ext,
                                     size_t sigma_stride, const uint64_t* d_z_ext, size_t z_stride, uint32_t n_sets_total,
                                     uint32_t set_lo, uint32_t n_sets, uint32_t chunk_len, uint32_t m_cols, int head, uint32_t log_ext,
                                     uint32_t rot_step, uint32_t last_rotation, const uint64_t* d_l0, const uint64_t* d_l_last,
                                     const uint64_t* d_l_active, const uint64_t beta[4], const uint64_t gamma[4],
                                     const uint64_t delta[4], const uint64_t coset_g[4], const uint64_t omega_ext[4],
                                     const uint64_t y[4], uint64_t* d_h);
/* evaluate_h, lookup arguments (n_lookups of them against one table column), per lookup in halo2's order:
 *   l0 (1 - z);  l_last (z^2 - z);  l_active ( z(omega X)(a' + beta)(s' + gamma) - z(X)(a + beta)(s + gamma) );
 *   l0 (a' - s');  l_active (a' - s')(a' - a'(omega^-1 X)). */
int pz_quotient_lookup_dev(pz_ctx* ctx, const uint64_t* d_input_ext, size_t input_stride, const uint64_t* d_table_ext,
                           const uint64_t* d_perm_input_ext, size_t perm_input_stride, const uint64_t* d_perm_table_ext,
                           size_t perm_table_stride, const uint64_t* d_z_ext, size_t z_stride, uint32_t n_lookups,
                           uint32_t log_ext, uint32_t rot_step, const uint64_t* d_l0, const uint64_t* d_l_last,
                           const uint64_t* d_l_active, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4],
                           uint64_t* d_h);
/* division by the vanishing polynomial on the extended coset: d_h[i] /= (coset_g * omega_ext^i)^(2^log_n) - 1,
 * i < 2^(log_n + log_e) (the divisor takes 2^log_e distinct values; log_e = 0: the points are ONE coset of the 2^log_n-th roots and
 * the divisor is the constant coset_g^n - 1 -- the prover evaluates a quotient of degree < 3n on three such cosets instead of halo2's
 * 4n-point coset, paillier_halo2_amd/prover.py). */
int pz_quotient_finish_dev(pz_ctx* ctx, uint64_t* d_h, uint32_t log_n, uint32_t log_e, const uint64_t coset_g[4],
                           const uint64_t omega_ext[4]);
/* d_a[col][i] *= c * g^i, i < n (distribute_powers; with g = 1/coset_g the un-scaling step of extended_to_coeff).
 * c may be NULL (= 1). */
int pz_fr_distribute_powers_dev(pz_ctx* ctx, uint64_t* d_a, size_t n_cols, size_t col_stride, size_t n,
                                const uint64_t g[4], const uint64_t c[4]);
/* random linear combination of n_cols polynomials (n elements each): d_out[i] = sum_j v^(n_cols-1-j) * p_j[i], computed
 * as acc = acc*v + p_j over the columns; accumulate != 0 starts from the current d_out (continuing a Horner fold
 * across calls).  The polynomial folding step of the multiopen argument, before pz_poly_div_linear_dev. */
int pz_fr_lincomb_dev(pz_ctx* ctx, const uint64_t* d_polys, size_t n_cols, size_t col_stride, size_t n,
                      const uint64_t v[4], uint64_t* d_out, int accumulate);
/* kate_division: d_q[col] = (p_col(X) - p_col(x)) / (X - x) for n-coefficient polynomials: n-1 coefficients, the
 * n-th slot is written as zero (d_q may alias d_coeffs). */
int pz_poly_div_linear_dev(pz_ctx* ctx, const uint64_t* d_coeffs, size_t n_cols, size_t col_stride, size_t n,
                           const uint64_t x[4], uint64_t* d_q, size_t q_stride);

/* SHPLONK multi-point opening (halo2 multiopen::shplonk::prover, the last step of create_proof; SURVEY.md section 8f
 * rank 3).  Queries are grouped by rotation set: set k opens set_n_polys[k] coefficient-form polynomials (device
 * pointers, n coefficients each, listed set after set in d_polys) at set_n_points[k] points given as indices into
 * `points` (host, n_points_total x 4 limbs; the union of all sets' points, no duplicates).  evals (host): for each set,
 * for each of its polynomials, its value at each of the set's points.  y, v, u: the transcript's challenges (host).
 *   begin : d_h = sum_k v^k (sum_j y^j P_kj - R_k) / Z_{S_k}   (n coefficients; commit it, hash it, draw u)
 *   finish: d_h2 = (sum_k v^k z_k (sum_j y^j P_kj - R_k(u)) - Z_T(u) h) / (X - u) / z_0,  z_k = Z_{T \ S_k}(u);
 *           frees the state.  d_h2 may not alias d_h.
 * At most 16 sets, 8 points per set, 32 points overall; one opening in flight per context (the state lives in the context's
 * workspace). */
typedef struct pz_shplonk pz_shplonk;
int pz_shplonk_begin_dev(pz_ctx* ctx, size_t n, uint32_t n_sets, const uint32_t* set_n_polys, const uint64_t* const* d_polys,
                         const uint32_t* set_n_points, const uint32_t* point_idx, uint32_t n_points_total, const uint64_t* points,
                         const uint64_t* evals, const uint64_t y[4], const uint64_t v[4], uint64_t* d_h, pz_shplonk** state);
int pz_shplonk_finish_dev(pz_ctx* ctx, pz_shplonk* state, const uint64_t u[4], const uint64_t* d_h, uint64_t* d_h2);
int pz_shplonk_free(pz_ctx* ctx, pz_shplonk* state);

/* ---------------------------------------------------------------------------------------------
 * Patch point D as entry points: keygen and create_proof, one call per transcript round.
 * Replaces, for halo2-lib circuits (one gate q (a + b c - d) per advice column, range-check lookups of lookup-enabled advice columns
 * against one table column, a permutation over [advice | lookup advice | constants]): halo2-axiom's keygen_vk / keygen_pk /
 * create_proof as reached from /root/reference/src/bench.rs:161-171 (bench_builder -> keygen -> gen_proof).  The composition is the
 * one of paillier_halo2_amd/host/create_proof.hpp (the code tests/cpp/prove_connected runs; the quotient on three cosets of <omega_n>:
 * DESIGN.md section 6.3); the transcript stays with the caller: every phase returns what must be absorbed (affine points of 8 words,
 * evaluations of 4 words, all Montgomery) and takes the challenges squeezed after it (Montgomery representatives below r, else
 * PZ_ERR_INVALID).  The circuit STRUCTURE is an input of keygen, as it is of halo2's: selector positions, the constants column, and
 * the copy constraints as a permutation map (cell (c, r) of the permutation's columns -> (map_col, map_row); identity where
 * unconstrained) over m = n_adv + n_lk + 1 columns; the lookup table is 0 .. 2^lookup_bits - 1.  chunk = 2 (cs.degree() = 4).
 *
 * pz_pk_create   selectors: u8 [n_adv][2^k]; constants: n_constants x 4 words, canonical integers (NOT Montgomery), row order;
 *                map_col / map_row: u32 [m][2^k] (all host).  Builds and keeps RESIDENT the commitments, the coefficient forms and the
 *                extended forms of the fixed and sigma columns, l_0 / l_last / l_active, and one proof's workspace (at config c2:
 *                116 + 30 GB).  n_adv, n_lk >= 1; lookup_bits < k.  tile: columns extended per step of the quotient (even; 64).  The key holds device memory of `ctx`
 *                (which must outlive it) and serves ONE proof at a time.  The selectors are uploaded as bytes and become field elements on
 *                the device (pz_fr_from_mask_dev); the host arrays are not referenced after the call returns.
 *                ext_resident_cols: PZ_PK_EXT_ALL = the extended-coset forms of every selector and sigma column stay resident (halo2's
 *                ProvingKey: fixed_cosets / permutation.cosets); a number R = the STREAMED proving key: only selector j / sigma j with j < R
 *                keep them, the others are re-extended from their coefficient forms per tile inside pz_proof_quotient (+ one transform per
 *                such column, three cosets, per proof) -- the memory plan of shapes whose extended key does not fit (BASELINE config c5:
 *                3072-bit n, k = 19: 239 GB; R = 0 there).  The proof does not depend on R, byte for byte.
 * pz_pk_create_dev  the same with selectors / map_col / map_row ALREADY ON THE DEVICE (device pointers of `ctx`'s device: a structure
 *                generated there -- pz_circuit_structure_dev -- or uploaded by the caller): nothing of them crosses PCIe or is copied on the
 *                host; they are only read during the call.  constants stay a (small) host array.
 * pz_pk_info     n_fixed = n_adv + 2 (selectors | constants | table); blinding_words = 64-bit words of caller randomness one proof
 *                consumes; evals_words = length of pz_proof_evaluate's output.
 * pz_pk_commitments  the verifying key's commitments: n_fixed x 8 and m x 8 words.
 *
 * pz_proof_begin d_cols: device, [m][2^k] elements: the advice then the lookup-advice columns as pz_circuit_expand_cols_dev wrote
 *                them (rows >= max_rows zero); the last column and the blinding rows are filled here; CONSUMED (ends in coefficient
 *                form).  blinding: n_blinding >= blinding_words random words (host) -- a production prover hands over OS randomness; the
 *                elements are drawn uniformly below r by rejection (254-bit candidates), blinding_words includes the margin.  The
 *                deterministic stream from `seed` (tests, benches: NOT zero-knowledge) must be asked for explicitly: blinding = NULL
 *                AND n_blinding = PZ_BLINDING_SEEDED_TEST_STREAM; NULL with any other count is PZ_ERR_INVALID.
 *                -> advice_affine: (n_adv + n_lk) x 8.                                            [transcript: ... -> theta]
 * pz_proof_lookups       -> n_lk x 8 each (permuted inputs A', permuted tables S')                 [-> beta, gamma]
 * pz_proof_products      -> n_sets x 8 (permutation products), n_lk x 8 (lookup products), 8 (the vanishing argument's random
 *                        polynomial)                                                              [-> y]
 * pz_proof_quotient      -> 3 x 8: the quotient's pieces h_0, h_1, h_2                             [-> x]
 * pz_proof_evaluate      -> evals_words words: for each family in this order, for each of its polynomials, its values at the
 *                        family's points (indices into {x, wx, w^2 x, w^3 x, w^-(blinding_factors+1) x, w^-1 x}):
 *                        advice n_adv {0,1,2,3} | lookup advice then the constants column n_lk + 1 {0} | fixed n_fixed {0} |
 *                        sigma m {0} | permutation products n_sets {0,1,4} | lookup products n_lk {0,1} | A' n_lk {0,5} |
 *                        S' n_lk {0} | random 1 {0} | h_0 + x^n h_1 + x^2n h_2 1 {0} (the verifier computes the last itself:
 *                        not absorbed)                                                            [-> SHPLONK's y, v]
 * pz_proof_open_begin    -> 8: SHPLONK's first commitment                                          [-> u]
 * pz_proof_open_finish   -> 8: the second; quotient_degree_ok = 0 if the quotient's degree exceeds 3n - 4 (an unsatisfied
 *                        witness: the proof will not verify).
 * Phases out of order return PZ_ERR_INVALID; after any error only pz_proof_free is valid.  pz_proof_free releases the key's
 * workspace for the next proof; pz_pk_free refuses (PZ_ERR_INVALID) while a proof is open.
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * The circuit STRUCTURE of the reference's drivers, generated on the device (csrc/pz_structure.hip) -- what halo2's keygen extracts by
 * synthesising paillier_enc_test / paillier_enc_add_test (/root/reference/src/bench.rs:33-117) once: selector positions, the
 * copy-constraint permutation, the constants column, the break-point column layout.  It depends on the shape only: key size, limb
 * width, lookup bits and the BITS of the two fixed exponents (paillier.rs:50-55 hands m and n to pow_mod_fixed_exp), so with the
 * reference's circuit every new message needs a new structure and a new key:  pz_circuit_structure_dev -> pz_pk_create_dev ->
 * pz_structure_free -> proofs.  The compiled counterpart of paillier_halo2_amd/circuit_structure.py, equal to it array for array.
 * kind: 0 encrypt (exp_g = the message m, exp_r = the modulus n: ceil(limbs_n * limb_bits / 64) words each; only their bits are used), 1 add (no exponents),
 * 2 the uniform-shape encrypt circuit (exp_g ignored: one structure for every message of a key).  limb_bits 16..90, lookup_bits < k.
 * minimum_rows: the argument of the tester's calculate_params -- it fixes the NUMBER of advice / lookup-advice columns as
 * ceil(cells / (2^k - minimum_rows)) (20 on the reference's bench path, 9 under MockProver); columns are FILLED to
 * max_rows = 2^k - (blinding_factors + 3) (halo2-lib's FlexGateConfig::max_rows [D]).  PZ_ERR_UNSUPPORTED beyond 2^31 cells.
 * pz_structure_info: n_adv configured advice columns of which n_adv_filled hold cells; n_lk; max_rows; the counts K4 needs
 * (n_steps_g, n_steps_r) and the stream's sizes.
 * pz_structure_arrays: DEVICE pointers d_selectors u8 [n_adv][2^k], d_map_col / d_map_row u32 [m][2^k] (m = n_adv + n_lk + 1),
 * d_col_starts u64 [n_adv + 1] (what pz_circuit_expand_cols_dev takes); HOST pointers constants (n_constants x 4 words, canonical)
 * and col_starts_host.  All owned by the structure, valid until pz_structure_free.  Any output may be NULL.
 * ------------------------------------------------------------------------------------------- */
typedef struct pz_structure pz_structure;
int pz_circuit_structure_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits, uint32_t k,
                             const uint64_t* exp_g, const uint64_t* exp_r, size_t minimum_rows, uint32_t blinding_factors,
                             pz_structure** out);
int pz_structure_info(const pz_structure* st, size_t* n_adv, size_t* n_adv_filled, size_t* n_lk, size_t* max_rows, size_t* n_constants,
                      size_t* n_cells, size_t* n_lookups, size_t* n_steps_g, size_t* n_steps_r);
int pz_structure_arrays(const pz_structure* st, const uint8_t** d_selectors, const uint32_t** d_map_col, const uint32_t** d_map_row,
                        const uint64_t** d_col_starts, const uint64_t** constants, const uint64_t** col_starts_host);
int pz_structure_free(pz_structure* st);

typedef struct pz_pk pz_pk;
typedef struct pz_proof pz_proof;
#define PZ_BLINDING_SEEDED_TEST_STREAM (~(size_t)0)
#define PZ_PK_EXT_ALL (~(size_t)0)
int pz_pk_create(pz_ctx* ctx, const pz_bases* bases_lagrange, const pz_bases* bases_monomial, uint32_t k, uint32_t lookup_bits,
                 uint32_t blinding_factors, size_t max_rows, size_t n_adv, size_t n_lk, const uint8_t* selectors,
                 const uint64_t* constants, size_t n_constants, const uint32_t* map_col, const uint32_t* map_row, size_t tile,
                 size_t ext_resident_cols, pz_pk** out);
int pz_pk_create_dev(pz_ctx* ctx, const pz_bases* bases_lagrange, const pz_bases* bases_monomial, uint32_t k, uint32_t lookup_bits,
                     uint32_t blinding_factors, size_t max_rows, size_t n_adv, size_t n_lk, const uint8_t* d_selectors,
                     const uint64_t* constants, size_t n_constants, const uint32_t* d_map_col, const uint32_t* d_map_row, size_t tile,
                     size_t ext_resident_cols, pz_pk** out);
int pz_pk_info(const pz_pk* pk, size_t* n_fixed, size_t* n_perm_cols, size_t* n_sets, size_t* blinding_words, size_t* evals_words);
int pz_pk_commitments(const pz_pk* pk, uint64_t* fixed_affine, uint64_t* sigma_affine);
int pz_pk_free(pz_pk* pk);
int pz_proof_begin(pz_pk* pk, uint64_t* d_cols, uint64_t seed, const uint64_t* blinding, size_t n_blinding, pz_proof** out,
                   uint64_t* advice_affine);
int pz_proof_lookups(pz_proof* proof, const uint64_t theta[4], uint64_t* perm_inputs_affine, uint64_t* perm_tables_affine);
int pz_proof_products(pz_proof* proof, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* perm_z_affine,
                      uint64_t* lookup_z_affine, uint64_t* random_affine);
int pz_proof_quotient(pz_proof* proof, const uint64_t y[4], uint64_t* h_affine);
int pz_proof_evaluate(pz_proof* proof, const uint64_t x[4], uint64_t* evals);
int pz_proof_open_begin(pz_proof* proof, const uint64_t y[4], const uint64_t v[4], uint64_t* w1_affine);
int pz_proof_open_finish(pz_proof* proof, const uint64_t u[4], uint64_t* w2_affine, int* quotient_degree_ok);
int pz_proof_free(pz_proof* proof);

/* ---------------------------------------------------------------------------------------------
 * measurement helpers (used by bench.py; not part of the reference surface).  Issue-rate microbenchmarks: libpz_probe.so.
 * ------------------------------------------------------------------------------------------- */
/* HIP-event timing of the dominant kernel on the context's stream: accumulated since the last
 * reset, for kernel class `which` (0 = MSM bucket accumulation, 1 = NTT passes, 2 = modexp trace,
 * 3 = witness expand, 4 = MSM whole pipeline, 5 = MSM digit sort (histogram, scans, scatter), 6 = MSM bucket folds and
 * reduction tree).  Enabled by pz_timing_enable(ctx, 1). */
int pz_timing_enable(pz_ctx* ctx, int on);
int pz_timing_reset(pz_ctx* ctx);
int pz_timing_get(pz_ctx* ctx, int which, double* total_ms, uint64_t* launches);
#ifdef __cplusplus
}
#endif
#endif /* PZ_H */
