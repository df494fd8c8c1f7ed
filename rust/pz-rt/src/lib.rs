//! pz-rt -- what the four patch points of INTEGRATION.md call: a process-wide context per GPU, status checking with the reference's
//! error behaviour (no CPU fallback: a non-zero status is a panic or `plonk::Error::Synthesis`, never a silent second path), a cache
//! of window-shifted base tables keyed by the SRS slice, and the witness trace of `PaillierChip::encrypt`.
//!
//! SOURCE ONLY here (no cargo / rustc in the build image); the C++ mirrors of these helpers ARE built and tested:
//! `paillier_halo2_amd/host/prove_c2.cpp` (device-resident flow), `host/msm_sharded.cpp` (N contexts), `host/paillier_chip.hpp`.
//! Reference call sites: /root/reference/src/paillier.rs:51,55,57,81 (patch point C), src/bench.rs:161-171 (A, B, D).
use core::ffi::c_int;
use once_cell::sync::OnceCell;
use pz_sys::*;
use std::collections::HashMap;
use std::sync::Mutex;

/// `PZ_ABI_VERSION` this crate was written against (include/pz.h)
pub const ABI: c_int = 7;

pub struct Ctx(pub *mut pz_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {} // a pz_ctx serialises its entry points internally (pz.h: recursive mutex per context)

static CTXS: OnceCell<Mutex<HashMap<c_int, &'static Ctx>>> = OnceCell::new();

/// the context of device `dev`, created on first use (`pz_init(1, &[dev], ..)`); one process per GPU uses `ctx()` = `ctx_on(0)`
pub fn ctx_on(dev: c_int) -> *mut pz_ctx {
    let map = CTXS.get_or_init(|| Mutex::new(HashMap::new()));
    let mut m = map.lock().unwrap();
    if let Some(c) = m.get(&dev) {
        return c.0;
    }
    assert_eq!(unsafe { pz_abi_version() }, ABI, "libpz_hip.so was built for another PZ_ABI_VERSION");
    let mut p: *mut pz_ctx = core::ptr::null_mut();
    check_raw(unsafe { pz_init(1, &dev, &mut p) }, core::ptr::null());
    let c: &'static Ctx = Box::leak(Box::new(Ctx(p)));
    m.insert(dev, c);
    c.0
}
pub fn ctx() -> *mut pz_ctx {
    ctx_on(0)
}
/// Reserve one arena over the device memory that is free now (less `leave_bytes` for the runtime): from here on this context's keys,
/// workspaces, structures and tables are carved out of it and a per-message `DeviceKey::for_message` / drop makes no driver allocation call
/// (pz.h `pz_dev_arena`; at config c5 half of a keygen + proof step was `hipMalloc` / `hipFree` without it).  Call once, before the SRS is
/// loaded.  -> the arena's size
pub fn reserve_device_memory(leave_bytes: usize) -> usize {
    let (mut free, mut total) = (0usize, 0usize);
    check(unsafe { pz_dev_mem_info(ctx(), &mut free, &mut total) });
    let bytes = free.saturating_sub(leave_bytes);
    check(unsafe { pz_dev_arena(ctx(), bytes) });
    bytes
}

/// a further context on device 0 (its own stream: the witness / commitment / transform contexts of INTEGRATION.md section 5a)
pub fn new_ctx() -> *mut pz_ctx {
    let mut p: *mut pz_ctx = core::ptr::null_mut();
    check_raw(unsafe { pz_init(1, &0, &mut p) }, core::ptr::null());
    p
}

fn check_raw(status: c_int, ctx: *const pz_ctx) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(pz_strerror(status)) }.to_string_lossy().into_owned();
        let detail = if ctx.is_null() { String::new() } else { unsafe { std::ffi::CStr::from_ptr(pz_last_hip_error(ctx)) }.to_string_lossy().into_owned() };
        // the reference panics where num-bigint would (BigUint % 0, paillier.rs:91) and unwraps every plonk::Error (bench.rs:46-74)
        panic!("pz status {status}: {msg} {detail}");
    }
}
/// panic on a non-zero status (the reference `.unwrap()`s every `Result` of the chip: bench.rs:46,49,57,60,62,66,74)
pub fn check(status: c_int) {
    check_raw(status, ctx());
}

/// `pz_bases_load_g1` once per SRS slice (keyed by address and length: `ParamsKZG` owns its vectors for the process lifetime)
pub fn bases_for(points: *const u64, n: usize) -> *const pz_bases {
    static TABLES: OnceCell<Mutex<HashMap<(usize, usize), usize>>> = OnceCell::new();
    let mut t = TABLES.get_or_init(|| Mutex::new(HashMap::new())).lock().unwrap();
    if let Some(h) = t.get(&(points as usize, n)) {
        return *h as *const pz_bases;
    }
    let mut h: *mut pz_bases = core::ptr::null_mut();
    check(unsafe { pz_bases_load_g1(ctx(), points, n, 0, 0, &mut h) });
    t.insert((points as usize, n), h as usize);
    h
}

/// the whole witness trace of `PaillierChip::encrypt` (paillier.rs:32-60): g^m chain | r^n chain | final mul_mod, each step a|b|q|r
pub struct EncryptTrace {
    pub steps: Vec<u64>,
    pub n_steps_g: u32,
    pub n_steps_r: u32,
    pub c: Vec<u64>,
}
pub fn encrypt_trace(n: &[u64], g: &[u64], m: &[u64], r: &[u64]) -> EncryptTrace {
    let ln = n.len();
    let bits = |e: &[u64]| -> usize {
        let top = e.iter().rposition(|w| *w != 0);
        top.map_or(0, |t| t * 64 + 64 - e[t].leading_zeros() as usize) + e.iter().map(|w| w.count_ones() as usize).sum::<usize>()
    };
    let cap = bits(m) + bits(n) + 1;
    let mut out = EncryptTrace { steps: vec![0u64; cap * 4 * 2 * ln], n_steps_g: 0, n_steps_r: 0, c: vec![0u64; 2 * ln] };
    check(unsafe {
        pz_paillier_encrypt(ctx(), ln as u32, 1, n.as_ptr(), g.as_ptr(), m.as_ptr(), r.as_ptr(), out.steps.as_mut_ptr(), cap,
                            &mut out.n_steps_g, &mut out.n_steps_r, out.c.as_mut_ptr())
    });
    out
}

/// Patch point D through the library's stepper (include/pz.h "patch point D as entry points"): the proving key
/// resident on the device, built from the structure halo2's `Assembly` holds.  `create_proof` then is `ProofSession`'s seven methods with the
/// caller's transcript in between (reference: /root/reference/src/bench.rs:161-171, `gen_proof` -> halo2-axiom `create_proof`).
pub struct DeviceKey {
    pub pk: *mut pz_pk,
    pub n_fixed: usize,
    pub n_perm_cols: usize,
    pub n_sets: usize,
    pub blinding_words: usize,
    pub evals_words: usize,
}
impl DeviceKey {
    /// selectors: `n_adv * 2^k` bytes; constants: canonical 4-limb integers in row order; map_col / map_row: `(n_adv + n_lk + 1) * 2^k` each
    #[allow(clippy::too_many_arguments)]
    pub fn new(lagrange: *const pz_bases, monomial: *const pz_bases, k: u32, lookup_bits: u32, blinding_factors: u32, max_rows: usize,
               n_adv: usize, n_lk: usize, selectors: &[u8], constants: &[u64], map_col: &[u32], map_row: &[u32]) -> Self {
        let n = 1usize << k;
        assert_eq!(selectors.len(), n_adv * n);
        assert_eq!(map_col.len(), (n_adv + n_lk + 1) * n);
        assert_eq!(map_row.len(), map_col.len());
        assert_eq!(constants.len() % 4, 0);
        let mut pk: *mut pz_pk = core::ptr::null_mut();
        check(unsafe {
            pz_pk_create(ctx(), lagrange, monomial, k, lookup_bits, blinding_factors, max_rows, n_adv, n_lk, selectors.as_ptr(), constants.as_ptr(),
                         constants.len() / 4, map_col.as_ptr(), map_row.as_ptr(), 64, usize::MAX /* PZ_PK_EXT_ALL */, &mut pk)
        });
        let (mut a, mut b, mut c, mut d, mut e) = (0usize, 0usize, 0usize, 0usize, 0usize);
        check(unsafe { pz_pk_info(pk, &mut a, &mut b, &mut c, &mut d, &mut e) });
        DeviceKey { pk, n_fixed: a, n_perm_cols: b, n_sets: c, blinding_words: d, evals_words: e }
    }
    /// the verifying key's commitments (affine, Montgomery; 8 words per point): fixed columns, sigma columns
    pub fn vk_commitments(&self) -> (Vec<u64>, Vec<u64>) {
        let (mut f, mut s) = (vec![0u64; 8 * self.n_fixed], vec![0u64; 8 * self.n_perm_cols]);
        check(unsafe { pz_pk_commitments(self.pk, f.as_mut_ptr(), s.as_mut_ptr()) });
        (f, s)
    }
}
impl Drop for DeviceKey {
    fn drop(&mut self) {
        unsafe { pz_pk_free(self.pk) };
    }
}

impl DeviceKey {
    /// A NEW MESSAGE of the reference's circuit (its bits are circuit structure: paillier.rs:50-55 -> pow_mod_fixed_exp) from the library
    /// alone: the structure generated on the device (`pz_circuit_structure_dev`: what halo2's keygen would extract by synthesising
    /// bench.rs:33-75 once), the key built from its device arrays (`pz_pk_create_dev`: nothing crosses PCIe), the structure released.
    /// -> (key, column starts for `pz_circuit_expand_cols_dev` as a device pointer owner, n_adv, n_lk, n_steps_g, n_steps_r).
    /// `ext_resident_cols`: `usize::MAX` = extended key resident; 0 = the streamed proving key (BASELINE config c5: 3072-bit, k = 19).
    pub fn for_message(lagrange: *const pz_bases, monomial: *const pz_bases, k: u32, lookup_bits: u32, m: &[u64], n: &[u64],
                       minimum_rows: usize, ext_resident_cols: usize) -> (Self, MessageShape) {
        let mut st: *mut pz_structure = core::ptr::null_mut();
        check(unsafe { pz_circuit_structure_dev(ctx(), 0, n.len() as u32, 64, lookup_bits, k, m.as_ptr(), n.as_ptr(), minimum_rows, 6, &mut st) });
        let (mut n_adv, mut filled, mut n_lk, mut max_rows, mut n_const, mut cells, mut lks, mut sg, mut sr) = (0usize, 0usize, 0usize, 0usize, 0usize, 0usize, 0usize, 0usize, 0usize);
        check(unsafe { pz_structure_info(st, &mut n_adv, &mut filled, &mut n_lk, &mut max_rows, &mut n_const, &mut cells, &mut lks, &mut sg, &mut sr) });
        let (mut d_sel, mut d_mc, mut d_mr, mut d_starts, mut consts, mut starts_h) =
            (core::ptr::null::<u8>(), core::ptr::null::<u32>(), core::ptr::null::<u32>(), core::ptr::null::<u64>(), core::ptr::null::<u64>(), core::ptr::null::<u64>());
        check(unsafe { pz_structure_arrays(st, &mut d_sel, &mut d_mc, &mut d_mr, &mut d_starts, &mut consts, &mut starts_h) });
        let mut pk: *mut pz_pk = core::ptr::null_mut();
        check(unsafe {
            pz_pk_create_dev(ctx(), lagrange, monomial, k, lookup_bits, 6, max_rows, n_adv, n_lk, d_sel, consts, n_const, d_mc, d_mr, 64, ext_resident_cols, &mut pk)
        });
        // K4 needs the break points after the structure is gone: keep a device copy of the (n_adv + 1)-entry table
        let mut d_copy: *mut core::ffi::c_void = core::ptr::null_mut();
        check(unsafe { pz_dev_alloc(ctx(), (n_adv + 1) * 8, &mut d_copy) });
        check(unsafe { pz_upload(ctx(), d_copy, starts_h as *const core::ffi::c_void, (n_adv + 1) * 8) });
        check(unsafe { pz_sync(ctx()) });
        check(unsafe { pz_structure_free(st) });
        let (mut a, mut b, mut c, mut d, mut e) = (0usize, 0usize, 0usize, 0usize, 0usize);
        check(unsafe { pz_pk_info(pk, &mut a, &mut b, &mut c, &mut d, &mut e) });
        (DeviceKey { pk, n_fixed: a, n_perm_cols: b, n_sets: c, blinding_words: d, evals_words: e },
         MessageShape { d_col_starts: d_copy as *const u64, n_adv, n_lk, max_rows, n_steps_g: sg, n_steps_r: sr })
    }
}
/// what K3 / K4 need to write the witness of one message shape (`pz_paillier_encrypt_dev`, `pz_circuit_expand_cols_dev`)
pub struct MessageShape {
    pub d_col_starts: *const u64,
    pub n_adv: usize,
    pub n_lk: usize,
    pub max_rows: usize,
    pub n_steps_g: usize,
    pub n_steps_r: usize,
}

/// The two-context recipe: proofs are independent, so while `prove` runs proof i's seven phases on the key's
/// context, a SECOND host thread writes proof i + 1's witness (K3's four workgroups + K4's stores) into the other column slot on a second
/// context -- distinct `pz_ctx` run concurrently (pz.h), and the stepper blocks its own thread only.  `witness(i, ctx, d_cols)` queues
/// K3 + K4 of proof i on `ctx` and returns after `pz_sync(ctx)`; `prove(i, d_cols)` is the caller's `ProofSession` walk with its
/// transcript.  Measured through the same calls from Python threads at config c2: the stepper's serial 8xx ms per proof becomes the
/// compiled prover's 74x (profiles/r06_stepper_two_contexts.json).
pub fn prove_pipelined<W, P>(proofs: usize, slots: [*mut u64; 2], witness: W, mut prove: P)
where
    W: Fn(usize, *mut pz_ctx, *mut u64) + Sync,
    P: FnMut(usize, *mut u64),
{
    struct SendPtr(*mut u64);
    unsafe impl Send for SendPtr {}
    let wctx = new_ctx();
    witness(0, wctx, slots[0]);
    for i in 0..proofs {
        std::thread::scope(|s| {
            if i + 1 < proofs {
                let next = SendPtr(slots[(i + 1) & 1]);
                let w = &witness;
                let wc = Ctx(wctx);
                s.spawn(move || {
                    let next = next;
                    w(i + 1, wc.0, next.0)
                });
            }
            prove(i, slots[i & 1]);
        });
    }
    unsafe { pz_free(wctx) };
}

/// one proof in flight on a `DeviceKey`: every method runs a phase on the device and returns what the transcript absorbs before the next
/// challenge exists (points: 8 words each; evaluations: 4 words each, the family order of include/pz.h).  Challenges: Montgomery limbs of
/// `Fr` (`mont(&c)` = the in-memory representation of halo2curves' `Fr`).
pub struct ProofSession<'k> {
    p: *mut pz_proof,
    key: &'k DeviceKey,
    n_adv_lk: usize,
    n_lk: usize,
}
impl<'k> ProofSession<'k> {
    /// d_cols: the K4 columns on the device (`(n_adv + n_lk + 1) * 2^k` elements; consumed); random: `key.blinding_words` words of OS randomness
    pub fn begin(key: &'k DeviceKey, d_cols: *mut u64, n_adv: usize, n_lk: usize, random: &[u64]) -> (Self, Vec<u64>) {
        assert!(random.len() >= key.blinding_words);
        let mut p: *mut pz_proof = core::ptr::null_mut();
        let mut advice = vec![0u64; 8 * (n_adv + n_lk)];
        check(unsafe { pz_proof_begin(key.pk, d_cols, 0, random.as_ptr(), random.len(), &mut p, advice.as_mut_ptr()) });
        (ProofSession { p, key, n_adv_lk: n_adv + n_lk, n_lk }, advice)
    }
    pub fn lookups(&mut self, theta: &[u64; 4]) -> (Vec<u64>, Vec<u64>) {
        let (mut a, mut s) = (vec![0u64; 8 * self.n_lk], vec![0u64; 8 * self.n_lk]);
        check(unsafe { pz_proof_lookups(self.p, theta.as_ptr(), a.as_mut_ptr(), s.as_mut_ptr()) });
        (a, s)
    }
    pub fn products(&mut self, beta: &[u64; 4], gamma: &[u64; 4]) -> (Vec<u64>, Vec<u64>, [u64; 8]) {
        let (mut z, mut zl, mut rnd) = (vec![0u64; 8 * self.key.n_sets], vec![0u64; 8 * self.n_lk], [0u64; 8]);
        check(unsafe { pz_proof_products(self.p, beta.as_ptr(), gamma.as_ptr(), z.as_mut_ptr(), zl.as_mut_ptr(), rnd.as_mut_ptr()) });
        (z, zl, rnd)
    }
    pub fn quotient(&mut self, y: &[u64; 4]) -> [u64; 24] {
        let mut h = [0u64; 24];
        check(unsafe { pz_proof_quotient(self.p, y.as_ptr(), h.as_mut_ptr()) });
        h
    }
    pub fn evaluate(&mut self, x: &[u64; 4]) -> Vec<u64> {
        let mut e = vec![0u64; self.key.evals_words];
        check(unsafe { pz_proof_evaluate(self.p, x.as_ptr(), e.as_mut_ptr()) });
        e
    }
    pub fn open_begin(&mut self, y: &[u64; 4], v: &[u64; 4]) -> [u64; 8] {
        let mut w = [0u64; 8];
        check(unsafe { pz_proof_open_begin(self.p, y.as_ptr(), v.as_ptr(), w.as_mut_ptr()) });
        w
    }
    /// -> (second opening commitment, whether the quotient's degree is within 3n - 4: false = the witness does not satisfy the circuit)
    pub fn open_finish(&mut self, u: &[u64; 4]) -> ([u64; 8], bool) {
        let (mut w, mut ok) = ([0u64; 8], 0 as c_int);
        check(unsafe { pz_proof_open_finish(self.p, u.as_ptr(), w.as_mut_ptr(), &mut ok) });
        (w, ok != 0)
    }
    pub fn advice_points(&self) -> usize {
        self.n_adv_lk
    }
}
impl Drop for ProofSession<'_> {
    fn drop(&mut self) {
        unsafe { pz_proof_free(self.p) };
    }
}
