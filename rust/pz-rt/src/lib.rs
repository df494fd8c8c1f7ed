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
pub const ABI: c_int = 5;

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
