//! The `bulletproofs` crate as dusk-network/bulletproofs_gadgets sees it, over `libbpg.so`: the R1CS prover and
//! verifier of the reference's hot path on an AMD MI355X.
//!
//! Crate name, module paths, type names and call shapes are those the reference imports
//! (ref: src/gadgets/boolean.rs:1-5, scalar.rs:1-4, point/edwards_point.rs:2-4, point/ristretto_point.rs:2-4,
//! sk_knowledge/sk_know_gadget.rs:5-8, tests/edwards_point.rs:8-9): with the `[patch]` lines of `rust/README.md` the
//! reference's own `src/gadgets/**` and `tests/**` build unmodified and ARE the gadget layer; every
//! `cs.multiply` / `cs.constrain` / `cs.allocate_multiplier` they make crosses the C ABI.  `fast` holds the same
//! gadgets as single library calls (optional).  Not compiled in this repository's image (no Rust toolchain there).
#![allow(non_snake_case)]

pub mod fast;
pub mod r1cs;

use std::ffi::CStr;
use std::ptr;
use std::sync::Arc;

use bpg_sys as sys;

/// The process-wide context: one GPU per process (`BPG_DEVICE`, default 0), as SURVEY 8e lays the ranks out.
pub(crate) struct Context(pub(crate) *mut sys::bpg_ctx);
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

pub(crate) fn last_error() -> String {
    unsafe {
        let p = sys::bpg_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

pub(crate) fn context() -> &'static Context {
    use std::sync::Once;
    static INIT: Once = Once::new();
    static mut CTX: Option<Context> = None;
    unsafe {
        INIT.call_once(|| {
            let dev = std::env::var("BPG_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut h = ptr::null_mut();
            let rc = sys::bpg_init(dev, &mut h);
            // no CPU fallback: without an MI355X the library refuses, and so does this crate
            assert!(rc == sys::BPG_OK, "bpg_init({}) failed ({}): {}", dev, rc, last_error());
            CTX = Some(Context(h));
        });
        CTX.as_ref().unwrap()
    }
}

/// Owner of one `bpg_gens` (generators resident in HBM).
pub(crate) struct GensHandle(pub(crate) *mut sys::bpg_gens);
unsafe impl Send for GensHandle {}
unsafe impl Sync for GensHandle {}
impl Drop for GensHandle {
    fn drop(&mut self) {
        unsafe { sys::bpg_gens_free(self.0) }
    }
}

fn make_gens(capacity: usize) -> Arc<GensHandle> {
    let mut g = ptr::null_mut();
    let rc = unsafe { sys::bpg_gens_create(context().0, capacity, 1, &mut g) };
    assert!(rc == sys::BPG_OK, "bpg_gens_create({}) failed ({}): {}", capacity, rc, last_error());
    Arc::new(GensHandle(g))
}

/// `bulletproofs::PedersenGens` (ref: tests/edwards_point.rs:89).  Every generator object of the library carries
/// the default Pedersen bases; this one is the smallest.
#[derive(Clone)]
pub struct PedersenGens {
    pub(crate) gens: Arc<GensHandle>,
}
impl Default for PedersenGens {
    fn default() -> Self {
        PedersenGens { gens: make_gens(1) }
    }
}

/// `bulletproofs::BulletproofGens` (ref: tests/edwards_point.rs:90).  Built on the GPU, resident there.
#[derive(Clone)]
pub struct BulletproofGens {
    pub gens_capacity: usize,
    pub party_capacity: usize,
    pub(crate) gens: Arc<GensHandle>,
}
impl BulletproofGens {
    /// The reference only ever passes `party_capacity = 1` (R1CS proofs use `share(0)`); anything else panics.
    pub fn new(gens_capacity: usize, party_capacity: usize) -> Self {
        assert_eq!(party_capacity, 1, "R1CS proofs use a single party's generators");
        BulletproofGens { gens_capacity, party_capacity, gens: make_gens(gens_capacity) }
    }
}
