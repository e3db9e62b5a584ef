//! OPTIONAL fast path, not part of the `bulletproofs` API: the reference's gadgets (`bulletproofs_gadgets::gadgets`) as
//! single library calls.  The circuits are built inside `libbpg.so` call for call as the reference builds them
//! (ref: src/gadgets/**), so multiplier and constraint numbering -- and with it the proof bytes -- are the
//! reference's; what a caller saves is one FFI crossing per `multiply` / `constrain` (27 multipliers per bit of a
//! scalar multiplication).  A gadget here is an opaque handle to four linear combinations held by the library: code
//! that builds `SonnyEdwardsPointGadget { X, Y, Z, T }` literals (ref: tests/edwards_point.rs:36-41) stays on the
//! reference's own gadget source over `crate::r1cs`.
use std::ptr;

use bpg_sys as sys;
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use zerocaf::edwards::EdwardsPoint as SonnyEdwardsPoint;
use zerocaf::ristretto::RistrettoPoint as SonnyRistrettoPoint;

use crate::r1cs::{check, ConstraintSystem, LinearCombination, Prover, Variable, Verifier};

/// Four linear combinations held by the library.
struct PointHandle(*mut sys::bpg_point_gadget);
impl Drop for PointHandle {
    fn drop(&mut self) {
        unsafe { sys::bpg_point_free(self.0) }
    }
}

/// X | Y | Z | T as 32-byte little-endian Sonny field elements (`FieldElement::to_bytes`), the value bridge of
/// `Scalar::from_bytes_mod_order(fe.to_bytes())` (ref: edwards_point.rs:21-24).
fn coords(p: &SonnyEdwardsPoint) -> [u8; 128] {
    let mut out = [0u8; 128];
    out[0..32].copy_from_slice(&p.X.to_bytes());
    out[32..64].copy_from_slice(&p.Y.to_bytes());
    out[64..96].copy_from_slice(&p.Z.to_bytes());
    out[96..128].copy_from_slice(&p.T.to_bytes());
    out
}

fn marshal_bit(bit: &LinearCombination) -> (Vec<u64>, Vec<u8>) {
    bit.marshal()
}

/// ref: src/gadgets/boolean.rs:12-21
pub fn binary_constrain_gadget(cs: &mut dyn ConstraintSystem, bit: Variable) {
    check(unsafe { sys::bpg_gadget_binary_constrain(cs.raw(), bit.pack()) }).expect("binary_constrain_gadget");
}

/// ref: src/gadgets/scalar.rs:9-26 (panics on a zero assignment, as the reference does at :13-15)
pub fn nonzero_gadget(var: LinearCombination, var_assignment: Option<Scalar>, cs: &mut dyn ConstraintSystem) {
    let (v, c) = var.marshal();
    let a = var_assignment.map(|s| *s.as_bytes());
    let ap = a.as_ref().map(|b| b.as_ptr()).unwrap_or(ptr::null());
    check(unsafe { sys::bpg_gadget_nonzero(cs.raw(), v.as_ptr(), c.as_ptr(), v.len(), ap) }).expect("nonzero_gadget");
}

/// ref: src/gadgets/point/edwards_point.rs:10-16
pub struct SonnyEdwardsPointGadget {
    h: PointHandle,
}

impl SonnyEdwardsPointGadget {
    /// ref: edwards_point.rs:19-26
    pub fn from_point(point: &SonnyEdwardsPoint) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_point_from_coords(coords(point).as_ptr(), &mut h) }).expect("from_point");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// the `SonnyEdwardsPointGadget { X: .., Y: .., Z: .., T: .. }` literal of the reference's tests
    /// (tests/edwards_point.rs:36-41), for coordinates given as scalars
    pub fn from_scalars(X: Scalar, Y: Scalar, Z: Scalar, T: Scalar) -> Self {
        let mut b = [0u8; 128];
        for (k, s) in [X, Y, Z, T].iter().enumerate() {
            b[32 * k..32 * k + 32].copy_from_slice(s.as_bytes());
        }
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_point_from_coords(b.as_ptr(), &mut h) }).expect("from_scalars");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    pub fn from_vars(vars: [Variable; 4]) -> Self {
        let v: Vec<u64> = vars.iter().map(|x| x.pack()).collect();
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_point_from_vars(v.as_ptr(), &mut h) }).expect("from_vars");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// ref: edwards_point.rs:28-93
    pub fn add(&self, other: &SonnyEdwardsPointGadget, cs: &mut dyn ConstraintSystem) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_add(cs.raw(), self.h.0, other.h.0, &mut h) }).expect("add");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// ref: edwards_point.rs:97-137
    pub fn double(&self, cs: &mut dyn ConstraintSystem) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_double(cs.raw(), self.h.0, &mut h) }).expect("double");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// ref: edwards_point.rs:140-163
    pub fn scalar_mul(point: &SonnyEdwardsPointGadget, sk: &[Variable], cs: &mut dyn ConstraintSystem) -> Self {
        let v: Vec<u64> = sk.iter().map(|x| x.pack()).collect();
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_scalar_mul(cs.raw(), point.h.0, v.as_ptr(), v.len(), &mut h) }).expect("scalar_mul");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// ref: edwards_point.rs:166-174
    pub fn equal(&self, other: &SonnyEdwardsPointGadget, cs: &mut dyn ConstraintSystem) {
        check(unsafe { sys::bpg_edwards_equal(cs.raw(), self.h.0, other.h.0) }).expect("equal");
    }
    /// ref: edwards_point.rs:178-211
    pub fn satisfy_curve_eq(&self, cs: &mut dyn ConstraintSystem) {
        check(unsafe { sys::bpg_edwards_satisfy_curve_eq(cs.raw(), self.h.0) }).expect("satisfy_curve_eq");
    }
    /// ref: edwards_point.rs:215-253
    pub fn conditionally_select(&self, bit: LinearCombination, cs: &mut dyn ConstraintSystem) -> Self {
        let (v, c) = marshal_bit(&bit);
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_conditionally_select(cs.raw(), self.h.0, v.as_ptr(), c.as_ptr(), v.len(), &mut h) })
            .expect("conditionally_select");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
    /// ref: edwards_point.rs:255-276 (four commitments: X, Y, Z, T)
    pub fn prover_commit_to_sonny_edwards_point(prover: &mut Prover, p: &SonnyEdwardsPoint) -> (Self, Vec<CompressedRistretto>) {
        let mut rng = rand::thread_rng();
        let mut bl = [0u8; 128];
        for k in 0..4 {
            bl[32 * k..32 * k + 32].copy_from_slice(Scalar::random(&mut rng).as_bytes());
        }
        let mut V = [0u8; 128];
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_prover_commit(prover.raw_prover(), coords(p).as_ptr(), bl.as_ptr(), &mut h, V.as_mut_ptr()) })
            .expect("prover_commit_to_sonny_edwards_point");
        let comms = (0..4)
            .map(|k| {
                let mut c = [0u8; 32];
                c.copy_from_slice(&V[32 * k..32 * k + 32]);
                CompressedRistretto(c)
            })
            .collect();
        (SonnyEdwardsPointGadget { h: PointHandle(h) }, comms)
    }
    /// ref: edwards_point.rs:278-290 (asserts four commitments, as the reference does at :282)
    pub fn verifier_commit_to_sonny_edwards_point(verifier: &mut Verifier, commitments: &[CompressedRistretto]) -> Self {
        let mut V = Vec::with_capacity(32 * commitments.len());
        for c in commitments {
            V.extend_from_slice(c.as_bytes());
        }
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_edwards_verifier_commit(verifier.raw_verifier(), V.as_ptr(), commitments.len(), &mut h) })
            .expect("verifier_commit_to_sonny_edwards_point");
        SonnyEdwardsPointGadget { h: PointHandle(h) }
    }
}

/// ref: src/gadgets/point/ristretto_point.rs:14-20
pub struct SonnyRistrettoPointGadget {
    h: PointHandle,
}

impl SonnyRistrettoPointGadget {
    /// ref: ristretto_point.rs:23-32
    pub fn from_point(point: SonnyRistrettoPoint, cs: &mut dyn ConstraintSystem) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_ristretto_from_point(cs.raw(), coords(&point.0).as_ptr(), &mut h) }).expect("from_point");
        SonnyRistrettoPointGadget { h: PointHandle(h) }
    }
    /// the identity without the Ristretto constraint (ref: sk_know_gadget.rs:20-25)
    pub fn identity_unconstrained() -> Self {
        let mut b = [0u8; 128];
        b[32] = 1;
        b[64] = 1;
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_point_from_coords(b.as_ptr(), &mut h) }).expect("identity");
        SonnyRistrettoPointGadget { h: PointHandle(h) }
    }
    /// ref: ristretto_point.rs:92-168
    pub fn add(&self, cs: &mut dyn ConstraintSystem, other: SonnyRistrettoPointGadget) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_ristretto_add(cs.raw(), self.h.0, other.h.0, &mut h) }).expect("add");
        SonnyRistrettoPointGadget { h: PointHandle(h) }
    }
    /// ref: ristretto_point.rs:177-180
    pub fn double(&self, cs: &mut dyn ConstraintSystem) -> Self {
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_ristretto_double(cs.raw(), self.h.0, &mut h) }).expect("double");
        SonnyRistrettoPointGadget { h: PointHandle(h) }
    }
    /// ref: ristretto_point.rs:171-175
    pub fn equals(&self, cs: &mut dyn ConstraintSystem, other: SonnyRistrettoPointGadget) {
        check(unsafe { sys::bpg_ristretto_equals(cs.raw(), self.h.0, other.h.0) }).expect("equals");
    }
    /// ref: ristretto_point.rs:184-226
    pub fn conditionally_select(&self, bit: LinearCombination, cs: &mut dyn ConstraintSystem) -> Self {
        let (v, c) = marshal_bit(&bit);
        let mut h = ptr::null_mut();
        check(unsafe { sys::bpg_ristretto_conditionally_select(cs.raw(), self.h.0, v.as_ptr(), c.as_ptr(), v.len(), &mut h) })
            .expect("conditionally_select");
        SonnyRistrettoPointGadget { h: PointHandle(h) }
    }
}

/// ref: src/gadgets/sk_knowledge/sk_know_gadget.rs:13-38 (the whole loop runs inside the library, call for call)
pub fn sk_knowledge_gadget(cs: &mut dyn ConstraintSystem, basep: SonnyRistrettoPointGadget, pk: SonnyRistrettoPointGadget, sk: Vec<Variable>) {
    let v: Vec<u64> = sk.iter().map(|x| x.pack()).collect();
    check(unsafe { sys::bpg_sk_knowledge_gadget(cs.raw(), basep.h.0, pk.h.0, v.as_ptr(), v.len()) }).expect("sk_knowledge_gadget");
}
