//! `bulletproofs::r1cs` as the reference uses it.
use std::marker::PhantomData;
use std::ops::{Add, Mul, Neg, Sub};
use std::os::raw::{c_int, c_void};
use std::ptr;

use bpg_sys as sys;
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use rand::{CryptoRng, RngCore};

use merlin::Transcript;

use crate::{BulletproofGens, PedersenGens};

/// `bulletproofs::r1cs::R1CSError` (the reference matches on `VerificationError`, tests/edwards_point.rs:346).
#[derive(Clone, Debug, Eq, PartialEq)]
pub enum R1CSError {
    InvalidGeneratorsLength,
    FormatError,
    VerificationError,
    MissingAssignment,
    GadgetError { description: String },
}

pub(crate) fn check(rc: c_int) -> Result<(), R1CSError> {
    match rc {
        sys::BPG_OK => Ok(()),
        sys::BPG_EGENS => Err(R1CSError::InvalidGeneratorsLength),
        sys::BPG_EFORMAT => Err(R1CSError::FormatError),
        sys::BPG_EVERIFY => Err(R1CSError::VerificationError),
        sys::BPG_EMISSING => Err(R1CSError::MissingAssignment),
        sys::BPG_EGADGET => Err(R1CSError::GadgetError { description: crate::last_error() }),
        // bad arguments / HIP failures are bugs or a missing GPU, not proof outcomes: panic as dalek would
        other => panic!("libbpg error {}: {}", other, crate::last_error()),
    }
}

/// `bulletproofs::r1cs::Variable`
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
pub enum Variable {
    Committed(usize),
    MultiplierLeft(usize),
    MultiplierRight(usize),
    MultiplierOutput(usize),
    One(),
}
impl Variable {
    pub(crate) fn pack(self) -> u64 {
        match self {
            Variable::Committed(i) => i as u64,
            Variable::MultiplierLeft(i) => (1u64 << 32) | i as u64,
            Variable::MultiplierRight(i) => (2u64 << 32) | i as u64,
            Variable::MultiplierOutput(i) => (3u64 << 32) | i as u64,
            Variable::One() => 4u64 << 32,
        }
    }
    pub(crate) fn unpack(v: u64) -> Variable {
        let i = (v & 0xffff_ffff) as usize;
        match v >> 32 {
            0 => Variable::Committed(i),
            1 => Variable::MultiplierLeft(i),
            2 => Variable::MultiplierRight(i),
            3 => Variable::MultiplierOutput(i),
            _ => Variable::One(),
        }
    }
}

/// `bulletproofs::r1cs::LinearCombination`
#[derive(Clone, Debug, Default)]
pub struct LinearCombination {
    pub(crate) terms: Vec<(Variable, Scalar)>,
}
impl LinearCombination {
    pub(crate) fn marshal(&self) -> (Vec<u64>, Vec<u8>) {
        let vars = self.terms.iter().map(|(v, _)| v.pack()).collect();
        let mut coeffs = Vec::with_capacity(32 * self.terms.len());
        for (_, c) in &self.terms {
            coeffs.extend_from_slice(c.as_bytes());
        }
        (vars, coeffs)
    }
}
impl From<Variable> for LinearCombination {
    fn from(v: Variable) -> Self {
        LinearCombination { terms: vec![(v, Scalar::one())] }
    }
}
impl From<Scalar> for LinearCombination {
    fn from(s: Scalar) -> Self {
        LinearCombination { terms: vec![(Variable::One(), s)] }
    }
}
impl<L: Into<LinearCombination>> Add<L> for LinearCombination {
    type Output = Self;
    fn add(mut self, rhs: L) -> Self {
        self.terms.extend(rhs.into().terms);
        self
    }
}
impl<L: Into<LinearCombination>> Sub<L> for LinearCombination {
    type Output = Self;
    fn sub(mut self, rhs: L) -> Self {
        self.terms.extend(rhs.into().terms.into_iter().map(|(v, c)| (v, -c)));
        self
    }
}
impl Neg for LinearCombination {
    type Output = Self;
    fn neg(mut self) -> Self {
        for (_, c) in self.terms.iter_mut() {
            *c = -*c;
        }
        self
    }
}
impl Mul<LinearCombination> for Scalar {
    type Output = LinearCombination;
    fn mul(self, mut lc: LinearCombination) -> LinearCombination {
        for (_, c) in lc.terms.iter_mut() {
            *c = *c * self;
        }
        lc
    }
}
impl<L: Into<LinearCombination>> Add<L> for Variable {
    type Output = LinearCombination;
    fn add(self, rhs: L) -> LinearCombination {
        LinearCombination::from(self) + rhs
    }
}
impl<L: Into<LinearCombination>> Sub<L> for Variable {
    type Output = LinearCombination;
    fn sub(self, rhs: L) -> LinearCombination {
        LinearCombination::from(self) - rhs
    }
}
impl Mul<Variable> for Scalar {
    type Output = LinearCombination;
    fn mul(self, v: Variable) -> LinearCombination {
        LinearCombination { terms: vec![(v, self)] }
    }
}

/// `bulletproofs::r1cs::ConstraintSystem`: what the reference's gadgets take as `&mut dyn ConstraintSystem`
/// (ref: src/gadgets/boolean.rs:12, scalar.rs:9, sk_know_gadget.rs:14).
pub trait ConstraintSystem {
    #[doc(hidden)]
    fn raw(&mut self) -> *mut sys::bpg_cs;

    fn multiply(&mut self, left: LinearCombination, right: LinearCombination) -> (Variable, Variable, Variable) {
        let (lv, lc) = left.marshal();
        let (rv, rc) = right.marshal();
        let mut out = [0u64; 3];
        let code = unsafe {
            sys::bpg_cs_multiply(self.raw(), lv.as_ptr(), lc.as_ptr(), lv.len(), rv.as_ptr(), rc.as_ptr(), rv.len(), out.as_mut_ptr())
        };
        check(code).expect("multiply");
        (Variable::unpack(out[0]), Variable::unpack(out[1]), Variable::unpack(out[2]))
    }

    fn allocate_multiplier(&mut self, input_assignments: Option<(Scalar, Scalar)>) -> Result<(Variable, Variable, Variable), R1CSError> {
        let mut out = [0u64; 3];
        let code = unsafe {
            match input_assignments {
                Some((l, r)) => sys::bpg_cs_allocate_multiplier(self.raw(), l.as_bytes().as_ptr(), r.as_bytes().as_ptr(), out.as_mut_ptr()),
                None => sys::bpg_cs_allocate_multiplier(self.raw(), ptr::null(), ptr::null(), out.as_mut_ptr()),
            }
        };
        check(code)?;
        Ok((Variable::unpack(out[0]), Variable::unpack(out[1]), Variable::unpack(out[2])))
    }

    fn constrain(&mut self, lc: LinearCombination) {
        let (v, c) = lc.marshal();
        check(unsafe { sys::bpg_cs_constrain(self.raw(), v.as_ptr(), c.as_ptr(), v.len()) }).expect("constrain");
    }
}

/// `bulletproofs::r1cs::RandomizedConstraintSystem` (feature `yoloproofs`; imported by the reference at
/// src/gadgets/point/ristretto_point.rs:3): the constraint system handed to a randomized-constraints callback.
pub trait RandomizedConstraintSystem: ConstraintSystem {
    fn challenge_scalar(&mut self, label: &'static [u8]) -> Scalar {
        let mut out = [0u8; 32];
        check(unsafe { sys::bpg_cs_challenge_scalar(self.raw(), label.as_ptr(), label.len(), out.as_mut_ptr()) }).expect("challenge_scalar");
        Scalar::from_bits(out)
    }
}

/// The `rcs` of a callback: valid during that callback only.
pub struct RandomizingCS<'a> {
    raw: *mut sys::bpg_cs,
    _life: PhantomData<&'a mut ()>,
}
impl<'a> ConstraintSystem for RandomizingCS<'a> {
    fn raw(&mut self) -> *mut sys::bpg_cs {
        self.raw
    }
}
impl<'a> RandomizedConstraintSystem for RandomizingCS<'a> {}

type Callback = Box<dyn FnMut(&mut RandomizingCS) -> Result<(), R1CSError>>;

unsafe extern "C" fn randomized_trampoline(user: *mut c_void, rcs: *mut sys::bpg_cs) -> c_int {
    let cb = &mut *(user as *mut Callback);
    let mut cs = RandomizingCS { raw: rcs, _life: PhantomData };
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| cb(&mut cs))) {
        Ok(Ok(())) => sys::BPG_OK,
        Ok(Err(R1CSError::MissingAssignment)) => sys::BPG_EMISSING,
        Ok(Err(R1CSError::GadgetError { .. })) => sys::BPG_EGADGET,
        Ok(Err(_)) => sys::BPG_EINVAL,
        Err(_) => sys::BPG_EINVAL, // never unwind into C frames
    }
}

/// `bulletproofs::r1cs::RandomizableConstraintSystem`
pub trait RandomizableConstraintSystem: ConstraintSystem {
    #[doc(hidden)]
    fn keep(&mut self) -> &mut Vec<Box<Callback>>;

    /// `callback` runs inside `prove` / `verify`, after the first-phase commitments are in the transcript.
    fn specify_randomized_constraints<F>(&mut self, callback: F) -> Result<(), R1CSError>
    where
        F: 'static + FnMut(&mut RandomizingCS) -> Result<(), R1CSError>,
    {
        let mut boxed: Box<Callback> = Box::new(Box::new(callback));
        let user = &mut *boxed as *mut Callback as *mut c_void;
        self.keep().push(boxed); // the library holds `user` until prove / verify has run
        check(unsafe { sys::bpg_cs_specify_randomized_constraints(self.raw(), Some(randomized_trampoline), user) })
    }
}

unsafe extern "C" fn rng_trampoline<R: RngCore + CryptoRng>(user: *mut c_void, out: *mut u8, n: usize) {
    let rng = &mut *(user as *mut R);
    rng.fill_bytes(std::slice::from_raw_parts_mut(out, n));
}

/// `bulletproofs::r1cs::R1CSProof`: the wire bytes (`to_bytes` / `from_bytes` as in bulletproofs).
#[derive(Clone, Debug, Eq, PartialEq)]
pub struct R1CSProof(pub(crate) Vec<u8>);
impl R1CSProof {
    pub fn to_bytes(&self) -> Vec<u8> {
        self.0.clone()
    }
    pub fn from_bytes(slice: &[u8]) -> Result<R1CSProof, R1CSError> {
        // the library parses (and rejects) at verification time; the cheap structural check happens here
        if slice.is_empty() || slice[0] > 1 || (slice.len() - 1) % 32 != 0 {
            return Err(R1CSError::FormatError);
        }
        Ok(R1CSProof(slice.to_vec()))
    }
}

/// `bulletproofs::r1cs::Prover` (ref: tests/edwards_point.rs:28-47)
pub struct Prover<'a, 'b> {
    raw: *mut sys::bpg_prover,
    callbacks: Vec<Box<Callback>>,
    _life: PhantomData<(&'a mut Transcript, &'b PedersenGens)>,
}
impl<'a, 'b> Prover<'a, 'b> {
    pub fn new(pc_gens: &'b PedersenGens, transcript: &'a mut Transcript) -> Self {
        let mut raw = ptr::null_mut();
        let rc = unsafe { sys::bpg_prover_new(pc_gens.gens.0, transcript.bpg_label().as_ptr(), transcript.bpg_label().len(), &mut raw) };
        check(rc).expect("Prover::new");
        Prover { raw, callbacks: Vec::new(), _life: PhantomData }
    }

    pub fn commit(&mut self, v: Scalar, v_blinding: Scalar) -> (CompressedRistretto, Variable) {
        let (mut V, mut var) = ([0u8; 32], 0u64);
        let rc = unsafe { sys::bpg_prover_commit(self.raw, v.as_bytes().as_ptr(), v_blinding.as_bytes().as_ptr(), V.as_mut_ptr(), &mut var) };
        check(rc).expect("Prover::commit");
        (CompressedRistretto(V), Variable::unpack(var))
    }

    /// Not in upstream's API (optional; for one large proof).  Upstream's `prove` asks `thread_rng()` for 32 bytes and
    /// then draws 2n + 3 blinding scalars from the transcript RNG, one Keccak permutation each, on the calling thread.
    /// This call asks `thread_rng()` for those 32 bytes NOW -- after the last `commit`, before the gadget calls -- and
    /// the library draws on a helper thread while the caller builds the circuit (`bpg_prover_begin_blindings`);
    /// `prove` then uses those bytes.  The proof has the same distribution; a later `commit` simply stops the helper.
    pub fn begin_blindings(&mut self) -> Result<(), R1CSError> {
        let mut ext = [0u8; 32];
        rand::thread_rng().fill_bytes(&mut ext);
        check(unsafe { sys::bpg_prover_begin_blindings(self.raw, ext.as_ptr()) })
    }

    /// `prove(bp_gens)`: external randomness from `rand::thread_rng()`, as in bulletproofs.
    pub fn prove(self, bp_gens: &BulletproofGens) -> Result<R1CSProof, R1CSError> {
        let mut rng = rand::thread_rng();
        self.prove_with_rng(bp_gens, &mut rng)
    }

    pub fn prove_with_rng<R: RngCore + CryptoRng>(self, bp_gens: &BulletproofGens, rng: &mut R) -> Result<R1CSProof, R1CSError> {
        check(unsafe { sys::bpg_prover_set_gens(self.raw, bp_gens.gens.0) })?;
        let mut out = vec![0u8; 1 << 16];
        let mut len = 0usize;
        let rc = unsafe {
            sys::bpg_prover_prove(self.raw, Some(rng_trampoline::<R>), rng as *mut R as *mut c_void, out.as_mut_ptr(), out.len(), &mut len)
        };
        check(rc)?;
        out.truncate(len);
        Ok(R1CSProof(out))
    }

    pub(crate) fn raw_prover(&mut self) -> *mut sys::bpg_prover {
        self.raw
    }
}
impl<'a, 'b> ConstraintSystem for Prover<'a, 'b> {
    fn raw(&mut self) -> *mut sys::bpg_cs {
        unsafe { sys::bpg_prover_cs(self.raw) }
    }
}
impl<'a, 'b> RandomizableConstraintSystem for Prover<'a, 'b> {
    fn keep(&mut self) -> &mut Vec<Box<Callback>> {
        &mut self.callbacks
    }
}
impl<'a, 'b> Drop for Prover<'a, 'b> {
    fn drop(&mut self) {
        unsafe { sys::bpg_prover_free(self.raw) }
    }
}

/// `bulletproofs::r1cs::Verifier` (ref: tests/edwards_point.rs:61-79)
pub struct Verifier<'a> {
    raw: *mut sys::bpg_verifier,
    callbacks: Vec<Box<Callback>>,
    _life: PhantomData<&'a mut Transcript>,
}
impl<'a> Verifier<'a> {
    pub fn new(transcript: &'a mut Transcript) -> Self {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::bpg_verifier_new(transcript.bpg_label().as_ptr(), transcript.bpg_label().len(), &mut raw) }).expect("Verifier::new");
        Verifier { raw, callbacks: Vec::new(), _life: PhantomData }
    }

    pub fn commit(&mut self, commitment: CompressedRistretto) -> Variable {
        let mut var = 0u64;
        check(unsafe { sys::bpg_verifier_commit(self.raw, commitment.as_bytes().as_ptr(), &mut var) }).expect("Verifier::commit");
        Variable::unpack(var)
    }

    /// `verify(&proof, &pc_gens, &bp_gens, &mut rng)` (ref: tests/edwards_point.rs:79)
    pub fn verify<R: RngCore + CryptoRng>(self, proof: &R1CSProof, _pc_gens: &PedersenGens, bp_gens: &BulletproofGens, rng: &mut R) -> Result<(), R1CSError> {
        let rc = unsafe {
            sys::bpg_verifier_verify(self.raw, bp_gens.gens.0, proof.0.as_ptr(), proof.0.len(), Some(rng_trampoline::<R>),
                                     rng as *mut R as *mut c_void)
        };
        check(rc)
    }

    pub(crate) fn raw_verifier(&mut self) -> *mut sys::bpg_verifier {
        self.raw
    }
}
impl<'a> ConstraintSystem for Verifier<'a> {
    fn raw(&mut self) -> *mut sys::bpg_cs {
        unsafe { sys::bpg_verifier_cs(self.raw) }
    }
}
impl<'a> RandomizableConstraintSystem for Verifier<'a> {
    fn keep(&mut self) -> &mut Vec<Box<Callback>> {
        &mut self.callbacks
    }
}
impl<'a> Drop for Verifier<'a> {
    fn drop(&mut self) {
        unsafe { sys::bpg_verifier_free(self.raw) }
    }
}
