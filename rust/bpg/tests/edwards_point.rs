// The reference's conditional-selection round trip (ref: tests/edwards_point.rs:18-111) against the `bpg` crate.
// What differs from the reference's file is marked `// bpg:`; everything else is the reference's flow, so this file
// shows the compile-shape a maintainer keeps.  Needs an MI355X at run time.  Not compiled in this repository's image.
#![allow(non_snake_case)]
extern crate bpg;
extern crate curve25519_dalek;
extern crate rand;
extern crate zerocaf;

use bpg::gadgets::SonnyEdwardsPointGadget;                      // bpg: was bulletproofs_gadgets::gadgets::point::edwards_point::*
use bpg::r1cs::{Prover, R1CSError, R1CSProof, Verifier};        // bpg: was bulletproofs::r1cs::...
use bpg::{BulletproofGens, PedersenGens, Transcript};           // bpg: was bulletproofs::..., merlin::Transcript
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use zerocaf::edwards::EdwardsPoint as SonnyEdwardsPoint;

fn cond_select_proof(
    pc_gens: &PedersenGens,
    bp_gens: &BulletproofGens,
    P1: SonnyEdwardsPoint,
    P2: SonnyEdwardsPoint,
    selector: Scalar,
) -> Result<(R1CSProof, Vec<CompressedRistretto>), R1CSError> {
    let mut transcript = Transcript::new(b"Conditional_Selection!");
    let mut prover = Prover::new(pc_gens, &mut transcript);
    let (comm, cond) = prover.commit(selector, Scalar::random(&mut rand::thread_rng()));
    let p1_gadget = SonnyEdwardsPointGadget::from_point(&P1);
    // bpg: the struct literal of the reference becomes a constructor (the four LCs live inside the library)
    let p2_gadget = SonnyEdwardsPointGadget::from_scalars(
        Scalar::from_bytes_mod_order(P2.X.to_bytes()),
        Scalar::from_bytes_mod_order(P2.Y.to_bytes()),
        Scalar::from_bytes_mod_order(P2.Z.to_bytes()),
        Scalar::from_bytes_mod_order(P2.T.to_bytes()),
    );
    let p1_gadget = p1_gadget.conditionally_select(cond.into(), &mut prover);
    p1_gadget.equal(&p2_gadget, &mut prover);
    let proof = prover.prove(bp_gens)?;
    Ok((proof, vec![comm]))
}

fn cond_select_verify(
    pc_gens: &PedersenGens,
    bp_gens: &BulletproofGens,
    P1: SonnyEdwardsPoint,
    P2: SonnyEdwardsPoint,
    commitments: Vec<CompressedRistretto>,
    proof: R1CSProof,
) -> Result<(), R1CSError> {
    let mut transcript = Transcript::new(b"Conditional_Selection!");
    let mut verifier = Verifier::new(&mut transcript);
    let cond = verifier.commit(commitments[0]);
    let p1_gadget = SonnyEdwardsPointGadget::from_point(&P1);
    let p2_gadget = SonnyEdwardsPointGadget::from_scalars(
        Scalar::from_bytes_mod_order(P2.X.to_bytes()),
        Scalar::from_bytes_mod_order(P2.Y.to_bytes()),
        Scalar::from_bytes_mod_order(P2.Z.to_bytes()),
        Scalar::from_bytes_mod_order(P2.T.to_bytes()),
    );
    let p1_gadget = p1_gadget.conditionally_select(cond.into(), &mut verifier);
    p1_gadget.equal(&p2_gadget, &mut verifier);
    verifier.verify(&proof, &pc_gens, &bp_gens, &mut rand::thread_rng())?;
    Ok(())
}

fn cond_selection_roundtrip_helper(P1: SonnyEdwardsPoint, P2: SonnyEdwardsPoint, selector: Scalar) -> Result<(), R1CSError> {
    let pc_gens = PedersenGens::default();
    let bp_gens = BulletproofGens::new(64, 1);
    let (proof, commitments) = cond_select_proof(&pc_gens, &bp_gens, P1, P2, selector)?;
    cond_select_verify(&pc_gens, &bp_gens, P1, P2, commitments, proof)
}

#[test]
fn conditionally_select_test() {
    use zerocaf::traits::Identity;
    let P1 = SonnyEdwardsPoint::new_random_point(&mut rand::thread_rng());
    let ID = SonnyEdwardsPoint::identity();
    assert!(cond_selection_roundtrip_helper(P1, P1, Scalar::one()).is_ok());
    assert!(cond_selection_roundtrip_helper(P1, P1, Scalar::zero()).is_err());
    assert!(cond_selection_roundtrip_helper(P1, ID, Scalar::zero()).is_ok());
    assert!(cond_selection_roundtrip_helper(P1, ID, Scalar::one()).is_err());
}
