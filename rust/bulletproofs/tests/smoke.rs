//! Smoke test of the stand-in crates, in the shape of a caller of `bulletproofs::r1cs` (not a copy of any reference
//! test): a prover shows that it knows x, y with x * y = 35 and x + y = 12 without revealing them; a verifier checks
//! the proof, and rejects it against the statement x + y = 13.  Needs an MI355X and `libbpg.so` (see ../../README.md);
//! has not been compiled in this repository's image.
use bulletproofs::r1cs::{ConstraintSystem, LinearCombination, Prover, R1CSError, R1CSProof, Variable, Verifier};
use bulletproofs::{BulletproofGens, PedersenGens};
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use merlin::Transcript;

/// x * y = product, x + y = sum
fn product_and_sum(cs: &mut dyn ConstraintSystem, x: Variable, y: Variable, product: u64, sum: u64) {
    let (_, _, out) = cs.multiply(x.into(), y.into());
    cs.constrain(out - LinearCombination::from(Scalar::from(product)));
    cs.constrain(x + y - LinearCombination::from(Scalar::from(sum)));
}

fn prove(pc: &PedersenGens, bp: &BulletproofGens, x: u64, y: u64) -> Result<(R1CSProof, [CompressedRistretto; 2]), R1CSError> {
    let mut transcript = Transcript::new(b"bpg smoke");
    let mut prover = Prover::new(pc, &mut transcript);
    let mut rng = rand::thread_rng();
    let (cx, vx) = prover.commit(Scalar::from(x), Scalar::random(&mut rng));
    let (cy, vy) = prover.commit(Scalar::from(y), Scalar::random(&mut rng));
    product_and_sum(&mut prover, vx, vy, 35, 12);
    Ok((prover.prove(bp)?, [cx, cy]))
}

fn verify(pc: &PedersenGens, bp: &BulletproofGens, proof: &R1CSProof, c: &[CompressedRistretto; 2], sum: u64) -> Result<(), R1CSError> {
    let mut transcript = Transcript::new(b"bpg smoke");
    let mut verifier = Verifier::new(&mut transcript);
    let vx = verifier.commit(c[0]);
    let vy = verifier.commit(c[1]);
    product_and_sum(&mut verifier, vx, vy, 35, sum);
    verifier.verify(proof, pc, bp, &mut rand::thread_rng())
}

#[test]
fn product_and_sum_round_trip() {
    let (pc, bp) = (PedersenGens::default(), BulletproofGens::new(16, 1));
    let (proof, c) = prove(&pc, &bp, 5, 7).unwrap();
    assert!(verify(&pc, &bp, &proof, &c, 12).is_ok());
    assert_eq!(verify(&pc, &bp, &proof, &c, 13), Err(R1CSError::VerificationError));
    // the wire format round-trips
    let again = R1CSProof::from_bytes(&proof.to_bytes()).unwrap();
    assert!(verify(&pc, &bp, &again, &c, 12).is_ok());
    // a circuit with two multipliers needs two generators per side
    let mut t = Transcript::new(b"bpg smoke");
    let mut prover = Prover::new(&pc, &mut t);
    let (_, vx) = prover.commit(Scalar::from(2u64), Scalar::from(1u64));
    let (_, _, sq) = prover.multiply(vx.into(), vx.into());
    let _ = prover.multiply(sq.into(), vx.into());
    assert_eq!(prover.prove(&BulletproofGens::new(1, 1)).map(|_| ()), Err(R1CSError::InvalidGeneratorsLength));
}
