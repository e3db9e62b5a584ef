//! `merlin::Transcript` as dusk-network/bulletproofs_gadgets uses it (ref: Cargo.toml:11 `merlin = "2.0.0"`).
//!
//! The reference creates a transcript with a label and hands it, untouched, to `Prover::new` / `Verifier::new`
//! (ref: tests/edwards_point.rs:25-28,59-61; src/gadgets/scalar.rs:38-40,56-57;
//! src/gadgets/sk_knowledge/sk_know_gadget.rs:48-50,76-77).  The STROBE-128 state itself lives inside `libbpg.so`
//! (one Keccak chain per proof, on the host); this type carries what the library needs to start it: the label.
//! `append_message` / `challenge_bytes` on a transcript the caller still holds are not offered: nothing in the
//! reference calls them, and a prefix appended before `Prover::new` would have to be replayed inside the library
//! ahead of its `r1cs v1` domain separator (no entry point for that yet).
//!
//! Not compiled in this repository's image (no Rust toolchain there); `tests/test_abi.py` checks the names the
//! reference imports against this file.

/// `merlin::Transcript`
#[derive(Clone, Debug)]
pub struct Transcript {
    label: &'static [u8],
}

impl Transcript {
    /// `Transcript::new(label)` -- the only method the reference calls.
    pub fn new(label: &'static [u8]) -> Transcript {
        Transcript { label }
    }

    #[doc(hidden)]
    pub fn bpg_label(&self) -> &'static [u8] {
        self.label
    }
}
