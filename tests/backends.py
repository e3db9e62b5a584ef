"""Backends for tests/ref_cases.py: the oracle (CPU) and the product (GPU via the C ABI)."""
from oracle import gadgets as ogadgets
from oracle import r1cs as orc1


class OracleBackend:
    VerificationError = orc1.VerificationError
    EdwardsGadget = ogadgets.SonnyEdwardsPointGadget
    RistrettoGadget = ogadgets.SonnyRistrettoPointGadget
    gadgets = ogadgets

    def __init__(self, orc):
        self.orc = orc
        self.pc_gens = orc1.PedersenGens(orc)
        self._gens = {}

    def bp_gens(self, cap):
        if cap not in self._gens:
            self._gens[cap] = orc1.BulletproofGens(self.orc, cap, 1)
        return self._gens[cap]

    def Transcript(self, label):
        return orc1.Transcript(self.orc, label)

    def Prover(self, tr):
        return orc1.Prover(self.orc, self.pc_gens, tr)

    def Verifier(self, tr):
        return orc1.Verifier(self.orc, tr)

    @staticmethod
    def lc(x):
        return orc1.LC.of(x)
