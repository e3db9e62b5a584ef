"""Backends for tests/ref_cases.py: the oracle (CPU) and the product (GPU via the C ABI)."""
from oracle import gadgets as ogadgets
from oracle import r1cs as orc1


class GpuBackend:
    """The product: every call goes through the C ABI of libbpg.so (HIP kernels on cuda:0)."""

    def __init__(self, ctx):
        from bulletproofs_gadgets_amd import r1cs as g
        import types
        self.g, self.ctx = g, ctx
        self.VerificationError = g.VerificationError
        self.EdwardsGadget = g.SonnyEdwardsPointGadget
        self.RistrettoGadget = g.SonnyRistrettoPointGadget
        self.gadgets = types.SimpleNamespace(binary_constrain_gadget=g.binary_constrain_gadget,
                                             nonzero_gadget=g.nonzero_gadget, sk_knowledge_gadget=g.sk_knowledge_gadget)
        self.pc_gens = None          # PedersenGens live inside the generator object
        self._gens = {}
        self._cur = None

    def bp_gens(self, cap):
        if cap not in self._gens:
            self._gens[cap] = self.g.BulletproofGens(self.ctx, cap, 1)
        self._cur = self._gens[cap]
        return self._cur

    def Transcript(self, label):
        return label                 # the ABI's Prover / Verifier own their transcript

    def Prover(self, label):
        return self.g.Prover(self._cur, label)

    def Verifier(self, label):
        return self.g.Verifier(label)

    def lc(self, x):
        return self.g.LinearCombination.of(x)


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
