// r1cs.hip -- Prover::prove / Verifier::verify / Prover::commit of bulletproofs::r1cs
// on MI355X (ref call sites: tests/edwards_point.rs:31,47,79; see r1cs.h), plus the
// scalar-vector kernels mod l (constraint flattening, l(x)/r(x)/t(x), inner-product
// folding) and the fixed-base Pedersen kernel.
//
// Design:
//  * Every group operation of the proof is a fixed-base MSM over the resident generator
//    table [G | H | B | B~] (msm.hip).  The inner-product argument never folds points:
//    the folded generators of round k are G'_i = sum_{j = i mod n_k} sG_j G_j with
//    sG_j = G_factor_j * prod_{t<k} u_t^{+-1}, so L_k and R_k are MSMs over the ORIGINAL
//    table with scalars a'_{.}*sG_j (and b'_{.}*sH_j): group results are exact, so the
//    bytes equal what per-round folding (dalek) produces, with no 253-doubling chain per
//    round on the GPU's slow single-lane path.
//  * Scalars mod l live in HBM as 32-byte little-endian values; one thread per element.
//  * The transcript (one Keccak chain) and the blinding RNG stay on the host.
#include <chrono>
#include <cstdio>
#include <cstring>

#include <hip/hip_ext.h>

#include "r1cs.h"

namespace bpg {

// =================================================================== device helpers

__device__ __forceinline__ sc ld_sc(const sc *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1];
    sc r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ __forceinline__ void st_sc(sc *p, const sc &v) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    q[1] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
}
__device__ __forceinline__ sc mul_(const sc &a, const sc &b) { sc r; sc_mul(r, a, b); return r; }
__device__ __forceinline__ sc add_(const sc &a, const sc &b) { sc r; sc_add(r, a, b); return r; }
__device__ __forceinline__ sc sub_(const sc &a, const sc &b) { sc r; sc_sub(r, a, b); return r; }

// Montgomery form (x R mod l, R = 2^256) is how these kernels halve their work: a plain product is two reductions
// (sc_mul = montmul, montmul by R^2), but montmul(a, b R) = a b is one.  So per-kernel constants (challenges, their
// powers) arrive as x R, power tables hold x^(2^k) R and a chain of products stays in that form, the inner-product
// vectors a, b live in HBM as a R, b R, and sums of products are fixed once per block, not per term.
__device__ __forceinline__ sc mm_(const sc &a, const sc &b) { sc r; sc_montmul(r, a, b); return r; }      // a b / R
// x^e R from a table of x^(2^k) R, k < 32
__device__ __forceinline__ sc pow_tbl_mont(const sc *tblR, uint32_t e) {
    sc acc = sc_r();
    for (int k = 0; e; k++, e >>= 1)
        if (e & 1) acc = mm_(acc, ld_sc(tblR + k));
    return acc;
}
// A thread of the strip kernels below owns elements i0, i0 + 256, .. (STRIP of them): one table walk for the first
// power, one multiplication by x^256 (table entry 8) per further element, and the loads stay coalesced.
constexpr int STRIP = 4;

// x^e from a table of x^(2^k), k < 32
__device__ __forceinline__ sc pow_tbl(const sc *tbl, uint32_t e) {
    sc acc = sc_one();
    for (int k = 0; e; k++, e >>= 1)
        if (e & 1) acc = mul_(acc, ld_sc(tbl + k));
    return acc;
}

// block-wide sum of NS scalars per thread (256 threads); thread 0 writes out[block*NS + s]
template <int NS>
__device__ void block_sum(sc (&v)[NS], sc *out) {
    __shared__ uint32_t sm[NS][8][256];
    const int t = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) sm[s][k][t] = v[s].v[k];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if (t < stride) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                sc a, b;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    a.v[k] = sm[s][k][t];
                    b.v[k] = sm[s][k][t + stride];
                }
                sc_add(a, a, b);
#pragma unroll
                for (int k = 0; k < 8; k++) sm[s][k][t] = a.v[k];
            }
        }
        __syncthreads();
    }
    if (t == 0) {
#pragma unroll
        for (int s = 0; s < NS; s++) {
            sc a;
#pragma unroll
            for (int k = 0; k < 8; k++) a.v[k] = sm[s][k][0];
            st_sc(out + (size_t)blockIdx.x * NS + s, a);
        }
    }
}

// =================================================================== kernels

// zpowR[q] = z^(q+1) R (Montgomery form: the flattening multiplies it by plain coefficients).  ztblR[k] = z^(2^k) R.
__global__ void __launch_bounds__(256) k_zpow(const sc *__restrict__ ztblR, sc *__restrict__ zpowR, uint32_t Q) {
    uint32_t q = blockIdx.x * (256u * STRIP) + threadIdx.x;
    if (q >= Q) return;
    sc p = pow_tbl_mont(ztblR, q + 1);
    const sc step = ld_sc(ztblR + 8);
    st_sc(zpowR + q, p);
#pragma unroll
    for (int k = 1; k < STRIP; k++) {
        q += 256u;
        if (q >= Q) return;
        p = mm_(p, step);
        st_sc(zpowR + q, p);
    }
}

// Flattened constraints: W[slot] = (+/-) sum over the slot's terms of z^(q+1) * coeff.
// Slots: [0,n) wL, [n,2n) wR, [2n,3n) wO, [3n,3n+m) wV (negated), 3n+m: wc (negated).
__global__ void __launch_bounds__(256) k_flatten(const uint32_t *__restrict__ slot_off, const uint32_t *__restrict__ t_coeff,
                                                 const uint32_t *__restrict__ t_q, const sc *__restrict__ coeffs,
                                                 const sc *__restrict__ zpowR, sc *__restrict__ W, uint32_t nslots,
                                                 uint32_t neg_from) {
    uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nslots) return;
    sc acc = sc_zero();
    for (uint32_t t = slot_off[s]; t < slot_off[s + 1]; t++) acc = add_(acc, mm_(ld_sc(zpowR + t_q[t]), ld_sc(coeffs + t_coeff[t])));
    if (s >= neg_from) sc_neg(acc, acc);
    st_sc(W + s, acc);
}

// per-block partial sums of  sum_t zpow[q_t] * coeffs[c_t]  (the constant terms -> wc)
__global__ void __launch_bounds__(256) k_dot_zpow(const uint32_t *__restrict__ t_coeff, const uint32_t *__restrict__ t_q,
                                                  const sc *__restrict__ coeffs, const sc *__restrict__ zpowR, uint32_t nterms,
                                                  sc *__restrict__ partial) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    sc v[1] = {sc_zero()};
    if (t < nterms) v[0] = mm_(ld_sc(zpowR + t_q[t]), ld_sc(coeffs + t_coeff[t]));
    block_sum<1>(v, partial);
}

// Provers that share their launches (finish_group): the latency-bound kernels around the accumulate pass then run once
// per group, not once per proof.  The blinding RNG advances eight states per permutation (two draws per group of 16).
constexpr size_t GROUP_MAX = 16;

// l(X), r(X) coefficient vectors and the per-block partial sums of t1..t6
//   l1 = a_L + y^-i wR   l2 = a_O   l3 = s_L      r0 = wO - y^i   r1 = y^i a_R + wL   r3 = y^i s_R
// (k_poly and k_eval, like the round kernels further down, take a group of up to 8 proofs: blockIdx.y = proof)
struct poly_proof {
    const sc *aL, *aR, *aO, *sL, *sR, *W, *ytbl, *yinvtbl;
    sc *l1, *r0, *r1, *r3, *partial;
};
struct poly_group {
    poly_proof p[GROUP_MAX];
};
__global__ void __launch_bounds__(256) k_poly(uint32_t n, const poly_group G) {
    const poly_proof &P = G.p[blockIdx.y];
    const sc *__restrict__ aL = P.aL, *__restrict__ aR = P.aR, *__restrict__ aO = P.aO, *__restrict__ sL = P.sL,
                          *__restrict__ sR = P.sR, *__restrict__ W = P.W, *__restrict__ ytbl = P.ytbl, *__restrict__ yinvtbl = P.yinvtbl;
    sc *__restrict__ l1 = P.l1, *__restrict__ r0 = P.r0, *__restrict__ r1 = P.r1, *__restrict__ r3 = P.r3, *__restrict__ partial = P.partial;
    uint32_t i = blockIdx.x * (256u * STRIP) + threadIdx.x;
    sc t[6];                                   // sums of products a b / R: brought back to plain once, below
#pragma unroll
    for (int k = 0; k < 6; k++) t[k] = sc_zero();
    if (i < n) {
        sc yiR = pow_tbl_mont(ytbl, i), yinviR = pow_tbl_mont(yinvtbl, i);        // ytbl / yinvtbl hold y^(+-2^k) R
        const sc ystep = ld_sc(ytbl + 8), yinvstep = ld_sc(yinvtbl + 8), one = sc_one();
#pragma unroll
        for (int k = 0; k < STRIP; k++) {
            if (k) {
                i += 256u;
                if (i >= n) break;
                yiR = mm_(yiR, ystep);
                yinviR = mm_(yinviR, yinvstep);
            }
            sc wL = ld_sc(W + i), wR = ld_sc(W + n + i), wO = ld_sc(W + 2 * (size_t)n + i);
            sc vl1 = add_(ld_sc(aL + i), mm_(wR, yinviR));
            sc vl2 = ld_sc(aO + i), vl3 = ld_sc(sL + i);
            sc vr0 = sub_(wO, mm_(yiR, one));
            sc vr1 = add_(mm_(ld_sc(aR + i), yiR), wL);
            sc vr3 = mm_(ld_sc(sR + i), yiR);
            st_sc(l1 + i, vl1);
            st_sc(r0 + i, vr0);
            st_sc(r1 + i, vr1);
            st_sc(r3 + i, vr3);
            t[0] = add_(t[0], mm_(vl1, vr0));
            t[1] = add_(t[1], add_(mm_(vl1, vr1), mm_(vl2, vr0)));
            t[2] = add_(t[2], add_(mm_(vl2, vr1), mm_(vl3, vr0)));
            t[3] = add_(t[3], add_(mm_(vl1, vr3), mm_(vl3, vr1)));
            t[4] = add_(t[4], mm_(vl2, vr3));
            t[5] = add_(t[5], mm_(vl3, vr3));
        }
        const sc rr = sc_rr();
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = mm_(t[k], rr);
    }
    block_sum<6>(t, partial);
}

// l(x), r(x) -> the IPP's a, b (length np, padded), and the generator factor vectors
//   sG_j = G_factors[j] = 1 (j < n) | u (pad)      sH_j = y^-j * G_factors[j]
struct eval_proof {
    uint32_t n1;             // multipliers of the first phase: G_factors = 1 below n1, u from n1 on (second phase and padding)
    sc xR, xRR, x3RR, u;     // x R, x R^2, x^3 R^2 (Montgomery-form constants, see mm_), u plain
    const sc *l1, *aO, *sL, *r0, *r1, *r3, *ytbl, *yinvtbl;
    sc *a, *b, *sG, *sH;     // a, b are written as a R, b R (the round kernels keep them so); sG, sH plain
};
struct eval_group {
    eval_proof p[GROUP_MAX];
};
__global__ void __launch_bounds__(256) k_eval(uint32_t n, uint32_t np, const eval_group G) {
    const eval_proof &P = G.p[blockIdx.y];
    const sc xR = P.xR, xRR = P.xRR, x3RR = P.x3RR, u = P.u;
    const sc *__restrict__ l1 = P.l1, *__restrict__ aO = P.aO, *__restrict__ sL = P.sL, *__restrict__ r0 = P.r0, *__restrict__ r1 = P.r1,
                          *__restrict__ r3 = P.r3, *__restrict__ ytbl = P.ytbl, *__restrict__ yinvtbl = P.yinvtbl;
    sc *__restrict__ a = P.a, *__restrict__ b = P.b, *__restrict__ sG = P.sG, *__restrict__ sH = P.sH;
    uint32_t i = blockIdx.x * (256u * STRIP) + threadIdx.x;
    if (i >= np) return;
    sc yinviR = pow_tbl_mont(yinvtbl, i);
    const sc yinvstep = ld_sc(yinvtbl + 8), one = sc_one(), rr = sc_rr();
#pragma unroll
    for (int k = 0; k < STRIP; k++) {
        if (k) {
            i += 256u;
            if (i >= np) return;
            yinviR = mm_(yinviR, yinvstep);
        }
        const bool first = i < P.n1;
        st_sc(sG + i, first ? one : u);
        st_sc(sH + i, mm_(yinviR, first ? one : u));              // y^-i (u): plain
        if (i < n) {
            // l = x (l1 + x (l2 + x l3));  r = r0 + x r1 + x^3 r3; both leave as (.) R
            sc l = mm_(add_(ld_sc(l1 + i), mm_(add_(ld_sc(aO + i), mm_(ld_sc(sL + i), xR)), xR)), xRR);
            sc r = add_(mm_(ld_sc(r0 + i), rr), add_(mm_(ld_sc(r1 + i), xRR), mm_(ld_sc(r3 + i), x3RR)));
            st_sc(a + i, l);
            st_sc(b + i, r);
        } else {
            sc negR;
            sc_neg(negR, pow_tbl_mont(ytbl, i));                  // padding: a = 0, b = -y^i
            st_sc(a + i, sc_zero());
            st_sc(b + i, negR);
        }
    }
}

// One IPP round, scalar side.  n_k = current vector length, half = n_k / 2.
// rows: rowL / rowR have 2*np + 1 scalars: [ G_0..G_{np-1} | H_0..H_{np-1} | B ].
//   G_j (idx = j mod n_k): idx >= half -> L gets a[idx-half] * sG_j, else R gets a[idx+half] * sG_j
//   H_j:                   idx <  half -> L gets b[idx+half] * sH_j, else R gets b[idx-half] * sH_j
// partial[block] = per-block sums of (c_L, c_R) terms, c_L = <a_lo, b_hi>, c_R = <a_hi, b_lo>.
// The three round kernels serve a group of up to 8 proofs per launch (blockIdx.y = proof): alone they are
// launch-bound (17 us for 2048 elements), and a group's rounds are in lock-step anyway.
struct ipp_proof {
    sc *a, *b, *sG, *sH;       // the proof's vectors: a, b in Montgomery form (a R, b R), sG, sH plain
    sc *rowL, *rowR;           // its block of the L / R scalar rows (plain: the MSM reads them)
    sc *cpart;                 // per-block partial sums of (c_L R, c_R R)
    sc *outL, *outR;           // where the rows keep the scalar of B
    sc w, uR, uinvR;           // Q = w B (plain); the round's challenge and its inverse, times R
};
struct ipp_group {
    ipp_proof p[GROUP_MAX];
};
__global__ void __launch_bounds__(256) k_ipp_rows(uint32_t np, uint32_t nk, const ipp_group G) {
    const ipp_proof &P = G.p[blockIdx.y];
    const sc *__restrict__ a = P.a, *__restrict__ b = P.b, *__restrict__ sG = P.sG, *__restrict__ sH = P.sH;
    sc *__restrict__ rowL = P.rowL, *__restrict__ rowR = P.rowR, *__restrict__ partial = P.cpart;
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    sc c[2] = {sc_zero(), sc_zero()};
    if (j < np) {
        const uint32_t half = nk >> 1, idx = j & (nk - 1);
        const bool hi = idx >= half;
        const uint32_t other = hi ? idx - half : idx + half;
        sc ga = mm_(ld_sc(a + other), ld_sc(sG + j));             // (a R) sG / R = a sG
        sc hb = mm_(ld_sc(b + other), ld_sc(sH + j));
        const sc zero = sc_zero();
        st_sc(rowL + j, hi ? ga : zero);
        st_sc(rowR + j, hi ? zero : ga);
        st_sc(rowL + np + j, hi ? zero : hb);
        st_sc(rowR + np + j, hi ? hb : zero);
        if (j < half) {
            c[0] = mm_(ld_sc(a + j), ld_sc(b + half + j));        // (a R)(b R) / R = a b R
            c[1] = mm_(ld_sc(a + half + j), ld_sc(b + j));
        }
    }
    block_sum<2>(c, partial);
}

// finish c_L, c_R and write the B-scalars  c_L * w,  c_R * w  (Q = w B)
__global__ void __launch_bounds__(256) k_ipp_c(uint32_t nblocks, const ipp_group G) {
    const ipp_proof &P = G.p[blockIdx.y];
    const sc *__restrict__ partial = P.cpart;
    sc *__restrict__ outL = P.outL, *__restrict__ outR = P.outR;
    const sc w = P.w;
    __shared__ sc res[2];
    sc c[2] = {sc_zero(), sc_zero()};
    for (uint32_t k = threadIdx.x; k < nblocks; k += 256) {
        c[0] = add_(c[0], ld_sc(partial + 2 * (size_t)k));
        c[1] = add_(c[1], ld_sc(partial + 2 * (size_t)k + 1));
    }
    block_sum<2>(c, res);
    __syncthreads();
    if (threadIdx.x == 0) {
        st_sc(outL, mm_(res[0], w));                              // (c R) w / R = c w
        st_sc(outR, mm_(res[1], w));
    }
}

// fold: a' = u a_lo + u^-1 a_hi, b' = u^-1 b_lo + u b_hi; sG_j *= (hi ? u : u^-1), sH_j *= (hi ? u^-1 : u)
__global__ void __launch_bounds__(256) k_ipp_fold(uint32_t np, uint32_t nk, const ipp_group G) {
    const ipp_proof &P = G.p[blockIdx.y];
    sc *__restrict__ a = P.a, *__restrict__ b = P.b, *__restrict__ sG = P.sG, *__restrict__ sH = P.sH;
    const sc uR = P.uR, uinvR = P.uinvR;
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= np) return;
    const uint32_t half = nk >> 1, idx = j & (nk - 1);
    const bool hi = idx >= half;
    st_sc(sG + j, mm_(ld_sc(sG + j), hi ? uR : uinvR));
    st_sc(sH + j, mm_(ld_sc(sH + j), hi ? uinvR : uR));
    if (j < half) {
        sc na = add_(mm_(ld_sc(a + j), uR), mm_(ld_sc(a + half + j), uinvR));
        sc nb = add_(mm_(ld_sc(b + j), uinvR), mm_(ld_sc(b + half + j), uR));
        st_sc(a + j, na);
        st_sc(b + j, nb);
    }
}

// Verifier: g / h scalars of the combined check, and the per-block sums of delta = <y^-i wR_i, wL_i>
//   s_i = prod_k (bit_{lg-1-k}(i) ? u_k : u_k^-1)
//   g_i = ufor_i (x y^-i wR_i - a s_i)      h_i = ufor_i (y^-i (x wL_i + wO_i - b s_{np-1-i}) - 1)
struct verify_args {
    sc x, a, b, u;
    uint32_t n, np, lg, n1;     // n1: multipliers of the first phase (u multiplies the scalars from n1 on)
};
__global__ void __launch_bounds__(256) k_verify_scalars(verify_args A, const sc *__restrict__ W, const sc *__restrict__ yinvtbl,
                                                        const sc *__restrict__ uk, const sc *__restrict__ ukinv,
                                                        sc *__restrict__ g, sc *__restrict__ h, sc *__restrict__ partial) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    sc d[1] = {sc_zero()};
    if (i < A.np) {
        sc s = sc_one(), srev = sc_one();
        const uint32_t irev = A.np - 1 - i;
        for (uint32_t k = 0; k < A.lg; k++) {
            const uint32_t bit = A.lg - 1 - k;
            sc f = ld_sc(uk + k), fi = ld_sc(ukinv + k);
            s = mul_(s, ((i >> bit) & 1) ? f : fi);
            srev = mul_(srev, ((irev >> bit) & 1) ? f : fi);
        }
        sc yinvi = pow_tbl(yinvtbl, i);
        sc wL = sc_zero(), wR = sc_zero(), wO = sc_zero();
        if (i < A.n) {
            wL = ld_sc(W + i);
            wR = ld_sc(W + A.n + i);
            wO = ld_sc(W + 2 * (size_t)A.n + i);
        }
        sc yneg_wR = mul_(wR, yinvi);
        d[0] = mul_(yneg_wR, wL);
        sc gi = sub_(mul_(A.x, yneg_wR), mul_(A.a, s));
        sc hi = sub_(mul_(yinvi, sub_(add_(mul_(A.x, wL), wO), mul_(A.b, srev))), sc_one());
        if (i >= A.n1) {
            gi = mul_(gi, A.u);
            hi = mul_(hi, A.u);
        }
        st_sc(g + i, gi);
        st_sc(h + i, hi);
    }
    block_sum<1>(d, partial);
}

// the generator factors after the fold: sG = sH = 1 over the fm folded generators of each proof (blockIdx.y = 2 j + side)
__global__ void __launch_bounds__(256) k_fill_ones(sc *__restrict__ s_all, uint32_t np, uint32_t fm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < fm) st_sc(s_all + (size_t)blockIdx.y * np + i, sc_one());
}

// batch verification: acc[i] += rho * row[i]
__global__ void __launch_bounds__(256) k_axpy(uint32_t n, sc rho, const sc *__restrict__ row, sc *__restrict__ acc) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    sc r = ld_sc(row + i), a = ld_sc(acc + i), t;
    sc_mul(t, rho, r);
    sc_add(a, a, t);
    st_sc(acc + i, a);
}

// Scalar::random's second half: 64 bytes of RNG output -> a scalar mod l (dalek: Scalar::from_bytes_mod_order_wide).
// raw[i] = 16 little-endian words; out[i] = raw[i] mod l.
__global__ void __launch_bounds__(256) k_reduce512(const uint4 *__restrict__ raw, sc *__restrict__ out, uint32_t count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint4 a = raw[4 * (size_t)i], b = raw[4 * (size_t)i + 1], c = raw[4 * (size_t)i + 2], d = raw[4 * (size_t)i + 3];
    sc lo, hi, r;
    lo.v[0] = a.x; lo.v[1] = a.y; lo.v[2] = a.z; lo.v[3] = a.w; lo.v[4] = b.x; lo.v[5] = b.y; lo.v[6] = b.z; lo.v[7] = b.w;
    hi.v[0] = c.x; hi.v[1] = c.y; hi.v[2] = c.z; hi.v[3] = c.w; hi.v[4] = d.x; hi.v[5] = d.y; hi.v[6] = d.z; hi.v[7] = d.w;
    sc_reduce512(r, lo, hi);
    st_sc(out + i, r);
}

// K3: fixed-base Pedersen commitments v B + r B~ with radix-16 comb tables, compressed on the device.
// comb[(base*64 + w)*8 + (k-1)] = k * 16^w * base   (affine Niels).  16 lanes per commitment: lane l
// adds the windows w = l mod 16 of both scalars (8 mixed additions), four butterfly steps sum the
// lanes.  (One lane per commitment walked 128 dependent additions: 0.64 ms for any batch size.)
// The compression -- one inverse square root, ~290 dependent field operations, three quarters of
// the kernel -- then runs with one lane per commitment in the block's first wave only: the sums go
// through LDS, the other three waves retire.  (Round 2 compressed in lane 0 of every 16: every wave
// of the launch stayed resident for the whole chain with 4 of 64 lanes working.)
__global__ void __launch_bounds__(256) k_pedersen(const aniels_rec *__restrict__ comb, const sc *__restrict__ scalars /* [m][2] */,
                                                  uint32_t m, uint8_t *__restrict__ out32) {
    __shared__ uint32_t sums[16][33];              // 16 commitments per block: X | Y | Z | T (row padded: no bank conflicts)
    const uint32_t lane_id = blockIdx.x * 256u + threadIdx.x;
    const uint32_t i = lane_id >> 4, sub = lane_id & 15u;
    ge acc;
    ge_identity(acc);
    if (i < m) {                                   // uniform over the 16 lanes of a commitment
        for (int base = 0; base < 2; base++) {
            sc s = ld_sc(scalars + 2 * (size_t)i + base);
            // signed radix-16 digits without a serial carry: s + 0x88..8 (s < 2^253: no overflow), digit = nibble - 8
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) s.v[k] = addc32(s.v[k], 0x88888888u, c, c);
#pragma unroll
            for (int jw = 0; jw < 4; jw++) {
                const int w = (int)sub + 16 * jw;
                const int d = (int)((s.v[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
                if (d != 0) {
                    const int k = d < 0 ? -d : d;
                    const uint32_t *rec = comb[(size_t)(base * 64 + w) * 8 + (k - 1)].w;
                    ge_aniels q;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        q.hYpX.v[t] = rec[t];
                        q.hYmX.v[t] = rec[8 + t];
                        q.Tdxy.v[t] = rec[16 + t];
                    }
                    ge_madd(acc, acc, q, d < 0);
                }
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        ge other;
        ge_shfl_xor(other, acc, o, 16);
        ge_add(acc, acc, other);
    }
    if (sub == 0) {
        uint32_t *row = sums[threadIdx.x >> 4];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            row[t] = acc.X.v[t];
            row[8 + t] = acc.Y.v[t];
            row[16 + t] = acc.Z.v[t];
            row[24 + t] = acc.T.v[t];
        }
    }
    __syncthreads();
    const uint32_t ci = blockIdx.x * 16u + threadIdx.x;
    if (threadIdx.x >= 16 || ci >= m) return;
    const uint32_t *row = sums[threadIdx.x];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        acc.X.v[t] = row[t];
        acc.Y.v[t] = row[8 + t];
        acc.Z.v[t] = row[16 + t];
        acc.T.v[t] = row[24 + t];
    }
    uint32_t enc[8];
    ristretto_encode_w(enc, acc);
    uint4 *o = reinterpret_cast<uint4 *>(out32 + 32 * (size_t)ci);
    o[0] = make_uint4(enc[0], enc[1], enc[2], enc[3]);
    o[1] = make_uint4(enc[4], enc[5], enc[6], enc[7]);
}

// comb table construction: thread (base, w, k) computes k * 16^w * base by double-and-add
__global__ void __launch_bounds__(64) k_build_comb(const aniels_rec *__restrict__ bases /* B, B~ */, aniels_rec *__restrict__ comb) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= 2 * 64 * 8) return;
    const int base = i / 512, w = (i / 8) % 64, k = (i % 8) + 1;
    const uint32_t *rec = bases[base].w;
    ge_aniels q;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        q.hYpX.v[t] = rec[t];
        q.hYmX.v[t] = rec[8 + t];
        q.Tdxy.v[t] = rec[16 + t];
    }
    ge acc;
    ge_identity(acc);
    for (int r = 0; r < k; r++) ge_madd(acc, acc, q, false);      // k * base
    for (int r = 0; r < 4 * w; r++) ge_double(acc, acc);          // * 16^w
    fe zinv;
    fe_invert(zinv, acc.Z);
    ge_aniels o;
    ge_to_aniels(o, acc, zinv);
    uint32_t *dst = comb[i].w;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        dst[t] = o.hYpX.v[t];
        dst[8 + t] = o.hYmX.v[t];
        dst[16 + t] = o.Tdxy.v[t];
    }
    fe nt;
    fe_neg(nt, o.Tdxy);
#pragma unroll
    for (int t = 0; t < 8; t++) dst[24 + t] = nt.v[t];
}

// =================================================================== host: store / sonny / proof bytes

uint32_t ConstraintStore::intern(const sc &c) {
    static const sc one = sc_one();
    static const sc minus_one = (-Scalar(1)).v;
    if (sc_eq(c, one)) return intern_small(1);
    if (sc_eq(c, minus_one)) return intern_small(-1);
    uint64_t h = ((uint64_t)c.v[1] << 32 | c.v[0]) ^ ((uint64_t)c.v[7] << 32 | c.v[4]) * 0x9e3779b97f4a7c15ULL;
    auto &ids = coeff_lookup[h];
    for (uint32_t id : ids)
        if (sc_eq(coeffs[id], c)) return id;
    coeffs.push_back(c);
    small_of_id.push_back(LinearCombination::small_of(c));
    ids.push_back((uint32_t)coeffs.size() - 1);
    return (uint32_t)coeffs.size() - 1;
}

uint32_t ConstraintStore::intern_small(int32_t k) {
    uint32_t *slot = k == 1 ? &id_one : k == -1 ? &id_minus_one : nullptr;
    if (!slot) return intern(LinearCombination::scalar_of(k));
    if (*slot == 0xffffffffu) {
        coeffs.push_back(LinearCombination::scalar_of(k));
        small_of_id.push_back(k);
        *slot = (uint32_t)coeffs.size() - 1;
    }
    return *slot;
}

static inline bool var_in_range(uint32_t packed, size_t n_mult, size_t n_commit) {
    const uint32_t kind = packed >> 28, index = packed & 0x0fffffffu;
    const size_t lim = kind == VAR_COMMITTED ? n_commit : kind == VAR_ONE ? 1 : kind <= VAR_OUTPUT ? n_mult : 0;
    return index < lim;
}

void ConstraintStore::push(const LinearCombination &lc, size_t n_mult, size_t n_commit, const Variable *minus_var) {
    const LinearCombination::Entry *e = lc.entries();
    const size_t n = lc.size();
    for (size_t i = 0; i < n; i++)
        if (!var_in_range(e[i].var, n_mult, n_commit)) throw R1CSError(BPG_EINVAL, "variable does not belong to this constraint system");
    if (follow_key) {
        if (push_follow(lc, minus_var)) {
            num_constraints++;
            return;
        }
        materialize();
    }
    const sc *big = lc.bigs();
    for (size_t i = 0; i < n; i++) {
        term_var.push_back(e[i].var);
        term_coeff.push_back(e[i].k != LinearCombination::BIG ? intern_small(e[i].k) : intern(*big++));
        term_q.push_back(num_constraints);
    }
    if (minus_var) {
        term_var.push_back(minus_var->kind << 28 | minus_var->index);
        term_coeff.push_back(intern_small(-1));
        term_q.push_back(num_constraints);
    }
    num_constraints++;
}

// One constraint against the followed circuit: same number of terms (the followed constraint neither ends earlier
// nor goes on), same variables, same coefficient values.  The comparison is exact -- a coefficient id of the
// followed circuit below this store's dictionary size names a value this store has seen and is compared with it; an
// id AT the size says the followed circuit met a new value here, so the value must be new here too (intern() then
// gives it exactly that id, as plain capture would).  Nothing is recorded: on a mismatch the caller copies the
// prefix out and records this constraint itself, and a coefficient interned on the way is the one plain capture
// would have interned at the same point.
bool ConstraintStore::push_follow(const LinearCombination &lc, const Variable *minus_var) {
    const CircuitKey &k = *follow_key;
    const size_t nk = k.term_var.size(), len = lc.size() + (minus_var ? 1 : 0), end = cur + len;
    const uint32_t q = num_constraints;
    if (end > nk) return false;
    if (len && (k.term_q[cur] != q || k.term_q[end - 1] != q)) return false;
    if (end < nk && k.term_q[end] == q) return false;
    const uint32_t *kv = k.term_var.data() + cur, *kc = k.term_coeff.data() + cur;
    const LinearCombination::Entry *e = lc.entries();
    const sc *big = lc.bigs();
    const size_t n = lc.size();
    size_t i = 0;
    for (; i < n; i++) {
        if (kv[i] != e[i].var) return false;
        const uint32_t id = kc[i];
        if (e[i].k != LinearCombination::BIG) {
            if (id < coeffs.size()) {
                if (small_of_id[id] != e[i].k) return false;
            } else if (id != coeffs.size() || intern_small(e[i].k) != id) {
                return false;
            }
        } else {
            const sc &c = *big++;
            if (id < coeffs.size()) {
                if (!sc_eq(coeffs[id], c)) return false;
            } else if (id != coeffs.size() || intern(c) != id) {
                return false;
            }
        }
    }
    if (minus_var) {
        if (kv[i] != (minus_var->kind << 28 | minus_var->index)) return false;
        const uint32_t id = kc[i];
        if (id != id_minus_one && (id != coeffs.size() || intern_small(-1) != id)) return false;
    }
    cur = end;
    return true;
}

void ConstraintStore::follow(std::shared_ptr<const CircuitKey> key, std::shared_ptr<void> entry) {
    if (!key || num_constraints || !term_var.empty() || !coeffs.empty()) return;      // only a store that has captured nothing
    follow_key = std::move(key);
    follow_entry = std::move(entry);
    cur = 0;
}

void ConstraintStore::materialize() {
    if (!follow_key) return;
    const CircuitKey &k = *follow_key;
    term_var.assign(k.term_var.begin(), k.term_var.begin() + cur);
    term_coeff.assign(k.term_coeff.begin(), k.term_coeff.begin() + cur);
    term_q.assign(k.term_q.begin(), k.term_q.begin() + cur);
    follow_key.reset();
    follow_entry.reset();
    cur = 0;
}

std::shared_ptr<const CircuitKey> ConstraintStore::snapshot(size_t n_mult) const {
    auto k = std::make_shared<CircuitKey>();
    if (follow_key) {
        k->term_var.assign(follow_key->term_var.begin(), follow_key->term_var.begin() + cur);
        k->term_coeff.assign(follow_key->term_coeff.begin(), follow_key->term_coeff.begin() + cur);
        k->term_q.assign(follow_key->term_q.begin(), follow_key->term_q.begin() + cur);
    } else {
        k->term_var = term_var;
        k->term_coeff = term_coeff;
        k->term_q = term_q;
    }
    k->num_constraints = num_constraints;
    k->num_coeffs = (uint32_t)coeffs.size();
    k->n_mult = (uint32_t)n_mult;
    return k;
}

Scalar sonny_a() { return -Scalar(1); }
Scalar sonny_d() {
    static const Scalar d = -(Scalar(86650) * Scalar(86649).invert());
    return d;
}
SonnyPoint sonny_add(const SonnyPoint &p, const SonnyPoint &q) {
    Scalar A = p.X * q.X, B = p.Y * q.Y, C = sonny_d() * p.T * q.T, D = p.Z * q.Z;
    Scalar E = (p.X + p.Y) * (q.X + q.Y) - A - B, F = D - C, G = D + C, H = B + A;
    return {E * F, G * H, F * G, E * H};
}

std::vector<uint8_t> R1CSProof::to_bytes() const {
    static const uint8_t zero[32] = {0};
    bool one_phase = !memcmp(A_I2, zero, 32) && !memcmp(A_O2, zero, 32) && !memcmp(S2, zero, 32);
    std::vector<uint8_t> out;
    auto put = [&](const uint8_t *p) { out.insert(out.end(), p, p + 32); };
    auto puts = [&](const Scalar &s) {
        uint8_t b[32];
        s.to_bytes(b);
        put(b);
    };
    out.push_back(one_phase ? 0 : 1);
    put(A_I1); put(A_O1); put(S1);
    if (!one_phase) { put(A_I2); put(A_O2); put(S2); }
    put(T_1); put(T_3); put(T_4); put(T_5); put(T_6);
    puts(t_x); puts(t_x_blinding); puts(e_blinding);
    for (size_t k = 0; k < L_vec.size() / 32; k++) {
        put(&L_vec[32 * k]);
        put(&R_vec[32 * k]);
    }
    puts(a); puts(b);
    return out;
}

bool R1CSProof::from_bytes(const uint8_t *d, size_t len, R1CSProof &o) {
    if (len < 1 || d[0] > 1) return false;
    const size_t k = d[0] == 0 ? 11 : 14;
    const uint8_t *p = d + 1;
    size_t body = len - 1;
    if (body % 32 || body / 32 < k + 2 || ((body / 32 - k - 2) & 1)) return false;
    size_t nch = body / 32, i = 0;
    auto take = [&](uint8_t *dst) { memcpy(dst, p + 32 * i++, 32); };
    auto takes = [&](Scalar &s) {
        sc t;
        sc_frombytes(t, p + 32 * i++);
        if (!sc_is_canonical(t)) return false;
        s = Scalar(t);
        return true;
    };
    take(o.A_I1); take(o.A_O1); take(o.S1);
    if (k == 14) { take(o.A_I2); take(o.A_O2); take(o.S2); }
    else { memset(o.A_I2, 0, 32); memset(o.A_O2, 0, 32); memset(o.S2, 0, 32); }
    take(o.T_1); take(o.T_3); take(o.T_4); take(o.T_5); take(o.T_6);
    if (!takes(o.t_x) || !takes(o.t_x_blinding) || !takes(o.e_blinding)) return false;
    size_t lg = (nch - i - 2) / 2;
    if (lg >= 32) return false;
    o.L_vec.resize(32 * lg);
    o.R_vec.resize(32 * lg);
    for (size_t r = 0; r < lg; r++) {
        memcpy(&o.L_vec[32 * r], p + 32 * i++, 32);
        memcpy(&o.R_vec[32 * r], p + 32 * i++, 32);
    }
    return takes(o.a) && takes(o.b);
}

// =================================================================== host: device arena + helpers

struct Arena {
    bpg_lane *lane;
    size_t used = 0;
    explicit Arena(bpg_lane *l) : lane(l) {}
    int reserve(size_t bytes) {
        if (bytes <= lane->ws_cap) return BPG_OK;
        if (lane->ws) (void)hipFree(lane->ws);
        lane->ws = nullptr;
        lane->ws_cap = 0;
        size_t want = bytes + bytes / 4 + (1 << 20);
        BPG_CHECK_HIP(device_malloc(&lane->ws, want));
        lane->ws_cap = want;
        return BPG_OK;
    }
    template <typename T>
    T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T *p = (T *)((uint8_t *)lane->ws + used);
        used += bytes;
        return p;
    }
};
static size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// Pinned staging for the small transfers of one prove / verify call (lane->h_io): a bump allocator that the call sizes up
// front.  h2d copies the source into the staging, d2h reserves staging for the result; neither enqueues anything: the
// pending items are moved by ONE kernel per flush() (k_small_copies: the device reads and writes the pinned host memory
// directly).  A copy command per item -- six per proof and phase -- went through a copy engine and back to the compute
// queue each time, and the copy engines' queues are FIFOs shared by every prover thread of the process, where a 32-byte
// table waits behind another thread's 8 MB of witness: with the results of the MSMs written to pinned memory by their
// last kernel (msm.hip) and these batched, proofs/s at 2^16 multipliers went from 446-449 to 481-494 and beyond.
// flush() before the kernel that reads an uploaded item; flush() + wait + collect() delivers the downloads.  Where the
// staging cannot be had (allocation failure, a call that outgrows its estimate) and for items above 64 KB the
// transfer is a plain hipMemcpyAsync on the pageable / pinned pointer.
struct small_copy {
    void *dst;
    const void *src;
    uint32_t bytes, pad;
};
__global__ void __launch_bounds__(256) k_small_copies(const small_copy *__restrict__ items) {
    const small_copy c = items[blockIdx.x];
    if ((((uintptr_t)c.dst | (uintptr_t)c.src | (uintptr_t)c.bytes) & 15u) == 0) {
        const uint4 *src = (const uint4 *)c.src;
        uint4 *dst = (uint4 *)c.dst;
        for (uint32_t i = threadIdx.x; i < c.bytes / 16; i += 256) dst[i] = src[i];
    } else {
        const uint32_t *src = (const uint32_t *)c.src;
        uint32_t *dst = (uint32_t *)c.dst;
        for (uint32_t i = threadIdx.x; i < c.bytes / 4; i += 256) dst[i] = src[i];
    }
}

struct PinArena {
    bpg_lane *lane;
    size_t used = 0;
    struct Pending {
        void *dst;
        const void *src;
        size_t bytes;
    };
    std::vector<Pending> pending;          // downloads: staging -> where the host wants them (collect)
    std::vector<small_copy> batch;         // device-side moves of the next flush
    static constexpr size_t SMALL = (size_t)1 << 16;
    explicit PinArena(bpg_lane *l) : lane(l) {}
    void reserve(size_t bytes) { (void)grow_pinned(&lane->h_io, &lane->h_io_cap, bytes); }
    void *take(size_t bytes) {
        const size_t b = (bytes + 63) & ~(size_t)63;
        if (!lane->h_io || used + b > lane->h_io_cap) return nullptr;
        void *p = (uint8_t *)lane->h_io + used;
        used += b;
        return p;
    }
    hipError_t h2d(void *d_dst, const void *h_src, size_t bytes, hipStream_t s) {
        void *st = take(bytes);
        if (st) memcpy(st, h_src, bytes);
        if (st && bytes <= SMALL && bytes % 4 == 0) {
            batch.push_back({d_dst, st, (uint32_t)bytes, 0});
            return hipSuccess;
        }
        return hipMemcpyAsync(d_dst, st ? st : h_src, bytes, hipMemcpyHostToDevice, s);
    }
    hipError_t d2h(void *h_dst, const void *d_src, size_t bytes, hipStream_t s) {
        void *st = take(bytes);
        if (st) pending.push_back({h_dst, st, bytes});
        if (st && bytes <= SMALL && bytes % 4 == 0) {
            batch.push_back({st, d_src, (uint32_t)bytes, 0});
            return hipSuccess;
        }
        return hipMemcpyAsync(st ? st : h_dst, d_src, bytes, hipMemcpyDeviceToHost, s);
    }
    // enqueue the batched moves (before the kernel that reads an upload; before the wait that precedes collect)
    hipError_t flush(hipStream_t s) {
        if (batch.empty()) return hipSuccess;
        const size_t n = batch.size();
        small_copy *items = (small_copy *)take(n * sizeof(small_copy));
        if (!items) {                         // no room for the list: item by item after all
            for (const small_copy &c : batch) {
                hipError_t e = hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDefault, s);
                if (e != hipSuccess) return e;
            }
            batch.clear();
            return hipSuccess;
        }
        memcpy(items, batch.data(), n * sizeof(small_copy));
        batch.clear();
        hipLaunchKernelGGL(k_small_copies, dim3((unsigned)n), dim3(256), 0, s, (const small_copy *)items);
        return hipGetLastError();
    }
    void collect() {                       // after flush() and a wait on the stream
        for (auto &p : pending) memcpy(p.dst, p.src, p.bytes);
        pending.clear();
    }
};

static void pow2_table(sc out[32], const sc &x) {
    out[0] = x;
    for (int k = 1; k < 32; k++) sc_mul(out[k], out[k - 1], out[k - 1]);
}

static unsigned grid256(size_t n) { return (unsigned)((n + 255) / 256); }
static unsigned grid_strip(size_t n) { return (unsigned)((n + 256 * STRIP - 1) / (256 * STRIP)); }      // the strip kernels: STRIP elements per thread
// x^(2^k) R, k < 32: the form the strip kernels take their power tables in (pow_tbl_mont)
static void pow2_table_mont(sc out[32], const sc &x) {
    pow2_table(out, x);
    for (int k = 0; k < 32; k++) sc_to_mont(out[k], out[k]);
}
static sc to_mont(const Scalar &x) {
    sc r;
    sc_to_mont(r, x.v);
    return r;
}

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// slots by variable: counting sort of the constraint terms (host, O(terms)).
// Constant (Variable::One) terms are numerous (every public coordinate is one) and all
// land in the single scalar wc, so they are kept apart and summed by a parallel
// dot-product kernel instead of by one thread.
struct FlatPlan {
    std::vector<uint32_t> slot_off, t_coeff, t_q, one_coeff, one_q;
    uint32_t nslots = 0, neg_from = 0;
    size_t nterms = 0, nones = 0;          // t_coeff.size(), one_coeff.size() (kept when the vectors are dropped)
};
static void build_flat_plan(const ConstraintStore &cs, uint32_t n, uint32_t m, bool want_wc, FlatPlan &fp) {
    fp.nslots = 3 * n + m;
    fp.neg_from = 3 * n;
    fp.slot_off.assign(fp.nslots + 1, 0);
    const size_t nt = cs.term_var.size();
    auto slot_of = [&](uint32_t tv) -> int64_t {
        uint32_t kind = tv >> 28, idx = tv & 0x0fffffffu;
        switch (kind) {
            case VAR_LEFT: return idx;
            case VAR_RIGHT: return (int64_t)n + idx;
            case VAR_OUTPUT: return 2 * (int64_t)n + idx;
            case VAR_COMMITTED: return 3 * (int64_t)n + idx;
            default: return -1;
        }
    };
    for (size_t t = 0; t < nt; t++) {
        int64_t s = slot_of(cs.term_var[t]);
        if (s >= 0) fp.slot_off[s + 1]++;
    }
    for (uint32_t s = 0; s < fp.nslots; s++) fp.slot_off[s + 1] += fp.slot_off[s];
    std::vector<uint32_t> cur(fp.slot_off.begin(), fp.slot_off.end() - 1);
    fp.t_coeff.resize(fp.slot_off[fp.nslots]);
    fp.t_q.resize(fp.slot_off[fp.nslots]);
    for (size_t t = 0; t < nt; t++) {
        int64_t s = slot_of(cs.term_var[t]);
        if (s < 0) {
            if (want_wc) {
                fp.one_coeff.push_back(cs.term_coeff[t]);
                fp.one_q.push_back(cs.term_q[t]);
            }
            continue;
        }
        uint32_t pos = cur[s]++;
        fp.t_coeff[pos] = cs.term_coeff[t];
        fp.t_q[pos] = cs.term_q[t];
    }
    fp.nterms = fp.t_coeff.size();
    fp.nones = fp.one_coeff.size();
}

// ---- circuit-structure cache ("cache the circuit structure across proofs of the same gadget").
// The plan depends on the captured term list only -- which variable, which interned coefficient id, which
// constraint -- not on the coefficient values or the witness, so every proof and every verification of one
// circuit shares it: the reference's *_roundtrip_helper functions re-capture the same circuit per call
// (ref: tests/edwards_point.rs:84-95).  A hit skips the counting sort and the upload of the term lists (12 bytes
// per term); the key is compared in full, the hash only finds the candidate.
struct FlatCached {
    uint32_t n = 0, m = 0;
    bool want_wc = false;
    uint64_t hash = 0;
    uint64_t ctx_id = 0;                                    // the context whose device holds the arrays below
    std::shared_ptr<const CircuitKey> key;                  // the captured term lists (also what later captures follow)
    FlatPlan fp;                                            // sizes only (the vectors live in HBM)
    uint32_t *d_off = nullptr, *d_tc = nullptr, *d_tq = nullptr, *d_oc = nullptr, *d_oq = nullptr;
    void *d_block = nullptr;
    ~FlatCached() {
        if (d_block) (void)hipFree(d_block);
    }
};
constexpr size_t FLAT_CACHE_ENTRIES = 8;

// four independent multiply-xorshift lanes over 64-bit words (the term lists of a 2^20-multiplier circuit are tens of MB)
static uint64_t hash_words(const std::vector<uint32_t> &v, uint64_t h) {
    const uint64_t K = 0x9E3779B97F4A7C15ull;
    uint64_t a[4] = {h, h ^ K, h + K, ~h};
    const size_t n8 = v.size() / 8;
    const uint32_t *p = v.data();
    for (size_t i = 0; i < n8; i++, p += 8)
        for (int l = 0; l < 4; l++) {
            uint64_t w;
            memcpy(&w, p + 2 * l, 8);
            a[l] = (a[l] ^ w) * K;
            a[l] ^= a[l] >> 29;
        }
    for (size_t i = n8 * 8; i < v.size(); i++) {
        a[0] = (a[0] ^ v[i]) * K;
        a[0] ^= a[0] >> 29;
    }
    uint64_t r = v.size();
    for (int l = 0; l < 4; l++) {
        r = (r ^ a[l]) * K;
        r ^= r >> 32;
    }
    return r;
}

// The circuit each side flattened last in this process: what the next Prover (role 0) / Verifier (role 1) follows
// while it captures (ConstraintStore::follow).  A wrong guess costs one early mismatch.
// (Weak references: the entries' device memory belongs to their context's cache and goes at bpg_shutdown; a strong
// reference here would free it from a static destructor, after the HIP runtime has gone.)
static std::mutex g_last_mu;
static std::weak_ptr<FlatCached> g_last_shape[2];

static void remember_shape(const std::shared_ptr<FlatCached> &e) {
    std::lock_guard<std::mutex> lk(g_last_mu);
    g_last_shape[e->want_wc ? 1 : 0] = e;
}
static std::shared_ptr<FlatCached> last_shape(int role) {
    std::lock_guard<std::mutex> lk(g_last_mu);
    return g_last_shape[role].lock();
}
void forget_shapes_of(uint64_t ctx_id) {                    // bpg_shutdown: the entries' device memory goes with the context
    std::lock_guard<std::mutex> lk(g_last_mu);
    for (auto &w : g_last_shape)
        if (auto e = w.lock())
            if (e->ctx_id == ctx_id) w.reset();
}

static int flat_plan_cached(bpg_ctx *ctx, bpg_lane *lane, ConstraintStore &cs, uint32_t n, uint32_t m, bool want_wc,
                            std::shared_ptr<FlatCached> &out) {
    // a capture that followed a cached circuit to its end IS that circuit (every term was compared on the way)
    if (cs.followed_fully()) {
        auto e = std::static_pointer_cast<FlatCached>(cs.follow_entry.lock());
        if (e && e->ctx_id == ctx->id && e->n == n && e->m == m && e->want_wc == want_wc) {
            ctx->flat_hits++;
            ctx->flat_followed++;
            out = e;
            return BPG_OK;
        }
    }
    cs.materialize();
    uint64_t h = 0x243F6A8885A308D3ull ^ ((uint64_t)n << 32) ^ m ^ (want_wc ? 1ull << 63 : 0);
    h = hash_words(cs.term_var, h);
    h = hash_words(cs.term_coeff, h);
    h = hash_words(cs.term_q, h);
    std::vector<std::shared_ptr<void>> candidates;
    {
        std::lock_guard<std::mutex> lk(ctx->flat_mu);
        candidates = ctx->flat_cache;
    }
    for (auto &c : candidates) {
        auto e = std::static_pointer_cast<FlatCached>(c);
        if (e->hash == h && e->n == n && e->m == m && e->want_wc == want_wc && e->key->num_constraints == cs.num_constraints &&
            e->key->term_var == cs.term_var && e->key->term_coeff == cs.term_coeff && e->key->term_q == cs.term_q) {
            {
                std::lock_guard<std::mutex> lk(ctx->flat_mu);              // most recently used last
                auto it = std::find(ctx->flat_cache.begin(), ctx->flat_cache.end(), c);
                if (it != ctx->flat_cache.end()) std::rotate(it, it + 1, ctx->flat_cache.end());
            }
            ctx->flat_hits++;
            remember_shape(e);
            out = e;
            return BPG_OK;
        }
    }
    auto e = std::make_shared<FlatCached>();
    e->n = n;
    e->m = m;
    e->want_wc = want_wc;
    e->hash = h;
    e->ctx_id = ctx->id;
    e->key = cs.snapshot(n);
    build_flat_plan(cs, n, m, want_wc, e->fp);
    FlatPlan &fp = e->fp;
    const size_t w_off = fp.nslots + 1, w_t = fp.nterms + 1, w_o = fp.nones + 1;
    const size_t words = (w_off + 63) / 64 * 64 + 2 * ((w_t + 63) / 64 * 64) + 2 * ((w_o + 63) / 64 * 64);
    BPG_CHECK_HIP(device_malloc(&e->d_block, words * 4));
    uint32_t *p = (uint32_t *)e->d_block;
    e->d_off = p;  p += (w_off + 63) / 64 * 64;
    e->d_tc = p;   p += (w_t + 63) / 64 * 64;
    e->d_tq = p;   p += (w_t + 63) / 64 * 64;
    e->d_oc = p;   p += (w_o + 63) / 64 * 64;
    e->d_oq = p;
    hipStream_t s = lane->stream;
    BPG_CHECK_HIP(hipMemcpyAsync(e->d_off, fp.slot_off.data(), w_off * 4, hipMemcpyHostToDevice, s));
    if (fp.nterms) {
        BPG_CHECK_HIP(hipMemcpyAsync(e->d_tc, fp.t_coeff.data(), fp.nterms * 4, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(e->d_tq, fp.t_q.data(), fp.nterms * 4, hipMemcpyHostToDevice, s));
    }
    if (fp.nones) {
        BPG_CHECK_HIP(hipMemcpyAsync(e->d_oc, fp.one_coeff.data(), fp.nones * 4, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(e->d_oq, fp.one_q.data(), fp.nones * 4, hipMemcpyHostToDevice, s));
    }
    BPG_CHECK_HIP(lane_sync(lane));                 // complete before any other thread's stream can read it
    for (auto *v : {&fp.slot_off, &fp.t_coeff, &fp.t_q, &fp.one_coeff, &fp.one_q}) std::vector<uint32_t>().swap(*v);
    {
        std::lock_guard<std::mutex> lk(ctx->flat_mu);
        if (ctx->flat_cache.size() >= FLAT_CACHE_ENTRIES) ctx->flat_cache.erase(ctx->flat_cache.begin());   // freed with its last user
        ctx->flat_cache.push_back(e);
        ctx->flat_misses++;
    }
    remember_shape(e);
    out = e;
    return BPG_OK;
}

// runs k_zpow + k_flatten (+ k_dot_zpow for the constant terms); W (device) receives
// nslots scalars; *d_one / *n_one receive the per-block partial sums of sum z^(q+1) c over
// the constant terms (the caller adds them up and negates to get wc).
static int gpu_flatten(bpg_lane *lane, Arena &ar, PinArena &pin, const ConstraintStore &cs, const FlatCached &fc, const sc &z, sc *d_W,
                       sc **d_one = nullptr, size_t *n_one = nullptr, hipEvent_t *ev = nullptr /* 4: zpow, flatten start / stop */) {
    hipStream_t s = lane->stream;
    const FlatPlan &fp = fc.fp;
    const uint32_t Q = cs.num_constraints;
    sc ztbl[32];
    pow2_table_mont(ztbl, z);
    sc *d_ztbl = ar.take<sc>(32), *d_zpow = ar.take<sc>(Q ? Q : 1), *d_coeffs = ar.take<sc>(cs.coeffs.size() ? cs.coeffs.size() : 1);
    BPG_CHECK_HIP(pin.h2d(d_ztbl, ztbl, sizeof ztbl, s));
    if (!cs.coeffs.empty()) BPG_CHECK_HIP(pin.h2d(d_coeffs, cs.coeffs.data(), cs.coeffs.size() * sizeof(sc), s));
    BPG_CHECK_HIP(pin.flush(s));
    if (Q) hipExtLaunchKernelGGL(k_zpow, dim3(grid_strip(Q)), dim3(256), 0, s, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, 0, d_ztbl, d_zpow, Q);
    if (fp.nslots)
        hipExtLaunchKernelGGL(k_flatten, dim3(grid256(fp.nslots)), dim3(256), 0, s, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr, 0, fc.d_off,
                              fc.d_tc, fc.d_tq, d_coeffs, d_zpow, d_W, fp.nslots, fp.neg_from);
    if (d_one) {
        const size_t no = fp.nones, nb = grid256(no ? no : 1);
        *n_one = no ? nb : 0;
        *d_one = ar.take<sc>(nb);
        if (no) hipLaunchKernelGGL(k_dot_zpow, dim3((unsigned)nb), dim3(256), 0, s, fc.d_oc, fc.d_oq, d_coeffs, d_zpow, (uint32_t)no, *d_one);
    }
    BPG_CHECK_HIP(hipGetLastError());
    return BPG_OK;
}

static size_t flatten_bytes(const ConstraintStore &cs, const FlatPlan &fp) {
    return pad256(32 * 32) + pad256((cs.num_constraints + 1) * 32) + pad256((cs.coeffs.size() + 1) * 32) +
           pad256((fp.nones / 256 + 2) * 32);
}

// =================================================================== generators

int gens_create(bpg_ctx *ctx, size_t capacity, Gens **out, int rank, int world, exchange_fn exchange, void *exchange_user) {
    if (!ctx || !out || capacity == 0 || capacity > ((size_t)1 << 26)) return BPG_EINVAL;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !exchange)) return BPG_EINVAL;
    bpg_lane *lane = lane_of(ctx);
    if (!lane) return BPG_EDEVICE;
    // everything allocated below is released by gens_destroy on every failure path
    struct Guard {
        Gens *g;
        ~Guard() { gens_destroy(g); }
    } guard{new Gens()};
    Gens *g = guard.g;
    g->ctx = ctx;
    g->capacity = capacity;
    g->rank = rank;
    g->world = world;
    g->exchange = exchange;
    g->exchange_user = exchange_user;
    g->lo = capacity * (size_t)rank / (size_t)world;
    g->hi = capacity * (size_t)(rank + 1) / (size_t)world;
    bpg_points *tbl = new bpg_points();
    tbl->ctx = ctx;
    tbl->n = g->npoints();
    tbl->d_table = nullptr;
    g->table = tbl;
    g->fb_c = fixed_base_window(tbl->n);
    g->fb_windows = fixed_base_windows(g->fb_c);
    g->tab_c = g->fb_c;
    if (world == 1 && capacity >= 4096 && (g->fb_c == 12 || g->fb_c == 14) && !getenv("BPG_NO_HALF_WINDOWS")) g->tab_c = g->fb_c / 2;
    g->tab_blocks = fixed_base_windows(g->tab_c);
    hipError_t e = device_malloc((void **)&tbl->d_table, (size_t)g->tab_blocks * tbl->n * sizeof(aniels_rec));
    if (e != hipSuccess) {
        set_last_error("hipMalloc(generator table)", e);
        return BPG_ENOMEM;
    }
    // BulletproofGens::new(capacity, 1): GeneratorsChain = SHAKE256("GeneratorsChain" | label), label = 'G'/'H' | u32le(0).
    // The XOF stream is sequential: a rank squeezes past the generators it does not hold.
    const size_t CH = (size_t)1 << 16;
    std::vector<uint8_t> stream(64 * (capacity < CH ? capacity : CH));
    int rc;
    for (int which = 0; which < 2; which++) {
        KeccakSponge x = shake256();
        const uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), 0, 0, 0, 0};
        x.absorb("GeneratorsChain", 15);
        x.absorb(label, 5);
        for (size_t c0 = 0; c0 < g->hi; c0 += CH) {
            const size_t c1 = c0 + CH < capacity ? c0 + CH : capacity;
            x.squeeze(stream.data(), 64 * (c1 - c0));
            const size_t a = c0 > g->lo ? c0 : g->lo, b = c1 < g->hi ? c1 : g->hi;
            if (a >= b) continue;
            aniels_rec *dst = tbl->d_table + (which ? g->off_H() : g->off_G()) + (a - g->lo);
            if ((rc = table_write_uniform(ctx, dst, stream.data() + 64 * (a - c0), b - a))) return rc;
        }
    }
    // PedersenGens::default(): B = ristretto basepoint, B_blinding = from_uniform_bytes(SHA3-512(compress(B)))
    ge Bp;
    ge_basepoint(Bp);
    uint8_t Benc[32], h[64];
    ristretto_encode(Benc, Bp);
    uint32_t bad = 0;
    if ((rc = table_write_compressed(ctx, tbl->d_table + g->off_B(), Benc, 1, &bad)) || bad) return rc ? rc : BPG_EDEVICE;
    KeccakSponge s3 = sha3_512();
    s3.absorb(Benc, 32);
    s3.squeeze(h, 64);
    if ((rc = table_write_uniform(ctx, tbl->d_table + g->off_Bb(), h, 1))) return rc;
    if ((rc = table_build_shifted(ctx, tbl->d_table, tbl->n, g->tab_c, g->tab_blocks))) return rc;
    BPG_CHECK_HIP(device_malloc(&g->d_comb, 2 * 64 * 8 * sizeof(aniels_rec)));
    hipLaunchKernelGGL(k_build_comb, dim3(16), dim3(64), 0, lane->stream, tbl->d_table + g->off_B(), (aniels_rec *)g->d_comb);
    BPG_CHECK_HIP(hipGetLastError());
    BPG_CHECK_HIP(lane_sync(lane));
    guard.g = nullptr;
    *out = g;
    return BPG_OK;
}

void gens_destroy(Gens *g) {
    if (!g) return;
    if (g->d_comb) (void)hipFree(g->d_comb);
    if (g->table) {
        (void)hipFree(g->table->d_table);
        delete g->table;
    }
    delete g;
}

// m Pedersen commitments (K3); scalars = [m][2] host; out32 host.  Staging lives in the lane (grown on demand):
// no hipMalloc / hipFree per call -- hipFree synchronises the whole device under every other prover thread.
static int pedersen_batch(Gens *g, const sc *vr, size_t m, uint8_t *out32) {
    if (!m) return BPG_OK;
    bpg_lane *lane = lane_of(g->ctx);
    if (!lane) return BPG_EDEVICE;
    hipStream_t s = lane->stream;
    int rc;
    if ((rc = grow(&lane->d_io, &lane->io_cap, m * 96))) return rc;
    sc *d_s = (sc *)lane->d_io;
    uint8_t *d_o = (uint8_t *)lane->d_io + m * 64;
    // Both directions through pinned staging of the lane's own, moved by the device: a copy kernel brings the scalars in
    // (16 lanes read each pair: not over the bus), and k_pedersen writes the 32-byte results straight to the host
    // buffer.  No copy-engine command on the stream (see PinArena).  Without the staging: plain copies.
    uint8_t *h_in = nullptr, *h_out = nullptr;
    if (grow_pinned(&lane->h_ped, &lane->h_ped_cap, m * 96) == BPG_OK) {
        h_in = (uint8_t *)lane->h_ped;
        h_out = h_in + m * 64;
        memcpy(h_in, vr, m * 64);
    }
    if (h_in) {
        const unsigned nb = (unsigned)((m * 4 + 255) / 256);
        hipLaunchKernelGGL(k_copy16, dim3(nb < 64 ? nb : 64), dim3(256), 0, s, (uint4 *)d_s, (const uint4 *)h_in, (uint32_t)(m * 4));
    } else {
        BPG_CHECK_HIP(hipMemcpyAsync(d_s, vr, m * 64, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_pedersen, dim3((unsigned)((m * 16 + 255) / 256)), dim3(256), 0, s, (const aniels_rec *)g->d_comb, d_s, (uint32_t)m,
                       h_out ? h_out : d_o);
    BPG_CHECK_HIP(hipGetLastError());
    if (!h_out) BPG_CHECK_HIP(hipMemcpyAsync(out32, d_o, m * 32, hipMemcpyDeviceToHost, s));
    BPG_CHECK_HIP(lane_sync(lane));
    if (h_out) memcpy(out32, h_out, m * 32);
    return BPG_OK;
}

// =================================================================== MSMs over the (possibly sharded) generator table

// generators [a, b) of G (which = 0) or H (which = 1) that are resident on this rank, as an MSM
// segment; the scalar of generator i sits sc_base + (i - a) scalars into a target's row
static msm_seg gen_seg(const Gens *g, int which, size_t a, size_t b, size_t sc_base) {
    const size_t x = a > g->lo ? a : g->lo, y = b < g->hi ? b : g->hi;
    const aniels_rec *tbl = g->table->d_table + (which ? g->off_H() : g->off_G());
    if (x >= y) return msm_seg{tbl, 0, sc_base};
    return msm_seg{tbl + (x - g->lo), y - x, sc_base + (x - a)};
}
// `count` records starting at B (first = 0) or B_blinding (first = 1): rank 0 alone adds these terms
static msm_seg pedersen_seg(const Gens *g, int first, size_t count, size_t sc_off) {
    return msm_seg{g->table->d_table + g->off_B() + first, g->rank == 0 ? count : 0, sc_off};
}

// all-gather this rank's partial sums (T x 128 B) and add them per target; every rank gets the
// same T compressed points.  Unsharded: compress the single partial.
static int finish_partials(Gens *g, const uint8_t *mine, int T, uint8_t *out32) {
    if (g->world == 1) {
        for (int t = 0; t < T; t++) {
            int rc = combine_partials(mine + 128 * t, 1, out32 + 32 * t);
            if (rc) return rc;
        }
        return BPG_OK;
    }
    const size_t bytes = (size_t)T * 128;
    std::vector<uint8_t> all(bytes * g->world), one(128 * (size_t)g->world);
    if (g->exchange(g->exchange_user, mine, bytes, all.data())) {
        set_last_error_msg("partial-sum exchange failed");
        return BPG_EDEVICE;
    }
    for (int t = 0; t < T; t++) {
        for (int r = 0; r < g->world; r++) memcpy(&one[128 * r], &all[(size_t)r * bytes + 128 * t], 128);
        int rc = combine_partials(one.data(), g->world, out32 + 32 * t);
        if (rc) return rc;
    }
    return BPG_OK;
}

// One fixed-base MSM with T targets over this rank's part of the generator table.  Rows of
// `stride` scalars each; segment k reads its scalars at segs[k].sc_off.
static int gens_msm(Gens *g, const msm_seg *segs, int nseg, const sc *d_rows, size_t stride, int T, uint8_t *out32,
                    float density = 1.f, HostPhase::Id ph_submit = HostPhase::AS_SUBMIT, HostPhase::Id ph_wait = HostPhase::AS_WAIT) {
    size_t n = 0;
    for (int k = 0; k < nseg; k++) n += segs[k].len;
    bpg_msm_job *job = nullptr;
    int rc;
    {
        HostPhase hp(ph_submit);
        rc = msm_submit(g->ctx, segs, nseg, d_rows, n, T, false, &job, msm_fixed_base{g->fb_c, g->fb_stride(), density}, stride);
    }
    if (rc) return rc;
    HostPhase hp(ph_wait);
    if (g->world == 1) return msm_wait(job, out32, false);
    std::vector<uint8_t> mine((size_t)T * 128);
    if ((rc = msm_wait(job, mine.data(), true))) return rc;
    return finish_partials(g, mine.data(), T, out32);
}

// =================================================================== Prover

Prover::Prover(Gens *gens, const void *label, size_t label_len) : gens_(gens), tr_(label, label_len) {
    tr_.r1cs_domain_sep();
    if (auto e = last_shape(0)) follow(e->key, e);
}

Variable Prover::commit(const Scalar &v, const Scalar &v_blinding, uint8_t V32[32]) {
    Variable var;
    commit_batch(&v, &v_blinding, 1, V32, &var);
    return var;
}

void Prover::commit_batch(const Scalar *v, const Scalar *v_blinding, size_t m, uint8_t *V32, Variable *vars) {
    std::vector<sc> vr(2 * m);
    for (size_t i = 0; i < m; i++) {
        vr[2 * i] = v[i].v;
        vr[2 * i + 1] = v_blinding[i].v;
    }
    cancel_blindings();                  // a commitment changes the transcript the early draws started from
    HostPhase hp(HostPhase::CAPTURE_COMMIT);
    int rc = pedersen_batch(gens_, vr.data(), m, V32);
    if (rc) throw R1CSError(rc, "pedersen commitment kernel failed");
    for (size_t i = 0; i < m; i++) {
        vars[i] = Variable{VAR_COMMITTED, (uint32_t)v_.size()};
        v_.push_back(v[i].v);
        v_blinding_.push_back(v_blinding[i].v);
        tr_.append_point("V", V32 + 32 * i);
    }
}

Scalar Prover::eval(const LinearCombination &lc) const {
    const LinearCombination::Entry *e = lc.entries();
    const sc *big = lc.bigs();
    const size_t n = lc.size(), nm = a_L_.size(), nc = v_.size();
    sc acc = sc_zero();
    for (size_t i = 0; i < n; i++) {
        const uint32_t kind = e[i].var >> 28, index = e[i].var & 0x0fffffffu;
        const int32_t k = e[i].k;
        const sc *val;
        switch (kind) {
            case VAR_LEFT: val = index < nm ? &a_L_[index] : nullptr; break;
            case VAR_RIGHT: val = index < nm ? &a_R_[index] : nullptr; break;
            case VAR_OUTPUT: val = index < nm ? &a_O_[index] : nullptr; break;
            case VAR_COMMITTED: val = index < nc ? &v_[index] : nullptr; break;
            case VAR_ONE: {                              // a constant: the coefficient itself
                if (index != 0) throw R1CSError(BPG_EINVAL, "variable does not belong to this constraint system");
                if (k == LinearCombination::BIG) sc_add(acc, acc, *big++);
                else sc_add(acc, acc, LinearCombination::scalar_of(k));
                continue;
            }
            default: val = nullptr;
        }
        if (!val) throw R1CSError(BPG_EINVAL, "variable does not belong to this constraint system");
        // almost every coefficient in the gadget circuits is +1 or -1: no multiplication then
        if (k == 1) {
            sc_add(acc, acc, *val);
        } else if (k == -1) {
            sc_sub(acc, acc, *val);
        } else {
            sc t;
            sc_mul(t, k == LinearCombination::BIG ? *big++ : LinearCombination::scalar_of(k), *val);
            sc_add(acc, acc, t);
        }
    }
    return Scalar(acc);
}

ConstraintSystem::Triple Prover::multiply(const LinearCombination &left, const LinearCombination &right) {
    Scalar l = eval(left), r = eval(right), o = l * r;
    uint32_t i = (uint32_t)a_L_.size();
    Triple t{{VAR_LEFT, i}, {VAR_RIGHT, i}, {VAR_OUTPUT, i}};
    a_L_.push_back(l.v);
    a_R_.push_back(r.v);
    a_O_.push_back(o.v);
    cs_.push(left, a_L_.size(), v_.size(), &t.l);          // left - a_L[i] = 0
    cs_.push(right, a_L_.size(), v_.size(), &t.r);         // right - a_R[i] = 0
    publish_multipliers();
    return t;
}

ConstraintSystem::Triple Prover::allocate_multiplier(const Scalar *left, const Scalar *right) {
    if (!left || !right) throw R1CSError(BPG_EMISSING, "R1CSError::MissingAssignment");
    uint32_t i = (uint32_t)a_L_.size();
    a_L_.push_back(left->v);
    a_R_.push_back(right->v);
    a_O_.push_back(((*left) * (*right)).v);
    publish_multipliers();
    return Triple{{VAR_LEFT, i}, {VAR_RIGHT, i}, {VAR_OUTPUT, i}};
}

static inline size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

int Prover::set_gens(Gens *g) {
    if (!g || g->ctx != gens_->ctx) return BPG_EINVAL;
    gens_ = g;
    return BPG_OK;
}

Scalar Prover::challenge_scalar(const std::string &label) {
    if (!phase2_) throw R1CSError(BPG_EINVAL, "challenge_scalar outside a randomized-constraints callback");
    return Scalar(tr_.challenge_scalar(label.c_str()));
}

// A_I, A_O, S over the multipliers [lo, hi): <a_L, G[lo..hi)> + <a_R, H[lo..hi)> + bl[0] B~, <a_O, G> + bl[1] B~,
// <s_L, G> + <s_R, H> + bl[2] B~ -- one 3-target MSM.  (Two-phase proofs only; the usual path commits inside finish_group.)
int Prover::commit3(size_t lo, size_t hi, const sc bl[3], const sc *sL, const sc *sR, uint8_t out96[96]) {
    bpg_lane *lane = lane_of(gens_->ctx);
    if (!lane) return BPG_EDEVICE;
    hipStream_t s = lane->stream;
    const size_t k = hi - lo, N1 = 2 * k + 1;
    Arena ar(lane);
    int rc = ar.reserve(3 * N1 * 32 + 4096);
    if (rc) return rc;
    sc *rows = ar.take<sc>(3 * N1);
    BPG_CHECK_HIP(hipMemsetAsync(rows, 0, 3 * N1 * 32, s));
    for (int t = 0; t < 3; t++) BPG_CHECK_HIP(hipMemcpyAsync(rows + t * N1, &bl[t], 32, hipMemcpyHostToDevice, s));
    if (k) {
        BPG_CHECK_HIP(hipMemcpyAsync(rows + 1, a_L_.data() + lo, k * 32, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(rows + 1 + k, a_R_.data() + lo, k * 32, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(rows + N1 + 1, a_O_.data() + lo, k * 32, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(rows + 2 * N1 + 1, sL, k * 32, hipMemcpyHostToDevice, s));
        BPG_CHECK_HIP(hipMemcpyAsync(rows + 2 * N1 + 1 + k, sR, k * 32, hipMemcpyHostToDevice, s));
    }
    BPG_CHECK_HIP(lane_sync(lane));
    msm_seg segs[3] = {pedersen_seg(gens_, 1, 1, 0), gen_seg(gens_, 0, lo, hi, 1), gen_seg(gens_, 1, lo, hi, 1 + k)};
    return gens_msm(gens_, segs, 3, rows, N1, 3, out96);
}

// Randomized constraints (bulletproofs r1cs, two-phase): commit to the first-phase wires, let the deferred callbacks
// draw their challenges and build the second phase, commit to its wires.  Leaves the transcript where the one-phase
// path has it before the challenges y, z; s_L_ / s_R_ then cover both phases.
int Prover::two_phase_prologue(R1CSProof &P) {
    const size_t n1 = a_L_.size();
    n1_ = n1;
    uint8_t o[96];
    const sc bl1[3] = {i_bl_, o_bl_, s_bl_};
    int rc = commit3(0, n1, bl1, s_L_.data(), s_R_.data(), o);
    if (rc) return rc;
    memcpy(P.A_I1, o, 32);
    memcpy(P.A_O1, o + 32, 32);
    memcpy(P.S1, o + 64, 32);
    tr_.append_point("A_I1", P.A_I1);
    tr_.append_point("A_O1", P.A_O1);
    tr_.append_point("S1", P.S1);
    tr_.r1cs_2phase_domain_sep();
    std::vector<RandomizedFn> cbs;
    cbs.swap(deferred_);
    consumed_ = true;
    phase2_ = true;
    try {
        for (auto &cb : cbs) cb(*this);
    } catch (...) {
        phase2_ = false;
        throw;
    }
    phase2_ = false;
    const size_t n = a_L_.size(), n2 = n - n1;
    if (gens_->capacity < next_pow2(n ? n : 1)) return BPG_EGENS;
    if (n2) {
        i_bl2_ = rng_->scalar();
        o_bl2_ = rng_->scalar();
        s_bl2_ = rng_->scalar();
        s_L_.resize(n);
        s_R_.resize(n);
        for (size_t i = n1; i < n; i++) s_L_[i] = rng_->scalar();
        for (size_t i = n1; i < n; i++) s_R_[i] = rng_->scalar();
        const sc bl2[3] = {i_bl2_, o_bl2_, s_bl2_};
        if ((rc = commit3(n1, n, bl2, s_L_.data() + n1, s_R_.data() + n1, o))) return rc;
        memcpy(P.A_I2, o, 32);
        memcpy(P.A_O2, o + 32, 32);
        memcpy(P.S2, o + 64, 32);
    } else {
        i_bl2_ = o_bl2_ = s_bl2_ = sc_zero();
        memset(P.A_I2, 0, 32);
        memset(P.A_O2, 0, 32);
        memset(P.S2, 0, 32);
    }
    tr_.append_point("A_I2", P.A_I2);
    tr_.append_point("A_O2", P.A_O2);
    tr_.append_point("S2", P.S2);
    two_phase_done_ = true;
    return BPG_OK;
}

// Inner-product round after which finish_group materializes the folded generators (0: never; BPG_IPP_FOLD
// overrides): when 2 x 2048 folded generators are left.  With the table's half windows the fold costs about 2.2
// rounds' worth of additions (3.4 with three-piece digits) whenever it is done.  Measured at 2^16 multipliers, 64
// threads x 8 statements: 421 proofs/s with the three-piece fold after round 5; with the half-window fold 452-455
// after round 5 and 443-462 after round 4 (one fold-free round less, but twice the folded table, bucket sets and
// later rounds: a wash, and the smaller tables win on memory).
static int ipp_fold_default(size_t lg) { return lg >= 12 ? (int)lg - 11 : 0; }

int Prover::prove(rng_fill_fn rng_fn, void *rng_user, R1CSProof &P) {
    int rc = prove_begin(rng_fn, rng_user);
    if (rc) return rc;
    draw_blindings();
    return prove_finish(P);
}

// ---- early blinding draws (one large proof; see BlindingSpec)
static void spec_release(BlindingSpec &sp) {
    if (!sp.raw) return;
    std::lock_guard<std::mutex> lk(sp.ctx->spec_mu);
    sp.ctx->spec_pool.emplace_back(sp.raw, sp.slots);
    sp.raw = nullptr;
}

static void spec_run(BlindingSpec *sp, int device, std::vector<sc> v_blinding, const uint8_t *ext32) {
    const double t0 = now_ms();
    for (const sc &vb : v_blinding) {             // (one permutation each: 8 ms for the 38 836 commitments of a 2^20-multiplier proof)
        uint8_t b[32];
        sc_tobytes(b, vb);
        sp->rng->rekey_with_witness_bytes("v_blinding", b, 32);
    }
    sp->rng->finalize(ext32);
    {   // a pinned buffer of the context's pool that is large enough, else a new one (this thread pays for the pinning)
        std::lock_guard<std::mutex> lk(sp->ctx->spec_mu);
        auto &pool = sp->ctx->spec_pool;
        for (size_t i = 0; i < pool.size(); i++)
            if (pool[i].second >= sp->slots) {
                sp->raw = (uint8_t *)pool[i].first;
                sp->slots = pool[i].second;
                pool.erase(pool.begin() + i);
                break;
            }
    }
    if (!sp->raw) {
        void *p = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&p, sp->slots * 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            sp->failed = true;
            sp->heads.store(true, std::memory_order_release);
            return;
        }
        sp->raw = (uint8_t *)p;
    }
    sp->i_bl = sp->rng->scalar();
    sp->o_bl = sp->rng->scalar();
    sp->s_bl = sp->rng->scalar();
    sp->heads.store(true, std::memory_order_release);
    size_t k = 0;
    while (!sp->cancel.load(std::memory_order_acquire)) {
        const bool fin = sp->final_.load(std::memory_order_acquire);       // read before the limit: final => the limit is 2 n
        size_t lim = sp->limit.load(std::memory_order_acquire);
        if (lim > sp->slots) lim = sp->slots;
        if (k < lim) {
            const size_t stop = lim - k > 512 ? k + 512 : lim;
            for (; k < stop; k++) sp->rng->fill_bytes(sp->raw + 64 * k, 64);
            sp->done.store(k, std::memory_order_release);
        } else if (fin) {
            break;
        } else {                                   // ahead of the capture (it allows ~12 M draws/s, the chain makes 5 M/s): rare
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
    }
    sp->busy_ms = now_ms() - t0;
}

int Prover::begin_blindings(const uint8_t ext32[32]) {
    if (!ext32) return BPG_EINVAL;
    if (ext_seeded_ || consumed_ || rng_) {
        set_last_error_msg("begin_blindings: called twice, or after prove()");
        return BPG_EINVAL;
    }
    memcpy(ext_, ext32, 32);
    ext_seeded_ = true;
    if (!deferred_.empty() || !gens_) return BPG_OK;          // two-phase proofs draw in two rounds: no early start
    std::unique_ptr<BlindingSpec> sp(new BlindingSpec());
    sp->ctx = gens_->ctx;
    sp->slots = 2 * gens_->capacity;
    Transcript t = tr_;                                        // what prove_begin does, on a copy
    t.append_u64("m", v_.size());
    sp->rng.reset(new TranscriptRng(t.build_rng()));
    sp->limit.store(2 * a_L_.size(), std::memory_order_release);
    BlindingSpec *raw = sp.get();
    const int device = gens_->ctx->device;
    try {
        sp->th = std::thread(spec_run, raw, device, v_blinding_, (const uint8_t *)ext_);     // ext_ outlives the helper (joined by ~Prover)
    } catch (...) {
        return BPG_OK;                                         // no helper: prove() draws as usual
    }
    spec_ = std::move(sp);
    return BPG_OK;
}

void Prover::cancel_blindings() {
    if (!spec_) return;
    spec_->cancel.store(true, std::memory_order_release);
    if (spec_->th.joinable()) spec_->th.join();
    spec_release(*spec_);
    spec_.reset();
}

// ---- transcript RNG (host, sequential Keccak)
int Prover::prove_begin(rng_fill_fn rng_fn, void *rng_user) {
    HostPhase hp(HostPhase::RNG_BEGIN);
    t_start_ = now_ms();
    s_raw_pin_ = nullptr;
    const size_t n = a_L_.size(), m = v_.size();
    if (consumed_) {
        set_last_error_msg("prove: this prover's randomized-constraint callbacks have already run (a prover proves once)");
        return BPG_EINVAL;
    }
    if (gens_->capacity < next_pow2(n ? n : 1)) return BPG_EGENS;
    tr_.append_u64("m", m);
    if (spec_) {
        // the helper started from this transcript (a later commit would have stopped it): tell it where the draws end
        if (!deferred_.empty() || 2 * n > spec_->slots || !raw_draws()) {
            cancel_blindings();
        } else {
            spec_->limit.store(2 * n, std::memory_order_release);
            spec_->final_.store(true, std::memory_order_release);
            return BPG_OK;
        }
    }
    rng_.reset(new TranscriptRng(tr_.build_rng()));
    for (size_t i = 0; i < m; i++) {
        uint8_t b[32];
        sc_tobytes(b, v_blinding_[i]);
        rng_->rekey_with_witness_bytes("v_blinding", b, 32);
    }
    uint8_t ext[32];
    if (ext_seeded_) memcpy(ext, ext_, 32);
    else rng_fn(rng_user, ext, 32);
    rng_->finalize(ext);
    return BPG_OK;
}

// a prover alone: its raw draws take the whole of the lane's pinned staging
void Prover::draw_blindings() {
    if (spec_) {                          // the helper has them (or will): finish_group streams its draws to the device
        while (!spec_->heads.load(std::memory_order_acquire)) {
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
        if (!spec_->failed) {
            i_bl_ = spec_->i_bl;
            o_bl_ = spec_->o_bl;
            s_bl_ = spec_->s_bl;
            return;
        }
        // no pinned buffer: the helper has drawn nothing; start over on this thread
        spec_->th.join();
        spec_.reset();
        Transcript t = tr_;               // tr_ holds "m" already
        rng_.reset(new TranscriptRng(t.build_rng()));
        for (size_t i = 0; i < v_.size(); i++) {
            uint8_t b[32];
            sc_tobytes(b, v_blinding_[i]);
            rng_->rekey_with_witness_bytes("v_blinding", b, 32);
        }
        rng_->finalize(ext_);
    }
    uint8_t *slot = nullptr;
    if (raw_draws() && gens_)
        if (bpg_lane *lane = lane_of(gens_->ctx))
            if (grow_pinned(&lane->h_raw, &lane->h_raw_cap, 128 * a_L_.size()) == BPG_OK) slot = (uint8_t *)lane->h_raw;
    draw_blindings_into(slot);
}

// pinned_slot: where the raw draws (128 n bytes) go in the lane's pinned staging; null: pageable memory of the prover's own
void Prover::draw_blindings_into(uint8_t *pinned_slot) {
    const double t0 = now_ms();
    const size_t n = a_L_.size();
    i_bl_ = rng_->scalar();
    o_bl_ = rng_->scalar();
    s_bl_ = rng_->scalar();
    if (raw_draws()) {
        uint8_t *dst = s_raw_pin_ = pinned_slot;
        if (!dst) {
            s_raw_.resize(128 * n);
            dst = s_raw_.data();
        }
        for (size_t i = 0; i < 2 * n; i++) rng_->fill_bytes(dst + 64 * i, 64);           // s_L[0..n) then s_R[0..n)
    } else {
        s_L_.resize(n);
        s_R_.resize(n);
        for (size_t i = 0; i < n; i++) s_L_[i] = rng_->scalar();
        for (size_t i = 0; i < n; i++) s_R_[i] = rng_->scalar();
    }
    timing.rng_ms = now_ms() - t0;
}

int Prover::prove_many(Prover *const *ps, size_t k, rng_fill_fn rng_fn, void *rng_user, R1CSProof *out) {
    if (k == 1) return ps[0]->prove(rng_fn, rng_user, out[0]);
    for (size_t j = 0; j < k; j++) ps[j]->cancel_blindings();       // early draws are for a prover proved alone
    for (size_t j = 0; j < k; j++) {
        int rc = ps[j]->prove_begin(rng_fn, rng_user);
        if (rc) return rc;
    }
    // chunks of GROUP_MAX provers: blindings drawn eight provers at a time (TranscriptRngLanes8), then one GPU pass for
    // the chunk.  (prove_begin has asked the external RNG for every prover already, in order: the bytes do not depend
    // on how the provers are chunked.)
    for (size_t c0 = 0; c0 < k; c0 += GROUP_MAX) {
        const size_t G = k - c0 < GROUP_MAX ? k - c0 : GROUP_MAX;
        bool done[GROUP_MAX] = {};
        // the chunk's raw draws go straight into the lane's pinned staging when its provers share a context and size
        const size_t n0 = ps[c0]->a_L_.size();
        bool one_ctx = ps[c0]->gens_ != nullptr && ps[c0]->raw_draws();
        for (size_t l = 1; l < G && one_ctx; l++)
            one_ctx = ps[c0 + l]->gens_ && ps[c0 + l]->gens_->ctx == ps[c0]->gens_->ctx && ps[c0 + l]->a_L_.size() == n0 && ps[c0 + l]->raw_draws();
        bpg_lane *lane = one_ctx ? lane_of(ps[c0]->gens_->ctx) : nullptr;
        const bool pinned = lane && grow_pinned(&lane->h_raw, &lane->h_raw_cap, G * 128 * n0) == BPG_OK;
        for (size_t j0 = c0; j0 < c0 + G; j0 += 8) {
            const size_t g = c0 + G - j0 < 8 ? c0 + G - j0 : 8;
            const size_t n = ps[j0]->a_L_.size();
            bool same = g > 1;
            for (size_t l = 1; l < g; l++) same = same && ps[j0 + l]->a_L_.size() == n;
            for (size_t l = 0; l < g; l++) same = same && ps[j0 + l]->deferred_.empty();
            TranscriptRng *r[8];
            for (size_t l = 0; l < 8; l++) r[l] = ps[j0 + (l < g ? l : 0)]->rng_.get();     // spare lanes repeat lane 0
            TranscriptRngLanes8 lanes;
            if (!same || !lanes.load(r)) {                      // different circuits: every prover on its own, start to finish
                for (size_t l = 0; l < g; l++) {                //   (one prover's raw draws fill the lane's pinned staging at a time:
                    if (pinned) continue;                       //    not while the chunk's other draws sit there)
                    ps[j0 + l]->draw_blindings();
                    if (int rc = finish_group(ps + j0 + l, 1, out + j0 + l)) return rc;
                    done[j0 + l - c0] = true;
                }
                if (!pinned) continue;
                for (size_t l = 0; l < g; l++)                  // pinned chunk whose lanes would not load (or a lone last prover):
                    ps[j0 + l]->draw_blindings_into((uint8_t *)lane->h_raw + (j0 + l - c0) * 128 * n);     // one by one, into its slot
                continue;
            }
            HostPhase hp(HostPhase::RNG_DRAW);
            const double t0 = now_ms();
            sc d[8];
            lanes.scalars(d);
            for (size_t l = 0; l < g; l++) ps[j0 + l]->i_bl_ = d[l];
            lanes.scalars(d);
            for (size_t l = 0; l < g; l++) ps[j0 + l]->o_bl_ = d[l];
            lanes.scalars(d);
            for (size_t l = 0; l < g; l++) ps[j0 + l]->s_bl_ = d[l];
            if (ps[j0]->raw_draws()) {
                uint8_t buf[8][64], *dst[8];
                for (size_t l = 0; l < g; l++) {
                    Prover &p = *ps[j0 + l];
                    if (pinned) {
                        dst[l] = p.s_raw_pin_ = (uint8_t *)lane->h_raw + (j0 + l - c0) * 128 * n;
                    } else {
                        p.s_raw_.resize(128 * n);
                        dst[l] = p.s_raw_.data();
                    }
                }
                for (size_t i = 0; i < 2 * n; i++) {
                    lanes.fill64(buf);
                    for (size_t l = 0; l < g; l++) memcpy(dst[l] + 64 * i, buf[l], 64);
                }
            } else {
                for (size_t l = 0; l < g; l++) {
                    ps[j0 + l]->s_L_.resize(n);
                    ps[j0 + l]->s_R_.resize(n);
                }
                for (size_t i = 0; i < n; i++) {
                    lanes.scalars(d);
                    for (size_t l = 0; l < g; l++) ps[j0 + l]->s_L_[i] = d[l];
                }
                for (size_t i = 0; i < n; i++) {
                    lanes.scalars(d);
                    for (size_t l = 0; l < g; l++) ps[j0 + l]->s_R_[i] = d[l];
                }
            }
            lanes.store(r, (int)g);
            const double per = (now_ms() - t0) / (double)g;
            for (size_t l = 0; l < g; l++) ps[j0 + l]->timing.rng_ms = per;
        }
        // the GPU part, over the runs of provers that are not finished yet
        for (size_t a = 0; a < G;) {
            if (done[a]) {
                a++;
                continue;
            }
            size_t b = a;
            while (b < G && !done[b]) b++;
            if (int rc = finish_chunk(ps + c0 + a, b - a, out + c0 + a)) return rc;
            a = b;
        }
    }
    return BPG_OK;
}

// the GPU part of one chunk of up to GROUP_MAX provers whose blindings are drawn (the chunk's raw draws sit in the lane's
// pinned staging until then): consecutive provers of one size and generator object share their MSM launches
int Prover::finish_chunk(Prover *const *ps, size_t k, R1CSProof *out) {
    for (size_t j0 = 0; j0 < k;) {
        size_t g = 1;
        while (j0 + g < k && ps[j0 + g]->a_L_.size() == ps[j0]->a_L_.size() && ps[j0 + g]->gens_ == ps[j0]->gens_ &&
               ps[j0]->deferred_.empty() && ps[j0 + g]->deferred_.empty())       // randomized constraints: one prover at a time
            g++;
        int rc = finish_group(ps + j0, g, out + j0);
        if (rc) return rc;
        j0 += g;
    }
    return BPG_OK;
}

int Prover::prove_finish(R1CSProof &P) {
    Prover *self = this;
    return finish_group(&self, 1, &P);
}

// Everything after the blinding draws, for g provers of the same size at once on the calling thread's lane.
// Each proof keeps its own transcript, challenges and vectors; what is shared is the launch of every
// multiscalar multiplication: one 3g-target MSM for the A_I / A_O / S commitments and one 2g-target MSM per
// inner-product round, so the latency-bound kernels around the accumulate pass run once per round, not once
// per round and proof.  g = 1 is the single proof.
int Prover::finish_group(Prover *const *ps, size_t g, R1CSProof *out) {
    struct PF {
        std::shared_ptr<FlatCached> fc;
        sc *d_rows, *d_aL, *d_aR, *d_aO, *d_sL, *d_sR;
        sc *d_W, *d_ytbl, *d_yinvtbl, *d_l1, *d_r0, *d_r1, *d_r3, *d_part;
        sc *d_a, *d_b, *d_sG, *d_sH, *d_rowLR, *d_cpart;
        std::vector<sc> h_part, wV;
        Scalar y, z, u, x, w, t[7], tb[7];
        sc ytbl[32], yinvtbl[32];                           // sources of async copies: alive until the next sync
    };
    Gens *gens = ps[0]->gens_;
    bpg_ctx *ctx = gens->ctx;
    bpg_lane *lane = lane_of(ctx);
    if (!lane) return BPG_EDEVICE;
    hipStream_t s = lane->stream;
    for (size_t j = 0; j < g; j++) {
        ps[j]->n1_ = ps[j]->a_L_.size();
        ps[j]->i_bl2_ = ps[j]->o_bl2_ = ps[j]->s_bl2_ = sc_zero();
    }
    // Two-phase state is per prove() call: whatever path leaves this function (a failed allocation, exchange or HIP
    // call included), a later prove() on the same prover starts from a clean slate instead of skipping the A / S
    // commitments of a prologue that belongs to this call.
    struct PhaseReset {
        Prover *const *ps;
        size_t g;
        ~PhaseReset() {
            for (size_t j = 0; j < g; j++) {
                ps[j]->two_phase_done_ = false;
                ps[j]->n1_ = 0;
                ps[j]->i_bl2_ = ps[j]->o_bl2_ = ps[j]->s_bl2_ = sc_zero();
            }
        }
    } phase_reset{ps, g};
    if (!ps[0]->deferred_.empty()) {
        if (g != 1) return BPG_EINVAL;
        int rc2 = ps[0]->two_phase_prologue(out[0]);
        if (rc2) return rc2;
    }
    const bool two_phase = ps[0]->two_phase_done_;
    const size_t n = ps[0]->a_L_.size();
    // BPG_TIME_KERNELS=1: a proof alone times its scalar-vector kernels (start / stop events on the dispatches)
    hipEvent_t *evk = nullptr;
    ps[0]->ktiming.valid = false;
    if (ctx->time_kernels && g == 1 && n) {
        evk = lane->ev_k;
        for (auto &e : lane->ev_k)
            if (!e) BPG_CHECK_HIP(hipEventCreate(&e));
    }
    for (size_t j = 1; j < g; j++)
        if (ps[j]->a_L_.size() != n || ps[j]->gens_ != gens) return BPG_EINVAL;
    const size_t np = next_pow2(n ? n : 1);
    size_t lg = 0;
    while (((size_t)1 << lg) < np) lg++;
    std::vector<PF> pf(g);

    // ---- device workspace
    const size_t N1 = 2 * n + 1, NR = 2 * np + 1;
    const size_t nblk_n = grid_strip(n ? n : 1), nblk_np = grid256(np);
    // materialize the folded generators after round k0 (see msm_fold_submit): later rounds run on 2m points per proof
    int k0 = ctx->tune.ipp_fold < 0 ? ipp_fold_default(lg) : ctx->tune.ipp_fold;
    if (gens->world != 1 || k0 < 1 || (size_t)k0 >= lg || (np >> k0) < 32) k0 = 0;
    // the rounds after the fold need the folded generators' own fixed-base table (below): no fold where that table
    // would pass 2^17 points or a side keeps more than 4096 generators (sizes this path has not been run at)
    if (k0 && ((np >> k0) > 4096 || g * 2 * (np >> k0) + 1 > ((size_t)1 << 17))) k0 = 0;
    const size_t fm = k0 ? np >> k0 : 0;                  // folded generators per side
    // The rounds after the fold run FIXED-BASE as well: the folded generators get their own table of 2^(c w) multiples
    // (one doubling chain per point, built once per group).  A variable-base MSM leaves ~250 doublings per target to
    // the host's window combination -- 16 targets x 11 rounds of them were more than half of a group's inner-product
    // latency and 2.8 ms of host CPU per proof; fixed-base leaves c - 1.
    const size_t fpts = g * 2 * fm + 1;                   // [tab_0 | .. | tab_{g-1} | B]
    const bool fold_fb = k0 != 0;
    const int fold_c = 9, fold_W = fold_fb ? fixed_base_windows(fold_c) : 1;
    // MSM scalar rows: 2g rows of NR before the fold, of g * 2 fm + 1 after it
    const size_t row_words = std::max(g * 2 * NR, 2 * g * (g * 2 * fm + 1));
    size_t need = pad256(g * 3 * N1 * 32) + pad256(row_words * 32) + pad256(g * 2 * np * 32) + pad256((size_t)fold_W * fpts * 128) + (1 << 16) +
                  g * pad256(128 * n + 64);
    for (size_t j = 0; j < g; j++) {
        Prover &p = *ps[j];
        int rcf = flat_plan_cached(ctx, lane, p.cs_, (uint32_t)n, (uint32_t)p.v_.size(), false, pf[j].fc);
        if (rcf) return rcf;
        need += 5 * pad256((n + 1) * 32) + pad256((pf[j].fc->fp.nslots + 1) * 32) + flatten_bytes(p.cs_, pf[j].fc->fp) +
                4 * pad256((n + 1) * 32) + pad256(nblk_n * 6 * 32) + 2 * pad256(32 * 32) + 4 * pad256(np * 32) +
                pad256(nblk_np * 2 * 32) + (1 << 12);
    }
    Arena ar(lane);
    int rc = ar.reserve(need);
    if (rc) return rc;
    PinArena pin(lane);
    {
        size_t pneed = 1 << 16;
        for (size_t j = 0; j < g; j++)
            pneed += 4096 + 3 * 1024 + (ps[j]->cs_.coeffs.size() + 1) * 32 + nblk_n * 6 * 32 + (ps[j]->v_.size() + 1) * 32 + 64 * 8 +
                     (n * 32 <= PinArena::SMALL ? 5 * (n * 32 + 64) : 0);
        pin.reserve(pneed);
    }
    std::unique_ptr<HostPhase> hp_up(new HostPhase(HostPhase::UPLOAD));
    sc *d_rows_all = ar.take<sc>(g * 3 * N1);             // proof j: rows 3j .. 3j+2
    sc *d_rowLR_all = ar.take<sc>(row_words);             // proof j: rows 2j, 2j+1
    sc *d_s_all = ar.take<sc>(g * 2 * np);                // proof j: [sG (np) | sH (np)]
    aniels_rec *d_tab_all = (aniels_rec *)ar.take<uint8_t>((size_t)fold_W * fpts * 128);   // proof j: [G' (fm) | H' (fm)]; then B; then the shifted blocks
    BPG_CHECK_HIP(hipMemsetAsync(d_rows_all, 0, g * 3 * N1 * 32, s));
    std::vector<std::vector<sc>> keep_sL(g), keep_sR(g);  // pageable sources of async copies: alive until the sync
    std::vector<std::vector<uint8_t>> keep_raw(g);
    uint4 *spec_d_raw = nullptr;
    double spec_wait_ms = 0;                               // of the commit phase: waiting for the helper's draws
    // the witness vectors go through the lane's pinned staging too (one host copy, then DMA)
    uint8_t *h_wit = n && grow_pinned(&lane->h_wit, &lane->h_wit_cap, g * (3 * n + 1) * 32) == BPG_OK ? (uint8_t *)lane->h_wit : nullptr;
    // the group's raw draws in one piece (device side), when the host side is one piece too (prove_many's pinned chunk)
    bool raw_whole = g > 1 && n && lane->h_raw != nullptr;
    for (size_t j = 0; j < g && raw_whole; j++)
        raw_whole = !ps[j]->spec_ && ps[j]->s_raw_pin_ == (uint8_t *)lane->h_raw + j * 128 * n;
    uint4 *d_raw_whole = raw_whole ? (uint4 *)ar.take<uint8_t>(g * 128 * n) : nullptr;
    for (size_t j = 0; j < g; j++) {
        Prover &p = *ps[j];
        PF &f = pf[j];
        // the blinding vectors are only needed up to the upload; the prover keeps nothing of them afterwards
        keep_sL[j] = std::move(p.s_L_);
        keep_sR[j] = std::move(p.s_R_);
        keep_raw[j] = std::move(p.s_raw_);
        p.s_L_.clear();
        p.s_R_.clear();
        p.s_raw_.clear();
        f.d_rows = d_rows_all + j * 3 * N1;
        f.d_aL = f.d_rows + 1;                              // row 0 = [i_bl | a_L | a_R]
        f.d_aR = f.d_rows + 1 + n;
        f.d_aO = f.d_rows + N1 + 1;                         // row 1 = [o_bl | a_O | 0]
        f.d_sL = f.d_rows + 2 * N1 + 1;                     // row 2 = [s_bl | s_L | s_R]
        f.d_sR = f.d_rows + 2 * N1 + 1 + n;
        BPG_CHECK_HIP(pin.h2d(f.d_rows, &p.i_bl_, 32, s));
        BPG_CHECK_HIP(pin.h2d(f.d_rows + N1, &p.o_bl_, 32, s));
        BPG_CHECK_HIP(pin.h2d(f.d_rows + 2 * N1, &p.s_bl_, 32, s));
        if (n) {
            const sc *hL = p.a_L_.data(), *hR = p.a_R_.data(), *hO = p.a_O_.data();
            const bool small_wit = n * 32 <= PinArena::SMALL;       // a small circuit's vectors join the batched move
            if (small_wit) {
                BPG_CHECK_HIP(pin.h2d(f.d_aL, hL, n * 32, s));
                BPG_CHECK_HIP(pin.h2d(f.d_aR, hR, n * 32, s));
                BPG_CHECK_HIP(pin.h2d(f.d_aO, hO, n * 32, s));
            } else if (h_wit) {
                // rows 0 and 1 are adjacent on the device -- [i_bl | a_L | a_R] [o_bl | a_O | 0] -- so a_L | a_R | o_bl | a_O is
                // ONE transfer (each command for the copy engines is a turn in their queues; see PinArena)
                sc *w = (sc *)(h_wit + j * (3 * n + 1) * 32);
                memcpy(w, hL, n * 32);
                memcpy(w + n, hR, n * 32);
                w[2 * n] = p.o_bl_;
                memcpy(w + 2 * n + 1, hO, n * 32);
                BPG_CHECK_HIP(hipMemcpyAsync(f.d_aL, w, (3 * n + 1) * 32, hipMemcpyHostToDevice, s));
            } else {
                BPG_CHECK_HIP(hipMemcpyAsync(f.d_aL, hL, n * 32, hipMemcpyHostToDevice, s));
                BPG_CHECK_HIP(hipMemcpyAsync(f.d_aR, hR, n * 32, hipMemcpyHostToDevice, s));
                BPG_CHECK_HIP(hipMemcpyAsync(f.d_aO, hO, n * 32, hipMemcpyHostToDevice, s));
            }
            const uint8_t *raw = p.s_raw_pin_ ? p.s_raw_pin_ : keep_raw[j].empty() ? nullptr : keep_raw[j].data();
            if (p.spec_) {                           // early draws: streamed below, once A_I and A_O are under way
                if (g != 1) return BPG_EINVAL;
                spec_d_raw = (uint4 *)ar.take<uint8_t>(128 * n);
            } else if (raw) {                        // raw draws: reduced mod l on the device, straight into row 2
                if (!p.s_raw_pin_ && keep_raw[j].size() != 128 * n) return BPG_EINVAL;
                p.s_raw_pin_ = nullptr;
                uint4 *d_raw = d_raw_whole ? d_raw_whole + j * 8 * n : (uint4 *)ar.take<uint8_t>(128 * n);
                if (!d_raw_whole) BPG_CHECK_HIP(hipMemcpyAsync(d_raw, raw, 128 * n, hipMemcpyHostToDevice, s));
                else if (j == 0) BPG_CHECK_HIP(hipMemcpyAsync(d_raw_whole, lane->h_raw, g * 128 * n, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_reduce512, dim3(grid256(2 * n)), dim3(256), 0, s, d_raw, f.d_sL, (uint32_t)(2 * n));
            } else {
                if (keep_sL[j].size() != n || keep_sR[j].size() != n) return BPG_EINVAL;
                BPG_CHECK_HIP(pin.h2d(f.d_sL, keep_sL[j].data(), n * 32, s));      // (batched when small; a plain copy otherwise)
                BPG_CHECK_HIP(pin.h2d(f.d_sR, keep_sR[j].data(), n * 32, s));
            }
        }
    }
    BPG_CHECK_HIP(pin.flush(s));
    BPG_CHECK_HIP(lane_sync(lane));
    keep_sL.clear();
    keep_sR.clear();
    keep_raw.clear();
    hp_up.reset();

    // ---- A_I1, A_O1, S1 of every proof: one 3g-target MSM over [B~ | G[0..n) | H[0..n)]
    const double t1 = now_ms();
    if (spec_d_raw) {
        // A_I and A_O need none of the s_L / s_R draws: commit them while the helper is still drawing, then stream the
        // draws to the device in chunks as they appear (reduced mod l there), and commit S when the last one is in
        Prover &p = *ps[0];
        BlindingSpec &sp = *p.spec_;
        msm_seg segs[3] = {pedersen_seg(gens, 1, 1, 0), gen_seg(gens, 0, 0, n, 1), gen_seg(gens, 1, 0, n, 1 + n)};
        uint8_t o[96];
        if ((rc = gens_msm(gens, segs, 3, d_rows_all, N1, 2, o))) return rc;
        const double t_wait0 = now_ms();
        const size_t total = 2 * n, CH = (size_t)1 << 16;
        for (size_t up = 0; up < total;) {
            const size_t want = total - up < CH ? total : up + CH;
            while (sp.done.load(std::memory_order_acquire) < want) {
                struct timespec ts = {0, 50000};
                nanosleep(&ts, nullptr);
            }
            BPG_CHECK_HIP(hipMemcpyAsync((uint8_t *)spec_d_raw + 64 * up, sp.raw + 64 * up, 64 * (want - up), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_reduce512, dim3(grid256(want - up)), dim3(256), 0, s, spec_d_raw + 4 * up, pf[0].d_sL + up, (uint32_t)(want - up));
            BPG_CHECK_HIP(hipGetLastError());
            up = want;
        }
        BPG_CHECK_HIP(lane_sync(lane));
        sp.th.join();                                  // done == 2 n: the helper has left its loop
        p.rng_ = std::move(sp.rng);                    // the t blindings come next from the same RNG
        p.timing.rng_ms = sp.busy_ms;
        spec_wait_ms = now_ms() - t_wait0;
        spec_release(sp);
        p.spec_.reset();
        if ((rc = gens_msm(gens, segs, 3, d_rows_all + 2 * N1, N1, 1, o + 64))) return rc;
        memcpy(out[0].A_I1, o, 32);
        memcpy(out[0].A_O1, o + 32, 32);
        memcpy(out[0].S1, o + 64, 32);
    } else if (!two_phase) {
        msm_seg segs[3] = {pedersen_seg(gens, 1, 1, 0), gen_seg(gens, 0, 0, n, 1), gen_seg(gens, 1, 0, n, 1 + n)};
        std::vector<uint8_t> o(96 * g);
        if ((rc = gens_msm(gens, segs, 3, d_rows_all, N1, (int)(3 * g), o.data()))) return rc;
        for (size_t j = 0; j < g; j++) {
            memcpy(out[j].A_I1, &o[96 * j], 32);
            memcpy(out[j].A_O1, &o[96 * j + 32], 32);
            memcpy(out[j].S1, &o[96 * j + 64], 32);
        }
    }
    const double commit_ms = (now_ms() - t1 - spec_wait_ms) / (double)g;      // (not the wait for a helper's draws: that is rng time)

    // ---- challenges y, z; flatten, l/r polynomials, t coefficients
    const double t2 = now_ms();
    std::unique_ptr<HostPhase> hp_poly(new HostPhase(HostPhase::FLATTEN_POLY));
    poly_group pgrp{};
    eval_group egrp{};
    for (size_t j = 0; j < g; j++) {
        Prover &p = *ps[j];
        PF &f = pf[j];
        R1CSProof &P = out[j];
        const size_t m = p.v_.size();
        p.timing.commit_ms = commit_ms;
        if (!two_phase) {                        // (two_phase_prologue has done all of this for a two-phase proof)
            p.tr_.append_point("A_I1", P.A_I1);
            p.tr_.append_point("A_O1", P.A_O1);
            p.tr_.append_point("S1", P.S1);
            p.tr_.r1cs_1phase_domain_sep();      // no randomized constraints (none of the reference's gadgets registers any)
            memset(P.A_I2, 0, 32);
            memset(P.A_O2, 0, 32);
            memset(P.S2, 0, 32);
            p.tr_.append_point("A_I2", P.A_I2);
            p.tr_.append_point("A_O2", P.A_O2);
            p.tr_.append_point("S2", P.S2);
        }
        f.y = p.tr_.challenge_scalar("y");
        f.z = p.tr_.challenge_scalar("z");
        f.d_W = ar.take<sc>(f.fc->fp.nslots + 1);
        if ((rc = gpu_flatten(lane, ar, pin, p.cs_, *f.fc, f.z.v, f.d_W, nullptr, nullptr, evk))) return rc;
        const Scalar y_inv = f.y.invert();
        pow2_table_mont(f.ytbl, f.y.v);
        pow2_table_mont(f.yinvtbl, y_inv.v);
        f.d_ytbl = ar.take<sc>(32);
        f.d_yinvtbl = ar.take<sc>(32);
        BPG_CHECK_HIP(pin.h2d(f.d_ytbl, f.ytbl, sizeof f.ytbl, s));
        BPG_CHECK_HIP(pin.h2d(f.d_yinvtbl, f.yinvtbl, sizeof f.yinvtbl, s));
        f.d_l1 = ar.take<sc>(n + 1);
        f.d_r0 = ar.take<sc>(n + 1);
        f.d_r1 = ar.take<sc>(n + 1);
        f.d_r3 = ar.take<sc>(n + 1);
        f.d_part = ar.take<sc>(nblk_n * 6);
        f.h_part.resize(nblk_n * 6);
        f.wV.resize(m);
        pgrp.p[j] = poly_proof{f.d_aL, f.d_aR, f.d_aO, f.d_sL, f.d_sR, f.d_W, f.d_ytbl, f.d_yinvtbl, f.d_l1, f.d_r0, f.d_r1, f.d_r3, f.d_part};
    }
    BPG_CHECK_HIP(pin.flush(s));
    if (n) {
        hipExtLaunchKernelGGL(k_poly, dim3((unsigned)nblk_n, (unsigned)g), dim3(256), 0, s, evk ? evk[4] : nullptr, evk ? evk[5] : nullptr, 0,
                              (uint32_t)n, pgrp);
        BPG_CHECK_HIP(hipGetLastError());
    }
    for (size_t j = 0; j < g; j++) {
        PF &f = pf[j];
        const size_t m = ps[j]->v_.size();
        if (n) BPG_CHECK_HIP(pin.d2h(f.h_part.data(), f.d_part, nblk_n * 6 * 32, s));
        if (m) BPG_CHECK_HIP(pin.d2h(f.wV.data(), f.d_W + 3 * n, m * 32, s));
    }
    BPG_CHECK_HIP(pin.flush(s));
    BPG_CHECK_HIP(lane_sync(lane));
    pin.collect();

    hp_poly.reset();
    // ---- T_1, T_3..T_6 of every proof: one batch of 5g Pedersen commitments
    {
        HostPhase hp_t(HostPhase::T_COMMIT);
        std::vector<sc> vr(10 * g);
        std::vector<uint8_t> Ts(160 * g);
        for (size_t j = 0; j < g; j++) {
            PF &f = pf[j];
            TranscriptRng &rng = *ps[j]->rng_;
            if (n)
                for (size_t b = 0; b < nblk_n; b++)
                    for (int k = 0; k < 6; k++) f.t[k + 1] += Scalar(f.h_part[b * 6 + k]);
            f.tb[1] = rng.scalar(); f.tb[3] = rng.scalar(); f.tb[4] = rng.scalar(); f.tb[5] = rng.scalar(); f.tb[6] = rng.scalar();
            const sc row[10] = {f.t[1].v, f.tb[1].v, f.t[3].v, f.tb[3].v, f.t[4].v, f.tb[4].v, f.t[5].v, f.tb[5].v, f.t[6].v, f.tb[6].v};
            memcpy(&vr[10 * j], row, sizeof row);
        }
        if ((rc = pedersen_batch(gens, vr.data(), 5 * g, Ts.data()))) return rc;
        for (size_t j = 0; j < g; j++) {
            R1CSProof &P = out[j];
            const uint8_t *T5 = &Ts[160 * j];
            memcpy(P.T_1, T5, 32); memcpy(P.T_3, T5 + 32, 32); memcpy(P.T_4, T5 + 64, 32);
            memcpy(P.T_5, T5 + 96, 32); memcpy(P.T_6, T5 + 128, 32);
        }
    }

    // ---- challenges u, x, w; l(x), r(x) -> IPP vectors
    for (size_t j = 0; j < g; j++) {
        Prover &p = *ps[j];
        PF &f = pf[j];
        R1CSProof &P = out[j];
        const size_t m = p.v_.size();
        p.tr_.append_point("T_1", P.T_1);
        p.tr_.append_point("T_3", P.T_3);
        p.tr_.append_point("T_4", P.T_4);
        p.tr_.append_point("T_5", P.T_5);
        p.tr_.append_point("T_6", P.T_6);
        f.u = p.tr_.challenge_scalar("u");
        f.x = p.tr_.challenge_scalar("x");
        const Scalar &x = f.x;
        const Scalar *t = f.t;
        Scalar *tb = f.tb;
        for (size_t i = 0; i < m; i++) tb[2] += Scalar(f.wV[i]) * Scalar(p.v_blinding_[i]);
        P.t_x = x * (t[1] + x * (t[2] + x * (t[3] + x * (t[4] + x * (t[5] + x * t[6])))));
        P.t_x_blinding = x * (tb[1] + x * (tb[2] + x * (tb[3] + x * (tb[4] + x * (tb[5] + x * tb[6])))));
        const Scalar ibl = Scalar(p.i_bl_) + f.u * Scalar(p.i_bl2_), obl = Scalar(p.o_bl_) + f.u * Scalar(p.o_bl2_),
                     sbl = Scalar(p.s_bl_) + f.u * Scalar(p.s_bl2_);
        P.e_blinding = x * (ibl + x * (obl + x * sbl));
        p.tr_.append_scalar("t_x", P.t_x.v);
        p.tr_.append_scalar("t_x_blinding", P.t_x_blinding.v);
        p.tr_.append_scalar("e_blinding", P.e_blinding.v);
        f.w = p.tr_.challenge_scalar("w");
        f.d_a = ar.take<sc>(np);
        f.d_b = ar.take<sc>(np);
        f.d_sG = d_s_all + j * 2 * np;
        f.d_sH = f.d_sG + np;
        f.d_cpart = ar.take<sc>(nblk_np * 2);
        f.d_rowLR = d_rowLR_all + j * 2 * NR;
        {
            const sc xR = to_mont(f.x), xRR = to_mont(Scalar(xR)), x3RR = to_mont(Scalar(to_mont(f.x * f.x * f.x)));
            egrp.p[j] = eval_proof{(uint32_t)p.n1_, xR, xRR, x3RR, f.u.v, f.d_l1, f.d_aO, f.d_sL, f.d_r0, f.d_r1, f.d_r3, f.d_ytbl, f.d_yinvtbl,
                                   f.d_a, f.d_b, f.d_sG, f.d_sH};
        }
        p.tr_.innerproduct_domain_sep(np);
        P.L_vec.assign(32 * lg, 0);
        P.R_vec.assign(32 * lg, 0);
    }
    hipExtLaunchKernelGGL(k_eval, dim3(grid_strip(np), (unsigned)g), dim3(256), 0, s, evk ? evk[6] : nullptr, evk ? evk[7] : nullptr, 0,
                          (uint32_t)n, (uint32_t)np, egrp);
    BPG_CHECK_HIP(hipGetLastError());
    const double poly_ms = (now_ms() - t2) / (double)g;

    // ---- inner-product argument, fold-free: one 2g-target MSM per round
    const double t3 = now_ms();
    size_t nk = np;
    std::vector<uint8_t> lr(64 * g);
    // until the fold: vectors over the np original generators, rows [G | H | B] of NR scalars.  After it: over
    // the fm folded ones of each proof; the group's tables form one point set [tab_0 | .. | tab_{g-1} | B] and
    // proof j's two rows are zero outside its own block, so one fixed-base MSM over the folded tables still serves the group.
    size_t cnp = np, cstride = NR, cblk = nblk_np;
    ipp_group grp{};
    for (size_t k = 0; k < lg; k++, nk >>= 1) {
        if (k0 && k == (size_t)k0) {
            HostPhase hp_f(HostPhase::FOLD);
            BPG_CHECK_HIP(lane_sync(lane));         // the folds of round k-1 wrote sG / sH
            msm_seg fs[2] = {gen_seg(gens, 0, 0, np, 0), gen_seg(gens, 1, 0, np, np)};
            bpg_msm_job *job = nullptr;
            if ((rc = msm_fold_submit(ctx, fs, d_s_all, 2 * np, (int)g, np, fm, msm_fixed_base{gens->tab_c, gens->npoints(), 1.f},
                                      d_tab_all, &job)))
                return rc;
            if ((rc = msm_fold_wait(job))) return rc;
            hipLaunchKernelGGL(k_copy16, dim3(1), dim3(256), 0, s, (uint4 *)(d_tab_all + g * 2 * fm),
                               (const uint4 *)(gens->table->d_table + gens->off_B()), (uint32_t)(sizeof(aniels_rec) / 16));
            if ((rc = table_build_shifted(ctx, d_tab_all, fpts, fold_c, fold_W))) return rc;
            cnp = fm;
            cstride = g * 2 * fm + 1;
            cblk = grid256(fm);
            BPG_CHECK_HIP(hipMemsetAsync(d_rowLR_all, 0, 2 * g * cstride * 32, s));
            hipLaunchKernelGGL(k_fill_ones, dim3(grid256(fm), (unsigned)(2 * g)), dim3(256), 0, s, d_s_all, (uint32_t)np, (uint32_t)fm);
            for (size_t j = 0; j < g; j++) pf[j].d_rowLR = d_rowLR_all + 2 * j * cstride + j * 2 * fm;
        }
        std::unique_ptr<HostPhase> hp_rows(new HostPhase(HostPhase::IPP_ROWS));
        const size_t boff = cnp == np ? 2 * np : cstride - 1;        // where a row keeps the scalar of B
        for (size_t j = 0; j < g; j++) {
            PF &f = pf[j];
            sc *row0 = cnp == np ? f.d_rowLR : d_rowLR_all + 2 * j * cstride;
            ipp_proof &q = grp.p[j];
            q.a = f.d_a; q.b = f.d_b; q.sG = f.d_sG; q.sH = f.d_sH;
            q.rowL = f.d_rowLR; q.rowR = f.d_rowLR + cstride;
            q.cpart = f.d_cpart;
            q.outL = row0 + boff; q.outR = row0 + cstride + boff;
            q.w = f.w.v;
        }
        const bool tk = evk && k == 0;
        hipExtLaunchKernelGGL(k_ipp_rows, dim3((unsigned)cblk, (unsigned)g), dim3(256), 0, s, tk ? evk[8] : nullptr, tk ? evk[9] : nullptr, 0,
                              (uint32_t)cnp, (uint32_t)nk, grp);
        hipLaunchKernelGGL(k_ipp_c, dim3(1, (unsigned)g), dim3(256), 0, s, (uint32_t)cblk, grp);
        BPG_CHECK_HIP(hipGetLastError());
        BPG_CHECK_HIP(lane_sync(lane));
        hp_rows.reset();
        if (cnp == np) {
            msm_seg segs[3] = {gen_seg(gens, 0, 0, np, 0), gen_seg(gens, 1, 0, np, np), pedersen_seg(gens, 0, 1, 2 * np)};
            if ((rc = gens_msm(gens, segs, 3, d_rowLR_all, NR, (int)(2 * g), lr.data(), 0.5f, HostPhase::IPP_SUBMIT, HostPhase::IPP_WAIT)))
                return rc;     // half of every row is zero
        } else {
            bpg_msm_job *job = nullptr;
            {
                HostPhase hp_s(HostPhase::IPP_SUBMIT);
                msm_seg one{d_tab_all, fpts, 0};                  // B sits behind the tables, where the rows keep its scalar
                rc = msm_submit(ctx, &one, 1, d_rowLR_all, cstride, (int)(2 * g), false, &job,
                                msm_fixed_base{fold_c, fpts, 1.f / (float)(2 * g)}, 0);
            }
            if (rc) return rc;
            HostPhase hp_w(HostPhase::IPP_WAIT);
            if ((rc = msm_wait(job, lr.data(), false))) return rc;
        }
        HostPhase hp_h(HostPhase::IPP_HOST);
        for (size_t j = 0; j < g; j++) {
            Prover &p = *ps[j];
            R1CSProof &P = out[j];
            const uint8_t *o = &lr[64 * j];
            memcpy(&P.L_vec[32 * k], o, 32);
            memcpy(&P.R_vec[32 * k], o + 32, 32);
            p.tr_.append_point("L", o);
            p.tr_.append_point("R", o + 32);
            const Scalar uk = p.tr_.challenge_scalar("u"), uk_inv = uk.invert();
            grp.p[j].uR = to_mont(uk);
            grp.p[j].uinvR = to_mont(uk_inv);
        }
        hipExtLaunchKernelGGL(k_ipp_fold, dim3((unsigned)cblk, (unsigned)g), dim3(256), 0, s, tk ? evk[10] : nullptr, tk ? evk[11] : nullptr, 0,
                              (uint32_t)cnp, (uint32_t)nk, grp);
        BPG_CHECK_HIP(hipGetLastError());
    }
    std::vector<sc> hab(2 * g);
    for (size_t j = 0; j < g; j++) {
        BPG_CHECK_HIP(pin.d2h(&hab[2 * j], pf[j].d_a, 32, s));
        BPG_CHECK_HIP(pin.d2h(&hab[2 * j + 1], pf[j].d_b, 32, s));
    }
    BPG_CHECK_HIP(pin.flush(s));
    BPG_CHECK_HIP(lane_sync(lane));
    pin.collect();
    const double ipp_ms = (now_ms() - t3) / (double)g, t_end = now_ms();
    if (evk && lg >= 1) {
        Prover &p = *ps[0];
        const FlatPlan &fp = pf[0].fc->fp;
        const double Q = p.cs_.num_constraints, nt = (double)fp.nterms, ns = fp.nslots, dn = (double)n, dnp = (double)np;
        // bytes every kernel has to move (32-byte scalars) and its Montgomery multiplications
        const double bytes[N_TIMED] = {32 * Q, 40 * nt + 36 * ns, 384 * dn, 192 * dn + 128 * dnp, 320 * dnp, 224 * dnp};
        const double mm[N_TIMED] = {Q * (1 + 8.0 / STRIP), nt, dn * (15 + (6 + 16.0) / STRIP), dn * 6 + dnp * (2 + 8.0 / STRIP), 3 * dnp, 4 * dnp};
        for (int i = 0; i < N_TIMED; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, evk[2 * i], evk[2 * i + 1]);
            p.ktiming.us[i] = ms * 1e3;
            p.ktiming.bytes[i] = bytes[i];
            p.ktiming.montmuls[i] = mm[i];
        }
        p.ktiming.valid = true;
    }
    for (size_t j = 0; j < g; j++) {
        Prover &p = *ps[j];
        sc_from_mont(hab[2 * j], hab[2 * j]);                 // the vectors live as a R, b R on the device
        sc_from_mont(hab[2 * j + 1], hab[2 * j + 1]);
        out[j].a = Scalar(hab[2 * j]);
        out[j].b = Scalar(hab[2 * j + 1]);
        p.timing.poly_ms = poly_ms;
        p.timing.ipp_ms = ipp_ms;
        p.timing.total_ms = t_end - p.t_start_;
        p.rng_.reset();
    }
    return BPG_OK;
}

// =================================================================== Verifier

Verifier::Verifier(const void *label, size_t label_len) : tr_(label, label_len) {
    tr_.r1cs_domain_sep();
    if (auto e = last_shape(1)) follow(e->key, e);
}

Variable Verifier::commit(const uint8_t V32[32]) {
    uint32_t i = (uint32_t)(V_.size() / 32);
    V_.insert(V_.end(), V32, V32 + 32);
    tr_.append_point("V", V32);
    return Variable{VAR_COMMITTED, i};
}

ConstraintSystem::Triple Verifier::multiply(const LinearCombination &left, const LinearCombination &right) {
    uint32_t i = (uint32_t)num_vars_++;
    Triple t{{VAR_LEFT, i}, {VAR_RIGHT, i}, {VAR_OUTPUT, i}};
    cs_.push(left, num_vars_, V_.size() / 32, &t.l);
    cs_.push(right, num_vars_, V_.size() / 32, &t.r);
    return t;
}

Scalar Verifier::challenge_scalar(const std::string &label) {
    if (!phase2_) throw R1CSError(BPG_EINVAL, "challenge_scalar outside a randomized-constraints callback");
    return Scalar(tr_.challenge_scalar(label.c_str()));
}

ConstraintSystem::Triple Verifier::allocate_multiplier(const Scalar *, const Scalar *) {
    uint32_t i = (uint32_t)num_vars_++;
    return Triple{{VAR_LEFT, i}, {VAR_RIGHT, i}, {VAR_OUTPUT, i}};
}

// Everything up to the final multiscalar multiplication: transcript replay, challenges,
// and the scalar that multiplies every point of the verification equation.  out.d_sc
// (device, in the lane's arena) holds the 2 n+ generator scalars [g | h];
// out.tail the scalars of [B, B~, per-proof points]; out.dyn those points (compressed).
int Verifier::prepare(const R1CSProof &P, Gens *gens, rng_fill_fn rng_fn, void *rng_user, VerifyRow &out) {
    bpg_ctx *ctx = gens->ctx;
    bpg_lane *lane = lane_of(ctx);
    if (!lane) return BPG_EDEVICE;
    hipStream_t s = lane->stream;
    const size_t n1 = num_vars_, m = V_.size() / 32;
    tr_.append_u64("m", m);
    if (!tr_.validate_and_append_point("A_I1", P.A_I1) || !tr_.validate_and_append_point("A_O1", P.A_O1) ||
        !tr_.validate_and_append_point("S1", P.S1))
        return BPG_EVERIFY;
    if (deferred_.empty()) {
        tr_.r1cs_1phase_domain_sep();
    } else {                                     // randomized constraints: the callbacks build the second phase now
        tr_.r1cs_2phase_domain_sep();
        std::vector<RandomizedFn> cbs;
        cbs.swap(deferred_);
        phase2_ = true;
        try {
            for (auto &cb : cbs) cb(*this);
        } catch (...) {
            phase2_ = false;
            throw;
        }
        phase2_ = false;
    }
    const size_t n = num_vars_;
    const size_t np = next_pow2(n ? n : 1);
    size_t lg = 0;
    while (((size_t)1 << lg) < np) lg++;
    if (gens->capacity < np) return BPG_EGENS;
    tr_.append_point("A_I2", P.A_I2);
    tr_.append_point("A_O2", P.A_O2);
    tr_.append_point("S2", P.S2);
    const Scalar y = tr_.challenge_scalar("y"), z = tr_.challenge_scalar("z");
    if (!tr_.validate_and_append_point("T_1", P.T_1) || !tr_.validate_and_append_point("T_3", P.T_3) ||
        !tr_.validate_and_append_point("T_4", P.T_4) || !tr_.validate_and_append_point("T_5", P.T_5) ||
        !tr_.validate_and_append_point("T_6", P.T_6))
        return BPG_EVERIFY;
    const Scalar u = tr_.challenge_scalar("u"), x = tr_.challenge_scalar("x");
    tr_.append_scalar("t_x", P.t_x.v);
    tr_.append_scalar("t_x_blinding", P.t_x_blinding.v);
    tr_.append_scalar("e_blinding", P.e_blinding.v);
    const Scalar w = tr_.challenge_scalar("w");
    // IPP verification scalars
    if (P.L_vec.size() != 32 * lg || P.R_vec.size() != 32 * lg || lg >= 32) return BPG_EVERIFY;
    tr_.innerproduct_domain_sep(np);
    std::vector<sc> uk(lg + 1), ukinv(lg + 1), u_sq(lg), u_inv_sq(lg);
    for (size_t k = 0; k < lg; k++) {
        if (!tr_.validate_and_append_point("L", &P.L_vec[32 * k]) || !tr_.validate_and_append_point("R", &P.R_vec[32 * k]))
            return BPG_EVERIFY;
        Scalar c = tr_.challenge_scalar("u"), ci = c.invert();
        uk[k] = c.v;
        ukinv[k] = ci.v;
        u_sq[k] = (c * c).v;
        u_inv_sq[k] = (ci * ci).v;
    }
    // external randomness -> r
    TranscriptRng rng = tr_.build_rng();
    uint8_t ext[32];
    rng_fn(rng_user, ext, 32);
    rng.finalize(ext);
    const Scalar r = rng.scalar();
    // second draw of the same RNG: the weight of this proof's equation in a batch (verify_batch).  It is bound to
    // the whole transcript (statement and proof) and to the external randomness, like r itself.
    out.rho = rng.scalar();

    // per-proof points -> this lane's point scratch: [A_I1 A_O1 S1 A_I2 A_O2 S2 | V.. | T.. | L.. | R..]
    const size_t ndyn = 6 + m + 5 + 2 * lg;
    int rc;
    std::vector<uint8_t> dyn;
    dyn.reserve(32 * ndyn);
    auto put = [&](const uint8_t *p) { dyn.insert(dyn.end(), p, p + 32); };
    put(P.A_I1); put(P.A_O1); put(P.S1); put(P.A_I2); put(P.A_O2); put(P.S2);
    dyn.insert(dyn.end(), V_.begin(), V_.end());
    put(P.T_1); put(P.T_3); put(P.T_4); put(P.T_5); put(P.T_6);
    dyn.insert(dyn.end(), P.L_vec.begin(), P.L_vec.end());
    dyn.insert(dyn.end(), P.R_vec.begin(), P.R_vec.end());

    std::shared_ptr<FlatCached> fc;
    if ((rc = flat_plan_cached(ctx, lane, cs_, (uint32_t)n, (uint32_t)m, true, fc))) return rc;
    const FlatPlan &fp = fc->fp;
    const size_t NS = 2 * np + 2 + ndyn, nblk = grid256(np);
    size_t need = pad256((fp.nslots + 1) * 32) + flatten_bytes(cs_, fp) + pad256(32 * 32) + 2 * pad256((lg + 1) * 32) +
                  pad256(NS * 32) + pad256(nblk * 32) + (1 << 16);
    Arena ar(lane);
    if ((rc = ar.reserve(need))) return rc;
    sc *d_W = ar.take<sc>(fp.nslots + 1);
    sc *d_one = nullptr;
    size_t n_one = 0;
    PinArena pin(lane);
    pin.reserve((1 << 16) + (cs_.coeffs.size() + 1) * 32 + 3 * 1024 + 2 * (lg + 1) * 32 + (nblk + m + 2) * 32 + (fp.nones / 256 + 2) * 32);
    if ((rc = gpu_flatten(lane, ar, pin, cs_, *fc, z.v, d_W, &d_one, &n_one))) return rc;
    sc yinvtbl[32];
    pow2_table(yinvtbl, y.invert().v);
    sc *d_yinvtbl = ar.take<sc>(32), *d_uk = ar.take<sc>(lg + 1), *d_ukinv = ar.take<sc>(lg + 1);
    sc *d_sc = ar.take<sc>(NS), *d_part = ar.take<sc>(nblk);
    BPG_CHECK_HIP(pin.h2d(d_yinvtbl, yinvtbl, sizeof yinvtbl, s));
    BPG_CHECK_HIP(pin.h2d(d_uk, uk.data(), (lg + 1) * 32, s));
    BPG_CHECK_HIP(pin.h2d(d_ukinv, ukinv.data(), (lg + 1) * 32, s));
    BPG_CHECK_HIP(pin.flush(s));
    verify_args A{x.v, P.a.v, P.b.v, u.v, (uint32_t)n, (uint32_t)np, (uint32_t)lg, (uint32_t)n1};
    // scalar row order = table order: [ g (G) | h (H) | B | B~ | dyn ]
    hipLaunchKernelGGL(k_verify_scalars, dim3((unsigned)nblk), dim3(256), 0, s, A, d_W, d_yinvtbl, d_uk, d_ukinv, d_sc, d_sc + np,
                       d_part);
    BPG_CHECK_HIP(hipGetLastError());
    std::vector<sc> h_part(nblk), wV(m + 1), h_one(n_one + 1);
    BPG_CHECK_HIP(pin.d2h(h_part.data(), d_part, nblk * 32, s));
    if (m) BPG_CHECK_HIP(pin.d2h(wV.data(), d_W + 3 * n, m * 32, s));
    if (n_one) BPG_CHECK_HIP(pin.d2h(h_one.data(), d_one, n_one * 32, s));
    BPG_CHECK_HIP(pin.flush(s));
    BPG_CHECK_HIP(lane_sync(lane));
    pin.collect();
    Scalar delta(0), wc(0);
    for (size_t b = 0; b < nblk; b++) delta += Scalar(h_part[b]);
    for (size_t b = 0; b < n_one; b++) wc -= Scalar(h_one[b]);
    const Scalar xx = x * x, rxx = r * xx, xxx = x * xx, ab = P.a * P.b;
    std::vector<sc> tail;
    tail.reserve(2 + ndyn);
    tail.push_back((w * (P.t_x - ab) + r * (xx * (wc + delta) - P.t_x)).v);      // B
    tail.push_back((-P.e_blinding - r * P.t_x_blinding).v);                       // B~
    tail.push_back(x.v); tail.push_back(xx.v); tail.push_back(xxx.v);             // A_I1 A_O1 S1
    tail.push_back((u * x).v); tail.push_back((u * xx).v); tail.push_back((u * xxx).v);   // A_I2 A_O2 S2
    for (size_t j = 0; j < m; j++) tail.push_back((Scalar(wV[j]) * rxx).v);       // V
    tail.push_back((r * x).v); tail.push_back((rxx * x).v); tail.push_back((rxx * xx).v);
    tail.push_back((rxx * xxx).v); tail.push_back((rxx * xx * xx).v);             // T_1 T_3 T_4 T_5 T_6
    for (size_t k = 0; k < lg; k++) tail.push_back(u_sq[k]);                      // L
    for (size_t k = 0; k < lg; k++) tail.push_back(u_inv_sq[k]);                  // R
    out.d_sc = d_sc;
    out.np = np;
    out.tail = std::move(tail);
    out.dyn = std::move(dyn);
    return BPG_OK;
}

// generators: fixed-base MSM over the shifted table; per-proof points: variable-base MSM.
// d_gh = [g[0..ng) | h[0..nh)] device scalars; tail = scalars of [B, B~ | dyn points].
static int verify_msm(bpg_lane *lane, Gens *gens, const sc *d_gh, size_t ng, size_t nh, sc *d_tail, const std::vector<sc> &tail,
                      const std::vector<uint8_t> &dyn) {
    bpg_ctx *ctx = gens->ctx;
    hipStream_t s = lane->stream;
    const size_t ndyn = dyn.size() / 32;
    int rc;
    if ((rc = grow((void **)&lane->d_dyn, &lane->dyn_cap, (ndyn + 1) * sizeof(aniels_rec)))) return rc;
    uint32_t bad = 0;
    const uint32_t *pending_bad = nullptr;      // decode and tail upload share one wait
    if ((rc = table_write_compressed(ctx, lane->d_dyn, dyn.data(), ndyn, &bad, &pending_bad))) return rc;
    if (bad) return BPG_EVERIFY;       // a point failed to decompress (dalek: optional_multiscalar_mul -> None)
    {
        PinArena pin(lane);            // (prepare's transfers are complete: the staging starts over)
        pin.reserve(tail.size() * 32 + 4096);
        BPG_CHECK_HIP(pin.h2d(d_tail, tail.data(), tail.size() * 32, s));
        BPG_CHECK_HIP(pin.flush(s));
    }
    BPG_CHECK_HIP(lane_sync(lane));
    if (pending_bad && *pending_bad) return BPG_EVERIFY;
    // d_gh = [g[0..ng) | h[0..nh)]: each rank takes the generators it holds; rank 0 adds the rest
    msm_seg gh[2] = {gen_seg(gens, 0, 0, ng, 0), gen_seg(gens, 1, 0, nh, ng)};
    msm_seg rest[2] = {{gens->table->d_table + gens->off_B(), 2}, {lane->d_dyn, ndyn}};
    bpg_msm_job *job_f = nullptr, *job_d = nullptr;
    if ((rc = msm_submit(ctx, gh, 2, d_gh, gh[0].len + gh[1].len, 1, false, &job_f, msm_fixed_base{gens->fb_c, gens->fb_stride()},
                         ng + nh)))
        return rc;
    // From here on job_f's kernels read the lane's arena (d_gh): whatever happens to the second job, the first is
    // waited on before this function returns and the caller reuses or frees those buffers.
    uint8_t parts[256], out[32];
    int rc_d = BPG_OK;
    if (gens->rank == 0) rc_d = msm_submit(ctx, rest, 2, d_tail, 2 + ndyn, 1, false, &job_d);
    rc = msm_wait(job_f, parts, true);
    if (job_d) {
        const int rc2 = msm_wait(job_d, parts + 128, true);      // reads d_tail and the lane's point scratch
        if (!rc) rc = rc2;
    }
    if (rc_d) return rc_d;
    if (rc) return rc;
    if (!job_d) {
        memset(parts + 128, 0, 128);          // identity: (0, 1, 1, 0)
        parts[128 + 32] = 1;
        parts[128 + 64] = 1;
    }
    if (gens->world == 1) {
        if ((rc = combine_partials(parts, 2, out))) return rc;
    } else {
        std::vector<uint8_t> all(256 * (size_t)gens->world);
        if (gens->exchange(gens->exchange_user, parts, 256, all.data())) {
            set_last_error_msg("partial-sum exchange failed");
            return BPG_EDEVICE;
        }
        if ((rc = combine_partials(all.data(), 2 * (size_t)gens->world, out))) return rc;
    }
    uint8_t acc = 0;
    for (int i = 0; i < 32; i++) acc |= out[i];
    return acc ? BPG_EVERIFY : BPG_OK;
}

int Verifier::verify(const R1CSProof &P, Gens *gens, rng_fill_fn rng_fn, void *rng_user) {
    bpg_lane *lane = lane_of(gens->ctx);
    if (!lane) return BPG_EDEVICE;
    VerifyRow row;
    int rc = prepare(P, gens, rng_fn, rng_user, row);
    if (rc) return rc;
    return verify_msm(lane, gens, row.d_sc, row.np, row.np, row.d_sc + 2 * row.np, row.tail, row.dyn);
}

// One combined multiscalar multiplication for k proofs: sum_j rho_j * (verification equation j).  rho_j is the second
// draw (after r) of verifier j's TranscriptRng, which has absorbed proof j's whole transcript and 32 bytes of the
// caller's RNG: a weight is never zero and never independent of the proof it weighs, whatever the caller's RNG
// returns.  BPG_OK iff every proof verifies (up to 2^-250 with a real external RNG).
int verify_batch(Verifier *const *vs, const R1CSProof *proofs, size_t k, Gens *gens, rng_fill_fn rng_fn, void *rng_user) {
    if (!k) return BPG_OK;
    bpg_lane *lane = lane_of(gens->ctx);
    if (!lane) return BPG_EDEVICE;
    hipStream_t s = lane->stream;
    const size_t cap = gens->capacity;
    // the accumulators live in the lane (grown on demand, kept): no hipMalloc / hipFree per call -- hipFree is a
    // device-wide synchronisation under every other thread that proves or verifies on this GPU
    int rcg;
    if ((rcg = grow(&lane->d_vacc, &lane->vacc_cap, 2 * cap * 32))) return rcg;
    sc *d_acc = (sc *)lane->d_vacc;
    BPG_CHECK_HIP(hipMemsetAsync(d_acc, 0, 2 * cap * 32, s));
    std::vector<sc> tail(2);
    std::vector<uint8_t> dyn;
    size_t np_max = 0;
    Scalar accB(0), accBb(0);
    for (size_t j = 0; j < k; j++) {
        VerifyRow row;
        int rc = vs[j]->prepare(proofs[j], gens, rng_fn, rng_user, row);
        if (rc) return rc;
        const Scalar rho(row.rho);
        if (rho.is_zero()) {                                  // probability 2^-252 with a working RNG: refuse rather than skip a proof
            set_last_error_msg("verify_batch: zero batching weight (broken external RNG?)");
            return BPG_EINVAL;
        }
        hipLaunchKernelGGL(k_axpy, dim3(grid256(row.np)), dim3(256), 0, s, (uint32_t)row.np, rho.v, row.d_sc, d_acc);
        hipLaunchKernelGGL(k_axpy, dim3(grid256(row.np)), dim3(256), 0, s, (uint32_t)row.np, rho.v, row.d_sc + row.np, d_acc + cap);
        BPG_CHECK_HIP(hipGetLastError());
        accB += rho * Scalar(row.tail[0]);
        accBb += rho * Scalar(row.tail[1]);
        for (size_t i = 2; i < row.tail.size(); i++) tail.push_back((rho * Scalar(row.tail[i])).v);
        dyn.insert(dyn.end(), row.dyn.begin(), row.dyn.end());
        if (row.np > np_max) np_max = row.np;
        BPG_CHECK_HIP(lane_sync(lane));     // the next prepare() reuses the lane's arena
    }
    tail[0] = accB.v;
    tail[1] = accBb.v;
    if ((rcg = grow(&lane->d_vtail, &lane->vtail_cap, tail.size() * 32))) return rcg;
    sc *d_tail = (sc *)lane->d_vtail;
    // [g[0..cap) | h[0..np_max)]: g scalars beyond np_max are zero and produce no bucket entries
    return verify_msm(lane, gens, d_acc, cap, np_max, d_tail, tail, dyn);
}

}  // namespace bpg
