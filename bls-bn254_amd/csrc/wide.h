// wide.h -- TWO WAVES PER TUPLE arithmetic for the latency-bound tails of the path ("wave-per-tuple" in the documents: one
// workgroup of 128 lanes per tuple; one wave until the end of round 3).
//
// The verification kernels run one lane per tuple: right for throughput, but a launch with few tuples (the ONE final
// exponentiation of an aggregate verify, a single pairing, a small batch) is then the latency of one lane's serial chain --
// 4.5 ms for the hard part of the final exponentiation (pairings.rs:117-178), whatever the batch size below 65 536.
// Here the 128 lanes of a workgroup share one tuple, and the unit of work of a lane is ONE Fp output: an Fp12 value lives in LDS
// as six Fp2 coefficients; an Fp12 product is 72 independent double products (the two components of the 36 Fp2 products a_i b_j,
// one per lane) followed by twelve linear combinations (w^6 = xi wraps the high half: per-lane coefficients, fp_lc_rt); a
// Granger-Scott cyclotomic squaring (pairings.rs:68-115) is eighteen independent Fp products (the two components of nine Fp2
// squarings) followed by twelve linear combinations.  Same field values as the serial code (tower.h), hence the same bytes after
// the final canonicalisation; chains ~15 x shorter than a lane's.
//
// Everything is written as PHASES: within a phase every lane works on its own operands and writes its own result; a
// phase boundary is a workgroup barrier on the device and the end of a loop over the lane ids in tests/hostsim, which runs
// this same code under the interval checker.
#pragma once
#include "quad.h"

namespace bn {

constexpr uint32_t WIDE_LANES = 128;
#if defined(__HIP_DEVICE_COMPILE__)
#define BN_WIDE_PHASE(lane, ...) { const uint32_t lane = threadIdx.x; { __VA_ARGS__ } __syncthreads(); }
#define BN_WIDE_NOINLINE __attribute__((noinline))
#else
#define BN_WIDE_PHASE(lane, ...) { for (uint32_t lane = 0; lane < WIDE_LANES; ++lane) { __VA_ARGS__ } }
#define BN_WIDE_NOINLINE
#endif

// LDS region of one tuple (dwords): the product area (36 Fp2) and WIDE_VALUES Fp12 values of 108 limbs, each stored as
// its six tower slots c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (18 limbs per slot).
enum : uint32_t { WIDE_PA_LIMBS = 36 * 18, WIDE_VALUES = 20, WIDE_LDS_DWORDS = WIDE_PA_LIMBS + 108 * WIDE_VALUES };
// value ids of the hard part
enum : uint32_t { WV_R = 0, WV_T = 1, WV_A = 2, WV_B = 3, WV_C = 4, WV_B2 = 5, WV_D2 = 6, WV_X = 7, WV_E = 8, WV_D = 9, WV_TMP = 10, WV_SLOT0 = 10 };   // chain slots 10..19 (TMP shares slot 0's place outside a chain)
// The primitives below are REAL functions on the device (one instance each per kernel), so the region's base reaches them as an
// argument: it is carried as an LDS-typed pointer, or every access would be a flat (generic address space) load or store
// instead of ds_read / ds_write.  wide_local(W, w) re-bases a reference into the same LDS array (the kernels' coordinate and
// line areas: Ws{lds, 1, byte offset, false}) the same way; on the host it is the reference itself.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) int32_t wide_lds_i32;
struct Wide { wide_lds_i32* lds; __device__ explicit Wide(int32_t* p) : lds((wide_lds_i32*)p) {} };
BN_INL int32_t* wide_base(const Wide& w) { return (int32_t*)w.lds; }
BN_INL Ws wide_local(const Wide& w, const Ws& r) { return Ws{wide_base(w), 1, r.lane4, false}; }
#else
struct Wide { int32_t* lds; BN_HD explicit Wide(int32_t* p) : lds(p) {} };
BN_INL int32_t* wide_base(const Wide& w) { return w.lds; }
BN_INL Ws wide_local(const Wide&, const Ws& r) { return r; }
#endif

BN_INL Ws wide_val(const Wide& w, uint32_t v, uint32_t slot) { return Ws{wide_base(w), 1, (WIDE_PA_LIMBS + 108u * v + 18u * slot) * 4u, false}; }
BN_INL Ws wide_pa(const Wide& w, uint32_t idx) { return Ws{wide_base(w), 1, 18u * idx * 4u, false}; }
// tower slot of the coefficient of w^k (f = sum_k f_k w^k, w^2 = v, w^6 = xi)
BN_INL uint32_t wide_slot_of_w(uint32_t k) { return (k & 1u) ? 3u + (k >> 1) : (k >> 1); }
// tower slot of z_i in the Granger-Scott labelling (z0 = c0.c0, z4 = c0.c1, z3 = c0.c2, z2 = c1.c0, z1 = c1.c1, z5 = c1.c2)
BN_INL uint32_t wide_slot_of_z(uint32_t i) { return i == 0 ? 0u : i == 1 ? 4u : i == 2 ? 3u : i == 3 ? 2u : i == 4 ? 1u : 5u; }

// canonical limbs of ONE Fp from a table (the Fp-granular form of fp2_load_limbs, pairing.h)
BN_INL Fp fp_load_limbs_ws(const Ws& w) {
  Fp c;
  BN_UNROLL for (int k = 0; k < NL; ++k) c.l[k] = ws_load(w, k);
  BN_TRK(set_trk(c, 0, 1, 0, 0.006, 1);)
  return c;
}
// dst = a * b.  Phase 1 (wide_mul_products): lane 2 (6 i + j) + h forms component h of a_i b_j (coefficients of w^i, w^j): ONE
// double product, x0 y0 + (-x1) y1 or x0 y1 + x1 y0.  Phase 2 (wide_mul_sums): lane 2 k + h forms component h of the coefficient
// of w^k: the products with i + j = k and xi times those with i + j = k + 6, as ONE linear combination with per-lane
// coefficients (fp_lc_rt): with t_i = a_i b_(k-i mod 6), direct for i <= k and wrapped (times xi = 9 + u) for i > k,
//   out.c0 = sum_i A_i t_i.c0 - W_i t_i.c1,   out.c1 = sum_i A_i t_i.c1 + W_i t_i.c0,   A_i = 1 | 9, W_i = 0 | 1.
// dst may be a or b.
BN_FUNC void wide_mul_sums(const Wide& W, uint32_t dst) {
  BN_WIDE_PHASE(lane,
    if (lane < 12u) {
      const uint32_t k = lane >> 1, h = lane & 1u;
      Fp x[12];
      int32_t kk[12];
      BN_UNROLL for (uint32_t i = 0; i < 6u; ++i) {
        const Fp2 t = fp2_load_mem(wide_pa(W, 6u * i + (k + 6u - i) % 6u));
        const bool wrap = i > k;
        x[i] = t.c0; x[6 + i] = t.c1;
        kk[i] = h ? (wrap ? 1 : 0) : (wrap ? 9 : 1);
        kk[6 + i] = h ? (wrap ? 9 : 1) : (wrap ? -1 : 0);
      }
      fp_store_mem(ws_at_lane(wide_val(W, dst, wide_slot_of_w(k)), 9u * h), fp_lc_rt<12>(x, kk));
    })
}
// the operands of product lane `pl` (< 72) of a * b as a double product a0 b0 + c0 d0, and where its result goes
BN_INL Ws wide_product_operands(const Wide& W, uint32_t pl, uint32_t a, uint32_t b, Fp& a0, Fp& b0, Fp& c0, Fp& d0) {
  const uint32_t pidx = pl >> 1, h = pl & 1u, i = pidx / 6u, j = pidx - 6u * i;
  const Ws xw = wide_val(W, a, wide_slot_of_w(i)), yw = wide_val(W, b, wide_slot_of_w(j));
  const Fp x1 = fp_load_mem(ws_at(xw, 9));
  a0 = fp_load_mem(xw);
  b0 = fp_load_mem(ws_at_lane(yw, 9u * h));                    // h = 0: y0, h = 1: y1
  c0 = fp_select(h != 0u, x1, fp_neg(x1));                     // h = 0: -x1, h = 1: x1
  d0 = fp_load_mem(ws_at_lane(yw, 9u * (1u - h)));             // h = 0: y1, h = 1: y0
  return ws_at_lane(wide_pa(W, pidx), 9u * h);
}
BN_FUNC void wide_mul_products(const Wide& W, uint32_t a, uint32_t b) {
  BN_WIDE_PHASE(lane,
    if (lane < 72u) {
      Fp a0, b0, c0, d0;
      const Ws out = wide_product_operands(W, lane, a, b, a0, b0, c0, d0);
      fp_store_mem(out, fp_dot2(a0, b0, c0, d0));
    })
}
BN_FUNC void wide_mul(const Wide& W, uint32_t dst, uint32_t a, uint32_t b) { wide_mul_products(W, a, b); wide_mul_sums(W, dst); }
// v = v^2 for v in the cyclotomic subgroup (same values as fp12_cyclotomic_sqr, tower.h).  Phase 1: lane 2 (3 p + q) + h forms
// component h of the square of a, b or a + b (q = 0, 1, 2) of the pair p: (z0, z1), (z2, z3), (z4, z5) -- one Fp product,
// (u0 + u1)(u0 - u1) or 2 u0 u1.  Phase 2: lane 2 o + h forms component h of r_o in the slot of z_o from the pair's ta = a^2,
// tb = b^2, s = (a + b)^2 and the old z_o: cyc_c0 / cyc_c1 / the z2 formula of fp12_cyclotomic_sqr (tower.h) written out per
// component as seven-term combinations:
//   r0, r4, r3 (cyc_c0):  c0 = 3 ta0 + 27 tb0 - 3 tb1 - 2 z0,   c1 = 3 ta1 + 3 tb0 + 27 tb1 - 2 z1
//   r1, r5     (cyc_c1):  c = 3 s - 3 ta - 3 tb + 2 z   (per component)
//   r2 = 3 xi (s - ta - tb) + 2 z2:  c0 = 27 (s0 - ta0 - tb0) - 3 (s1 - ta1 - tb1) + 2 z0,  c1 = 3 (s0 - ta0 - tb0) + 27 (s1 - ta1 - tb1) + 2 z1
BN_FUNC void wide_cyc_sqr(const Wide& W, uint32_t v) {
  BN_WIDE_PHASE(lane,
    if (lane < 18u) {
      const uint32_t sq = lane >> 1, h = lane & 1u, p = sq / 3u, q = sq - 3u * p;
      const Fp2 a = fp2_load_mem(wide_val(W, v, wide_slot_of_z(2u * p))), b = fp2_load_mem(wide_val(W, v, wide_slot_of_z(2u * p + 1u)));
      const Fp2 s = fp2_norm(fp2_add(a, b));
      const Fp2 u = fp2_select(q == 0u, a, fp2_select(q == 1u, b, s));
      // fp2_sqr (tower.h) one component per lane: (u0 + u1)(u0 - u1) | (2 u0) u1
      const Fp m0 = fp_select(h != 0u, fp_dbl(u.c0), fp_add(u.c0, u.c1)), m1 = fp_select(h != 0u, u.c1, fp_sub(u.c0, u.c1));
      fp_store_mem(ws_at_lane(wide_pa(W, sq), 9u * h), fp_mul(m0, m1));
    })
  BN_WIDE_PHASE(lane,
    if (lane < 12u) {
      const uint32_t o = lane >> 1, h = lane & 1u;
      const uint32_t p = (o < 2u) ? 0u : (o < 4u) ? 2u : 1u;       // r0, r1 <- (z0, z1); r2, r3 <- (z4, z5); r4, r5 <- (z2, z3)
      const Fp2 ta = fp2_load_mem(wide_pa(W, 3u * p)), tb = fp2_load_mem(wide_pa(W, 3u * p + 1u)), s = fp2_load_mem(wide_pa(W, 3u * p + 2u));
      const Ws zw = ws_at_lane(wide_val(W, v, wide_slot_of_z(o)), 9u * h);
      const Fp x[7] = {ta.c0, ta.c1, tb.c0, tb.c1, s.c0, s.c1, fp_load_mem(zw)};
      const int kind = o == 2u ? 2 : (o == 1u || o == 5u) ? 1 : 0;
      const int32_t tab[3][2][7] = {{{3, 0, 27, -3, 0, 0, -2}, {0, 3, 3, 27, 0, 0, -2}},
                                    {{-3, 0, -3, 0, 3, 0, 2}, {0, -3, 0, -3, 0, 3, 2}},
                                    {{-27, 3, -27, 3, 27, -3, 2}, {-3, -27, -3, -27, 3, 27, 2}}};
      int32_t kk[7];
      BN_UNROLL for (int t = 0; t < 7; ++t) {
        const int32_t c0 = h ? tab[0][1][t] : tab[0][0][t], c1 = h ? tab[1][1][t] : tab[1][0][t], c2 = h ? tab[2][1][t] : tab[2][0][t];
        kk[t] = kind == 0 ? c0 : kind == 1 ? c1 : c2;
      }
      fp_store_mem(zw, fp_lc_rt<7>(x, kk));
    })
}
// dst = conj(src) = (c0, -c1); dst = src (plain copy) when neg is false
BN_FUNC void wide_conj_or_copy(const Wide& W, uint32_t dst, uint32_t src, bool neg) {
  BN_WIDE_PHASE(lane,
    if (lane < 6u) {
      const Fp2 x = fp2_load_mem(wide_val(W, src, lane));
      fp2_store_mem(wide_val(W, dst, lane), fp2_select(neg && lane >= 3u, fp2_norm(fp2_neg(x)), x));
    })
}
// dst = Frobenius^K(src): the coefficient of w^k becomes conj^K(.) * xi^(k (p^K - 1) / 6)
template <int K>
BN_FUNC void wide_frob(const Wide& W, uint32_t dst, uint32_t src) {
  BN_WIDE_PHASE(lane,
    if (lane < 6u) {
      const uint32_t k = lane, g = k ? k - 1u : 0u;
      Fp2 c = fp2_load_mem(wide_val(W, src, wide_slot_of_w(k)));
      if (K & 1) c = fp2_norm(fp2_conj(c));
      Fp2 m;
      if (K == 1) m = fp2_mul(c, fp2_from_limbs(bnc::GAMMA1[g]));
      else if (K == 2) m = fp2_mul_fp(c, fp_from_limbs(bnc::GAMMA2[g]));
      else m = fp2_mul(c, fp2_from_limbs(bnc::GAMMA3[g]));
      fp2_store_mem(wide_val(W, dst, wide_slot_of_w(k)), fp2_select(k != 0u, m, c));
    })
}

// ---- Miller loops over prepared keys (no point arithmetic): per loop digit  R <- R^2, then R <- R * L  with the line value L
// assembled by five lanes from the key's table and the tuple's coordinates.  L is stored as a full Fp12 value (zero slots
// where the line has no coefficient) and multiplied in by the general wide_mul: at one wave per tuple the chain length counts,
// not the 6 of 36 products that are spent on zeros.  One pair: L = c0 py + c1 px w + c2 w^3 (ell, pairing.h) from the raw triple;
// two pairs: L = (T0 ysY + T1 Z) + T2 xsX v + (T3 xsZ + T4 X) v^2 + [(T5 ysX + T6 xsY) + (T7 ysZ + T8 Y) v] w from the pair table
// entry T0..T8 of ell_pair_expanded (pairing.h) and the nine coordinate values X, Y, Z, xs X, ys Y, xs Z, ys Z, ys X, xs Y.
// The product phase of dst = a * b (wide_mul's phase 1) with up to TWO line values evaluated by otherwise idle lanes in the SAME
// instruction stream: a component of an Fp2 product and a component of a line coefficient are both ONE double product,
//   product lane:  x0 y0 + (-x1) y1  |  x0 y1 + x1 y0            line lane (q, s, h):  TA_h sa + TB_h sb
// so every lane first fetches ITS four operands (the only divergent part: a few loads) and then all run the same fp_dot2.
// Lanes 0..71 form the products (skipped when `prod` is false: the two final lines have no squaring), lane 72 + 12 q + 2 s + h
// component h of the coefficient of tower slot s of line q, written to value WV_L + q.  Line operands:
//   mode 0: pair table entry (162 limbs, T0..T8) at ln (+ 162 q), cw = the nine coordinate values
//           L = (T0 ysY + T1 Z) + T2 xsX v + (T3 xsZ + T4 X) v^2 + [(T5 ysX + T6 xsY) + (T7 ysZ + T8 Y) v] w
//   mode 1: raw triple (54 limbs) at ln (+ 54 q), cw = (px, py):  L = c0 py + c1 px w + c2 w^3
//   mode 2: the triple parked in LDS by fp2_store_mem at ln, cw = (px, py); one line
// wide_mul_sums(dst) (wide_mul's phase 2) completes the product.
enum : uint32_t { WV_L = 1, WV_L2 = 2 };                            // the line values (the Miller loops do not use WV_T, WV_A)
BN_FUNC void wide_mul_products_lines(const Wide& W, bool prod, uint32_t a, uint32_t b, uint32_t nlines, const Ws& ln_in, const Ws& cw_in, int mode) {
  // This is a real function: what the kernel knows about its references is restated here -- the key's table is buffer-addressed,
  // dense (stride 1) and one base for the whole workgroup; the coordinates and a parked line live in the tuple's LDS region.
  const Ws tab = {ws_uniform(ln_in).base, 1, ln_in.lane4, true}, lnl = wide_local(W, ln_in), cw = wide_local(W, cw_in);
  BN_WIDE_PHASE(lane,
    const bool is_prod = prod && lane < 72u, is_line = lane >= 72u && lane < 72u + 12u * nlines;
    if (is_prod || is_line) {
      Fp a0, b0, c0, d0;
      bool used = true;
      Ws out;
      if (is_prod) {
        out = wide_product_operands(W, lane, a, b, a0, b0, c0, d0);
      } else {
        const uint32_t q = (lane - 72u) / 12u, r = lane - 72u - 12u * q, sl = r >> 1, h = r & 1u;      // line, tower slot, component
        if (mode == 0) {
          const uint32_t ia = sl == 0u ? 0u : sl == 1u ? 2u : sl == 2u ? 3u : sl == 3u ? 5u : 7u, ib = sl == 0u ? 1u : sl == 1u ? 2u : sl == 2u ? 4u : sl == 3u ? 6u : 8u;
          const uint32_t ca = sl == 0u ? 4u : sl == 1u ? 3u : sl == 2u ? 5u : sl == 3u ? 7u : 6u, cb = sl == 0u ? 2u : sl == 2u ? 0u : sl == 3u ? 8u : 1u;
          a0 = fp_load_limbs_lazy(ws_at_lane(tab, 162u * q + 18u * ia + 9u * h)); c0 = fp_load_limbs_lazy(ws_at_lane(tab, 162u * q + 18u * ib + 9u * h));   // pair-table entries are lazy (pairing.h)
          b0 = fp_load_mem(ws_at_lane(cw, 9u * ca));
          d0 = fp_select(sl == 1u, fp_zero(), fp_load_mem(ws_at_lane(cw, 9u * cb)));      // T2 xsX stands alone
          used = sl < 5u;                                                                    // slot 5 (v^2 w) has no coefficient
        } else {
          const uint32_t off = (sl == 3u ? 18u : sl == 4u ? 36u : 0u) + 9u * h;             // tower slots of w^0, w^1, w^3: c0 py, c1 px, c2
          a0 = mode == 2 ? fp_load_mem(ws_at_lane(lnl, off)) : fp_load_limbs_ws(ws_at_lane(tab, 54u * q + off));
          c0 = a0;
          b0 = fp_select(sl == 4u, fp_one(), fp_load_mem(ws_at_lane(cw, sl == 0u ? 9u : 0u)));
          d0 = fp_zero();
          used = sl == 0u || sl == 3u || sl == 4u;
        }
        out = ws_at_lane(wide_val(W, WV_L + q, sl), 9u * h);
      }
      const Fp r = fp_dot2(a0, b0, c0, d0);
      fp_store_mem(out, fp_select(used, r, fp_zero()));
    })
}
enum : uint32_t { WOP_MUL = 0, WOP_SQR = 1, WOP_CONJ = 2, WOP_COPY = 3, WOP_FROB1 = 4, WOP_FROB2 = 5, WOP_FROB3 = 6, WOP_SUMS = 7 };
// one primitive; WOP_SQR squares `dst` in place `a` times.  A real function: every primitive is instantiated once.
BN_HD BN_WIDE_NOINLINE inline void wide_exec(const Wide& W, uint32_t op, uint32_t dst, uint32_t a, uint32_t b) {
  switch (op) {
    case WOP_MUL: wide_mul_products(W, a, b);                       // falls through to the sums: one copy of each phase per kernel
    case WOP_SUMS: wide_mul_sums(W, dst); break;
    case WOP_SQR: for (uint32_t q = 0; q < a; ++q) wide_cyc_sqr(W, dst); break;
    case WOP_CONJ: wide_conj_or_copy(W, dst, a, true); break;
    case WOP_COPY: wide_conj_or_copy(W, dst, a, false); break;
    case WOP_FROB1: wide_frob<1>(W, dst, a); break;
    case WOP_FROB2: wide_frob<2>(W, dst, a); break;
    default: wide_frob<3>(W, dst, a); break;
  }
}
// the loop of miller_loop_1prepared / miller_loop_prepared (pairing.h); pair = false: raw table of 54-limb lines and pt = (px, py);
// pair = true: pair table of 162-limb entries and pt = the nine coordinate values.  Result in WV_R.
BN_HD BN_WIDE_NOINLINE inline void wide_mul_products_lines_call(const Wide& W, bool prod, uint32_t a, uint32_t b, uint32_t nlines, const Ws& ln, const Ws& cw, int mode) {
  wide_mul_products_lines(W, prod, a, b, nlines, ln, cw, mode);      // a real function, like wide_exec: instantiated once per kernel
}
BN_HD inline void wide_miller_prepared(const Wide& W, const Ws& table, const Ws& pt, bool pair) {
  BN_WIDE_PHASE(lane,
    if (lane < 6u) fp2_store_mem(wide_val(W, WV_R, lane), lane == 0u ? fp2_one() : fp2_zero());
  )
  const size_t per = pair ? 162 : 54;
  int ti = 0;
  for (int j = bnc::ATE_NAF_LEN - 2; j >= -1; --j) {                // j = -1: the two final lines (no squaring)
    const uint32_t lines = j >= 0 ? (ate_naf_digit(j) != 0 ? 2u : 1u) : 2u;
    // R^2's 36 products and this digit's line values in one phase: the lines depend on the table and the tuple only
    wide_mul_products_lines_call(W, j >= 0, WV_R, WV_R, lines, ws_at(table, per * (size_t)ti), pt, pair ? 0 : 1);
    ti += (int)lines;
    if (j >= 0) wide_exec(W, WOP_SUMS, WV_R, 0, 0);
    for (uint32_t q = 0; q < lines; ++q) wide_exec(W, WOP_MUL, WV_R, WV_R, WV_L + q);
  }
}

// The one-pair loop with a VARIABLE G2 point (miller_loop_1, pairing.h; pairings.rs:760-857): lanes 0 .. 3 keep the running point T
// and compute each line as a QUAD (quad.h: the independent Fp2 products of every level of doubling_step / addition_step side by
// side, DPP fetches inside the quad; r02 ran them on lane 0 alone: about half of the kernel's time), lane 0 parks the triple in
// LDS at lnw (54 limbs), five lanes evaluate it at P = pt and the whole wave multiplies it in.  q: the validated point (every
// lane holds it).  Result in WV_R.  On the host the quad's phase runs as four threads (tri_host_run4, tests/hostsim).
#if !defined(__HIP_DEVICE_COMPILE__)
void tri_host_run4(void (*fn)(void*, uint32_t), void* arg);       // hostsim: run fn(arg, role) on the four lanes of a quad
struct WideLineStep { G2J* T; const Fp2 *ax, *ay; const Ws* lnw; bool dbl; };
inline void wide_line_step_host(void* a, uint32_t role) {
  WideLineStep* w = (WideLineStep*)a;
  G2J& T = w->T[role];
  const Line l = fp2_norm_line(w->dbl ? quad_doubling_step(T, role) : quad_addition_step(T, *w->ax, *w->ay, role));
  if (role == 0u) line_store(*w->lnw, l);
}
#endif
BN_HD inline void wide_miller_1(const Wide& W, const G2A& q, const Ws& pt, const Ws& lnw) {
  BN_WIDE_PHASE(lane,
    if (lane < 6u) fp2_store_mem(wide_val(W, WV_R, lane), lane == 0u ? fp2_one() : fp2_zero());
  )
  const Fp2 nqy = fp2_norm(fp2_neg(q.y));
  const Fp2 g2 = fp2_const(bnc::GAMMA1[1]), g3 = fp2_const(bnc::GAMMA1[2]);
  const Fp2 q1x = fp2_mul(fp2_norm(fp2_conj(q.x)), g2), q1y = fp2_mul(fp2_norm(fp2_conj(q.y)), g3);        // pi(Q)
  const Fp2 q2x = fp2_mul(fp2_norm(fp2_conj(q1x)), g2);
  const Fp2 q2y = fp2_norm(fp2_neg(fp2_mul(fp2_norm(fp2_conj(q1y)), g3)));                                  // -pi^2(Q)
#if defined(__HIP_DEVICE_COMPILE__)
  G2J T = {q.x, q.y, fp2_one()};
#define BN_WIDE_LINE_STEP(DBL, AX, AY) { const uint32_t lane_ = threadIdx.x; \
    if (lane_ < 4u) { const Line l_ = fp2_norm_line((DBL) ? quad_doubling_step(T, lane_) : quad_addition_step(T, AX, AY, lane_)); if (lane_ == 0u) line_store(lnw, l_); } \
    __syncthreads(); }
#else
  G2J Tq[4] = {{q.x, q.y, fp2_one()}, {q.x, q.y, fp2_one()}, {q.x, q.y, fp2_one()}, {q.x, q.y, fp2_one()}};
#define BN_WIDE_LINE_STEP(DBL, AX, AY) { const Fp2 ax_ = (AX), ay_ = (AY); WideLineStep w_ = {Tq, &ax_, &ay_, &lnw, (DBL)}; tri_host_run4(wide_line_step_host, &w_); }
#endif
  for (int j = bnc::ATE_NAF_LEN - 2; j >= -2; --j) {                // j = -1, -2: the two Frobenius additions
    if (j >= 0) {
      BN_WIDE_LINE_STEP(true, q.x, q.y)                               // T <- 2T and its tangent: independent of R
      wide_mul_products_lines_call(W, true, WV_R, WV_R, 1u, lnw, pt, 2);      // R^2's products and the tangent's value in one phase
      wide_exec(W, WOP_SUMS, WV_R, 0, 0);
      wide_exec(W, WOP_MUL, WV_R, WV_R, WV_L);
    }
    const int d = j >= 0 ? ate_naf_digit(j) : 1;
    if (d != 0) {
      const Fp2 ax = j >= 0 ? q.x : j == -1 ? q1x : q2x, ay = j >= 0 ? (d > 0 ? q.y : nqy) : j == -1 ? q1y : q2y;
      BN_WIDE_LINE_STEP(false, ax, ay)
      wide_mul_products_lines_call(W, false, WV_R, WV_R, 1u, lnw, pt, 2);
      wide_exec(W, WOP_MUL, WV_R, WV_R, WV_L);
    }
  }
#undef BN_WIDE_LINE_STEP
}
// out = in^x by the signed chain of cyclotomic_exp_x_chain (pairing.h, BN_X_CHAIN): 62 squarings + 13 products; R is the accumulator
BN_HD inline void wide_exp_x(const Wide& W, uint32_t out, uint32_t in) {
  const ChainOp prog[BN_X_CHAIN_LEN] = BN_X_CHAIN;
  wide_exec(W, WOP_COPY, WV_SLOT0, in, 0);
  wide_exec(W, WOP_COPY, WV_R, in, 0);
  for (int k = 0; k < BN_X_CHAIN_LEN; ++k) {
    const ChainOp op = prog[k];
    if (op.load >= 0) wide_exec(W, WOP_COPY, WV_R, WV_SLOT0 + (uint32_t)op.load, 0);
    if (op.sq > 0) wide_exec(W, WOP_SQR, WV_R, (uint32_t)op.sq, 0);
    if (op.mul >= 0) wide_exec(W, WOP_MUL, WV_R, WV_R, WV_SLOT0 + (uint32_t)op.mul);
    if (op.store >= 0) wide_exec(W, WOP_COPY, WV_SLOT0 + (uint32_t)op.store, WV_R, 0);
    if (op.cstore >= 0) wide_exec(W, WOP_CONJ, WV_SLOT0 + (uint32_t)op.cstore, WV_R, 0);
  }
  wide_exec(W, WOP_COPY, out, WV_R, 0);
}
// The hard part of the final exponentiation (fe_h1 / fe_h2 / fe_h3 of pairing.h around three t -> t^x): WV_T holds
// t = f^((p^6-1)(p^2+1)); the result is left in WV_R.
BN_HD inline void wide_fe_hard(const Wide& W) {
  wide_exp_x(W, WV_X, WV_T);                                   // x0 = t^x
  wide_exec(W, WOP_CONJ, WV_A, WV_X, 0); wide_exec(W, WOP_SQR, WV_A, 1, 0);          // a = t^-2x
  wide_exec(W, WOP_COPY, WV_B, WV_A, 0); wide_exec(W, WOP_SQR, WV_B, 1, 0);          // t^-4x
  wide_exec(W, WOP_MUL, WV_B, WV_A, WV_B);                                           // b = t^-6x
  wide_exp_x(W, WV_X, WV_B);                                   // x1 = b^x = t^(-6x^2)
  wide_exec(W, WOP_CONJ, WV_C, WV_X, 0);                                             // c = t^(6x^2)
  wide_exec(W, WOP_CONJ, WV_B2, WV_B, 0); wide_exec(W, WOP_MUL, WV_B2, WV_C, WV_B2); // b2 = c conj(b)
  wide_exec(W, WOP_COPY, WV_D2, WV_C, 0); wide_exec(W, WOP_SQR, WV_D2, 1, 0);        // d2 = c^2
  wide_exp_x(W, WV_X, WV_D2);                                  // x2 = d2^x = t^(12x^3)
  wide_exec(W, WOP_MUL, WV_E, WV_B2, WV_X);                                          // e
  wide_exec(W, WOP_MUL, WV_D, WV_A, WV_E);                                           // d
  wide_exec(W, WOP_MUL, WV_R, WV_C, WV_E); wide_exec(W, WOP_MUL, WV_R, WV_T, WV_R);  // l0 = t (c e)
  wide_exec(W, WOP_FROB1, WV_TMP, WV_D, 0); wide_exec(W, WOP_MUL, WV_R, WV_R, WV_TMP);
  wide_exec(W, WOP_FROB2, WV_TMP, WV_E, 0); wide_exec(W, WOP_MUL, WV_R, WV_R, WV_TMP);
  wide_exec(W, WOP_CONJ, WV_TMP, WV_T, 0); wide_exec(W, WOP_MUL, WV_TMP, WV_TMP, WV_D);   // l3 = conj(t) d
  wide_exec(W, WOP_FROB3, WV_TMP, WV_TMP, 0); wide_exec(W, WOP_MUL, WV_R, WV_R, WV_TMP);
}

}  // namespace bn
