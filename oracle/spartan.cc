// ORACLE (test infrastructure only). CPU restatement of the libspartan prover/verifier. See spartan.h.
#include "spartan.h"

#include <chrono>
#include <cassert>

namespace orc {

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define ORC_CHECK(c)                                                                  \
  do {                                                                                \
    if (!(c)) { fprintf(stderr, "oracle: check failed %s:%d: %s\n", __FILE__, __LINE__, #c); abort(); } \
  } while (0)

// ============================== commitments.rs ==============================
MultiCommitGens MultiCommitGens::make(size_t n, const char* label) {  // commitments.rs:15-33
  Shake256 shake;
  shake.absorb((const uint8_t*)label, strlen(label));
  uint8_t bp[32];
  pt_basepoint_compressed(bp);
  shake.absorb(bp, 32);
  MultiCommitGens g;
  g.n = n;
  g.G.resize(n);
  uint8_t ub[64];
  for (size_t i = 0; i < n + 1; i++) {
    shake.squeeze(ub, 64);
    Pt p = pt_from_uniform_bytes(ub);
    if (i < n) g.G[i] = p; else g.h = p;
  }
  return g;
}
MultiCommitGens MultiCommitGens::scale(const Fq& s) const {  // :43-49
  MultiCommitGens g;
  g.n = n; g.h = h; g.G.resize(n);
  for (size_t i = 0; i < n; i++) g.G[i] = pt_mul(s, G[i]);
  return g;
}
void MultiCommitGens::split_at(size_t mid, MultiCommitGens* a, MultiCommitGens* b) const {  // :51-69
  a->n = mid; a->G.assign(G.begin(), G.begin() + mid); a->h = h;
  b->n = n - mid; b->G.assign(G.begin() + mid, G.end()); b->h = h;
}
Pt commit_scalar(const Fq& x, const Fq& blind, const MultiCommitGens& g1) {  // :73-78
  ORC_CHECK(g1.n == 1);
  Fq s[2] = {x, blind};
  Pt p[2] = {g1.G[0], g1.h};
  return pt_msm(s, p, 2);
}
Pt commit_vec(const Fq* v, size_t n, const Fq& blind, const MultiCommitGens& gn) {  // :80-92
  ORC_CHECK(gn.n == n);
  return pt_add(pt_msm(v, gn.G.data(), n), pt_mul(blind, gn.h));
}

DotProductProofGens DotProductProofGens::make(size_t n, const char* label) {  // nizk/mod.rs:415-418
  DotProductProofGens g;
  g.n = n;
  MultiCommitGens::make(n + 1, label).split_at(n, &g.gens_n, &g.gens_1);
  return g;
}
PolyCommitmentGens PolyCommitmentGens::make(size_t num_vars, const char* label) {  // dense_mlpoly.rs:31-35
  size_t right = num_vars - num_vars / 2;
  return PolyCommitmentGens{DotProductProofGens::make(pow2(right), label)};
}

// ============================== dense_mlpoly.rs ==============================
FqVec eq_evals(const FqVec& r) {  // :68-84
  size_t ell = r.size();
  FqVec evals(pow2(ell), fq_one());
  size_t size = 1;
  for (size_t j = 0; j < ell; j++) {
    size *= 2;
    for (size_t i = size - 1;; i -= 2) {
      Fq scalar = evals[i / 2];
      evals[i] = scalar * r[j];
      evals[i - 1] = scalar - evals[i];
      if (i == 1) break;
    }
  }
  return evals;
}
Fq eq_evaluate(const FqVec& r, const FqVec& rx) {  // :60-66
  ORC_CHECK(r.size() == rx.size());
  Fq acc = fq_one();
  for (size_t i = 0; i < rx.size(); i++) acc = acc * (r[i] * rx[i] + (fq_one() - r[i]) * (fq_one() - rx[i]));
  return acc;
}
void eq_factored_evals(const FqVec& r, FqVec* L, FqVec* R) {  // :90-98
  size_t left = r.size() / 2;
  *L = eq_evals(FqVec(r.begin(), r.begin() + left));
  *R = eq_evals(FqVec(r.begin() + left, r.end()));
}
void DensePoly::bound_poly_var_top(const Fq& r) {  // :215-223
  size_t n = len / 2;
  for (size_t i = 0; i < n; i++) Z[i] = Z[i] + r * (Z[i + n] - Z[i]);
  Z.resize(n);
  num_vars -= 1;
  len = n;
}
void DensePoly::bound_poly_var_bot(const Fq& r) {  // :225-233
  size_t n = len / 2;
  for (size_t i = 0; i < n; i++) Z[i] = Z[2 * i] + r * (Z[2 * i + 1] - Z[2 * i]);
  Z.resize(n);
  num_vars -= 1;
  len = n;
}
FqVec DensePoly::bound(const FqVec& L) const {  // :206-213
  size_t left = num_vars / 2, right = num_vars - left;
  size_t Ls = pow2(left), Rs = pow2(right);
  FqVec out(Rs, fq_zero());
  for (size_t j = 0; j < Ls; j++)
    for (size_t i = 0; i < Rs; i++) out[i] += L[j] * Z[j * Rs + i];
  return out;
}
static Fq dotproduct(const FqVec& a, const FqVec& b) {  // nizk/mod.rs:306-309, 435-438 ; bullet.rs:233-243
  ORC_CHECK(a.size() == b.size());
  Fq s = fq_zero();
  for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
  return s;
}
Fq DensePoly::evaluate(const FqVec& r) const {  // :236-242
  ORC_CHECK(r.size() == num_vars);
  FqVec chis = eq_evals(r);
  ORC_CHECK(chis.size() == Z.size());
  return dotproduct(Z, chis);
}
void DensePoly::split(size_t idx, DensePoly* a, DensePoly* b) const {  // :140-146
  ORC_CHECK(idx < len);
  *a = DensePoly(FqVec(Z.begin(), Z.begin() + idx));
  *b = DensePoly(FqVec(Z.begin() + idx, Z.begin() + 2 * idx));
}
void DensePoly::extend(const DensePoly& o) {  // :248-257
  ORC_CHECK(Z.size() == len && o.Z.size() == len);
  Z.insert(Z.end(), o.Z.begin(), o.Z.end());
  num_vars += 1;
  len *= 2;
}
DensePoly DensePoly::merge(const std::vector<const DensePoly*>& polys) {  // :259-272
  FqVec Z;
  for (auto p : polys) Z.insert(Z.end(), p->Z.begin(), p->Z.end());
  Z.resize(next_pow2(Z.size()), fq_zero());
  return DensePoly(std::move(Z));
}
DensePoly DensePoly::from_usize(const std::vector<size_t>& z) {  // :274-280
  FqVec Z(z.size());
  for (size_t i = 0; i < z.size(); i++) Z[i] = fq_from_u64((uint64_t)z[i]);
  return DensePoly(std::move(Z));
}
PolyCommitment poly_commit(const DensePoly& p, const PolyCommitmentGens& gens, RandomTape* tape, FqVec* blinds_out) {
  // dense_mlpoly.rs:179-204 + commit_inner :148-177
  size_t n = p.Z.size(), ell = p.num_vars;
  ORC_CHECK(n == pow2(ell));
  size_t Ls = pow2(ell / 2), Rs = pow2(ell - ell / 2);
  FqVec blinds = tape ? tape->random_vector("poly_blinds", Ls) : FqVec(Ls, fq_zero());
  PolyCommitment c;
  c.C.resize(Ls);
  const MultiCommitGens& g = gens.gens.gens_n;
  ORC_CHECK(g.n == Rs);
#pragma omp parallel for schedule(dynamic)
  for (size_t i = 0; i < Ls; i++) c.C[i] = compress(commit_vec(&p.Z[Rs * i], Rs, blinds[i], g));
  if (blinds_out) *blinds_out = blinds;
  return c;
}
void append_poly_commitment(Transcript& t, const char* label, const PolyCommitment& c) {  // :292-300
  t.append_message(label, "poly_commitment_begin");
  for (auto& pt : c.C) t.append_point("poly_commitment_share", pt.data());
  t.append_message(label, "poly_commitment_end");
}

// ============================== unipoly.rs ==============================
UniPoly UniPoly::from_evals(const FqVec& e) {  // :23-55
  ORC_CHECK(e.size() == 3 || e.size() == 4);
  UniPoly u;
  Fq two_inv = fq_invert(fq_from_u64(2));
  if (e.size() == 3) {
    Fq c = e[0];
    Fq a = two_inv * (e[2] - e[1] - e[1] + c);
    Fq b = e[1] - c - a;
    u.coeffs = {c, b, a};
  } else {
    Fq six_inv = fq_invert(fq_from_u64(6));
    Fq d = e[0];
    Fq a = six_inv * (e[3] - e[2] - e[2] - e[2] + e[1] + e[1] + e[1] - e[0]);
    Fq b = two_inv * (e[0] + e[0] - e[1] - e[1] - e[1] - e[1] - e[1] + e[2] + e[2] + e[2] + e[2] - e[3]);
    Fq c = e[1] - d - a - b;
    u.coeffs = {d, c, b, a};
  }
  return u;
}
Fq UniPoly::evaluate(const Fq& r) const {  // :72-80
  Fq eval = coeffs[0], power = r;
  for (size_t i = 1; i < coeffs.size(); i++) { eval += power * coeffs[i]; power *= r; }
  return eval;
}
Fq UniPoly::eval_at_one() const { Fq s = fq_zero(); for (auto& c : coeffs) s += c; return s; }
FqVec UniPoly::compress() const {  // :82-88
  FqVec c;
  c.push_back(coeffs[0]);
  c.insert(c.end(), coeffs.begin() + 2, coeffs.end());
  return c;
}
UniPoly UniPoly::decompress(const FqVec& c, const Fq& hint) {  // :96-110
  Fq lin = hint - c[0] - c[0];
  for (size_t i = 1; i < c.size(); i++) lin -= c[i];
  UniPoly u;
  u.coeffs = {c[0], lin};
  u.coeffs.insert(u.coeffs.end(), c.begin() + 1, c.end());
  return u;
}
void UniPoly::append_to_transcript(Transcript& t, const char* label) const {  // :112-120
  t.append_message(label, "UniPoly_begin");
  for (auto& c : coeffs) t.append_scalar("coeff", c);
  t.append_message(label, "UniPoly_end");
}

// ============================== nizk/mod.rs ==============================
static KnowledgeProof knowledge_prove(const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& x, const Fq& r, CP* C_out) {
  // :27-52
  t.append_protocol_name("knowledge proof");
  Fq t1 = tape.random_scalar("t1"), t2 = tape.random_scalar("t2");
  CP C = compress(commit_scalar(x, r, g));
  t.append_point("C", C.data());
  CP alpha = compress(commit_scalar(t1, t2, g));
  t.append_point("alpha", alpha.data());
  Fq c = t.challenge_scalar("c");
  *C_out = C;
  return KnowledgeProof{alpha, x * c + t1, r * c + t2};
}
static bool knowledge_verify(const KnowledgeProof& p, const MultiCommitGens& g, Transcript& t, const CP& C) {  // :54-74
  t.append_protocol_name("knowledge proof");
  t.append_point("C", C.data());
  t.append_point("alpha", p.alpha.data());
  Fq c = t.challenge_scalar("c");
  CP lhs = compress(commit_scalar(p.z1, p.z2, g));
  CP rhs = compress(pt_add(pt_mul(c, decompress(C)), decompress(p.alpha)));
  return lhs == rhs;
}
static EqualityProof equality_prove(const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& v1, const Fq& s1,
                                    const Fq& v2, const Fq& s2, CP* C1o, CP* C2o) {  // :88-116
  t.append_protocol_name("equality proof");
  Fq r = tape.random_scalar("r");
  CP C1 = compress(commit_scalar(v1, s1, g));
  t.append_point("C1", C1.data());
  CP C2 = compress(commit_scalar(v2, s2, g));
  t.append_point("C2", C2.data());
  CP alpha = compress(pt_mul(r, g.h));
  t.append_point("alpha", alpha.data());
  Fq c = t.challenge_scalar("c");
  if (C1o) *C1o = C1;
  if (C2o) *C2o = C2;
  return EqualityProof{alpha, c * (s1 - s2) + r};
}
static bool equality_verify(const EqualityProof& p, const MultiCommitGens& g, Transcript& t, const CP& C1, const CP& C2) {  // :118-143
  t.append_protocol_name("equality proof");
  t.append_point("C1", C1.data());
  t.append_point("C2", C2.data());
  t.append_point("alpha", p.alpha.data());
  Fq c = t.challenge_scalar("c");
  Pt C = pt_sub(decompress(C1), decompress(C2));
  CP rhs = compress(pt_add(pt_mul(c, C), decompress(p.alpha)));
  CP lhs = compress(pt_mul(p.z, g.h));
  return lhs == rhs;
}
static ProductProof product_prove(const MultiCommitGens& g, Transcript& t, RandomTape& tape, const Fq& x, const Fq& rX,
                                  const Fq& y, const Fq& rY, const Fq& z, const Fq& rZ, CP* Xo, CP* Yo, CP* Zo) {  // :159-227
  t.append_protocol_name("product proof");
  Fq b1 = tape.random_scalar("b1"), b2 = tape.random_scalar("b2"), b3 = tape.random_scalar("b3");
  Fq b4 = tape.random_scalar("b4"), b5 = tape.random_scalar("b5");
  CP X = compress(commit_scalar(x, rX, g)); t.append_point("X", X.data());
  CP Y = compress(commit_scalar(y, rY, g)); t.append_point("Y", Y.data());
  CP Z = compress(commit_scalar(z, rZ, g)); t.append_point("Z", Z.data());
  CP alpha = compress(commit_scalar(b1, b2, g)); t.append_point("alpha", alpha.data());
  CP beta = compress(commit_scalar(b3, b4, g)); t.append_point("beta", beta.data());
  MultiCommitGens gX; gX.n = 1; gX.G = {decompress(X)}; gX.h = g.h;
  CP delta = compress(commit_scalar(b3, b5, gX)); t.append_point("delta", delta.data());
  Fq c = t.challenge_scalar("c");
  ProductProof p;
  p.alpha = alpha; p.beta = beta; p.delta = delta;
  p.z[0] = b1 + c * x; p.z[1] = b2 + c * rX; p.z[2] = b3 + c * y; p.z[3] = b4 + c * rY; p.z[4] = b5 + c * (rZ - rX * y);
  *Xo = X; *Yo = Y; *Zo = Z;
  return p;
}
static bool product_check_eq(const CP& P, const CP& X, const Fq& c, const MultiCommitGens& g, const Fq& z1, const Fq& z2) {  // :229-241
  CP lhs = compress(pt_add(decompress(P), pt_mul(c, decompress(X))));
  CP rhs = compress(commit_scalar(z1, z2, g));
  return lhs == rhs;
}
static bool product_verify(const ProductProof& p, const MultiCommitGens& g, Transcript& t, const CP& X, const CP& Y, const CP& Z) {  // :243-289
  t.append_protocol_name("product proof");
  t.append_point("X", X.data()); t.append_point("Y", Y.data()); t.append_point("Z", Z.data());
  t.append_point("alpha", p.alpha.data()); t.append_point("beta", p.beta.data()); t.append_point("delta", p.delta.data());
  Fq c = t.challenge_scalar("c");
  MultiCommitGens gX; gX.n = 1; gX.G = {decompress(X)}; gX.h = g.h;
  return product_check_eq(p.alpha, X, c, g, p.z[0], p.z[1]) && product_check_eq(p.beta, Y, c, g, p.z[2], p.z[3]) &&
         product_check_eq(p.delta, Z, c, gX, p.z[2], p.z[4]);
}
static DotProductProof dotproduct_prove(const MultiCommitGens& g1, const MultiCommitGens& gn, Transcript& t, RandomTape& tape,
                                        const FqVec& x, const Fq& blind_x, const FqVec& a, const Fq& y, const Fq& blind_y) {
  // :311-370
  t.append_protocol_name("dot product proof");
  size_t n = x.size();
  ORC_CHECK(a.size() == n && gn.n == n && g1.n == 1);
  FqVec d = tape.random_vector("d_vec", n);
  Fq r_delta = tape.random_scalar("r_delta"), r_beta = tape.random_scalar("r_beta");
  CP Cx = compress(commit_vec(x.data(), n, blind_x, gn)); t.append_point("Cx", Cx.data());
  CP Cy = compress(commit_scalar(y, blind_y, g1)); t.append_point("Cy", Cy.data());
  t.append_scalars("a", a);
  CP delta = compress(commit_vec(d.data(), n, r_delta, gn)); t.append_point("delta", delta.data());
  Fq dp = dotproduct(a, d);
  CP beta = compress(commit_scalar(dp, r_beta, g1)); t.append_point("beta", beta.data());
  Fq c = t.challenge_scalar("c");
  DotProductProof p;
  p.delta = delta; p.beta = beta; p.z.resize(n);
  for (size_t i = 0; i < n; i++) p.z[i] = c * x[i] + d[i];
  p.z_delta = c * blind_x + r_delta;
  p.z_beta = c * blind_y + r_beta;
  return p;
}
static bool dotproduct_verify(const DotProductProof& p, const MultiCommitGens& g1, const MultiCommitGens& gn, Transcript& t,
                              const FqVec& a, const CP& Cx, const CP& Cy) {  // :372-405
  t.append_protocol_name("dot product proof");
  t.append_point("Cx", Cx.data()); t.append_point("Cy", Cy.data());
  t.append_scalars("a", a);
  t.append_point("delta", p.delta.data()); t.append_point("beta", p.beta.data());
  Fq c = t.challenge_scalar("c");
  bool ok = pt_eq(pt_add(pt_mul(c, decompress(Cx)), decompress(p.delta)), commit_vec(p.z.data(), p.z.size(), p.z_delta, gn));
  Fq dza = dotproduct(p.z, a);
  ok = ok && pt_eq(pt_add(pt_mul(c, decompress(Cy)), decompress(p.beta)), commit_scalar(dza, p.z_beta, g1));
  return ok;
}

// ============================== nizk/bullet.rs ==============================
struct BulletOut { BulletReductionProof proof; Fq a_hat, b_hat, blind_hat; Pt g_hat; };
static BulletOut bullet_prove(Transcript& t, const Pt& Q, const std::vector<Pt>& G_vec, const Pt& H, const FqVec& a_vec,
                              const FqVec& b_vec, const Fq& blind, const std::vector<std::pair<Fq, Fq>>& blinds_vec) {  // :32-132
  std::vector<Pt> G = G_vec;
  FqVec a = a_vec, b = b_vec;
  size_t n = G.size();
  ORC_CHECK((n & (n - 1)) == 0 && a.size() == n && b.size() == n && blinds_vec.size() == log_2(n));
  BulletOut o;
  Fq blind_final = blind;
  size_t round = 0;
  while (n != 1) {
    n /= 2;
    Fq c_L = fq_zero(), c_R = fq_zero();
    for (size_t i = 0; i < n; i++) { c_L += a[i] * b[n + i]; c_R += a[n + i] * b[i]; }
    const Fq& blind_L = blinds_vec[round].first;
    const Fq& blind_R = blinds_vec[round].second;
    FqVec sc(n + 2);
    std::vector<Pt> pts(n + 2);
    for (size_t i = 0; i < n; i++) { sc[i] = a[i]; pts[i] = G[n + i]; }
    sc[n] = c_L; pts[n] = Q; sc[n + 1] = blind_L; pts[n + 1] = H;
    Pt L = pt_msm(sc.data(), pts.data(), n + 2);
    for (size_t i = 0; i < n; i++) { sc[i] = a[n + i]; pts[i] = G[i]; }
    sc[n] = c_R; sc[n + 1] = blind_R;
    Pt R = pt_msm(sc.data(), pts.data(), n + 2);
    CP Lc = compress(L), Rc = compress(R);
    t.append_point("L", Lc.data());
    t.append_point("R", Rc.data());
    Fq u = t.challenge_scalar("u");
    Fq u_inv = fq_invert(u);
    for (size_t i = 0; i < n; i++) {
      a[i] = a[i] * u + u_inv * a[n + i];
      b[i] = b[i] * u_inv + u * b[n + i];
      Fq s2[2] = {u_inv, u};
      Pt p2[2] = {G[i], G[n + i]};
      G[i] = pt_msm(s2, p2, 2);
    }
    blind_final = blind_final + blind_L * u * u + blind_R * u_inv * u_inv;
    o.proof.L_vec.push_back(Lc);
    o.proof.R_vec.push_back(Rc);
    round++;
  }
  o.a_hat = a[0]; o.b_hat = b[0]; o.g_hat = G[0]; o.blind_hat = blind_final;
  return o;
}
static bool bullet_verify(const BulletReductionProof& p, size_t n, const FqVec& a, Transcript& t, const Pt& Gamma,
                          const std::vector<Pt>& G, Pt* g_hat, Pt* Gamma_hat, Fq* a_hat) {  // :137-231
  size_t lg_n = p.L_vec.size();
  if (lg_n >= 32 || n != ((size_t)1 << lg_n)) return false;
  FqVec ch(lg_n);
  for (size_t i = 0; i < lg_n; i++) {
    t.append_point("L", p.L_vec[i].data());
    t.append_point("R", p.R_vec[i].data());
    ch[i] = t.challenge_scalar("u");
  }
  FqVec ch_inv = ch;
  Fq allinv = fq_batch_invert(ch_inv.data(), lg_n);
  for (size_t i = 0; i < lg_n; i++) { ch[i] = fq_sqr(ch[i]); ch_inv[i] = fq_sqr(ch_inv[i]); }
  FqVec s(n);
  s[0] = allinv;
  for (size_t i = 1; i < n; i++) {
    size_t lg_i = 0;
    while (((size_t)2 << lg_i) <= i) lg_i++;
    size_t k = (size_t)1 << lg_i;
    s[i] = s[i - k] * ch[(lg_n - 1) - lg_i];
  }
  *g_hat = pt_msm(s.data(), G.data(), n);
  *a_hat = dotproduct(a, s);
  FqVec sc;
  std::vector<Pt> pts;
  for (size_t i = 0; i < lg_n; i++) { sc.push_back(ch[i]); pts.push_back(decompress(p.L_vec[i])); }
  for (size_t i = 0; i < lg_n; i++) { sc.push_back(ch_inv[i]); pts.push_back(decompress(p.R_vec[i])); }
  sc.push_back(fq_one()); pts.push_back(Gamma);
  *Gamma_hat = pt_msm(sc.data(), pts.data(), sc.size());
  return true;
}

// DotProductProofLog  nizk/mod.rs:440-525 / 527-578
static DotProductProofLog dotproductlog_prove(const DotProductProofGens& gens, Transcript& t, RandomTape& tape, const FqVec& x,
                                              const Fq& blind_x, const FqVec& a, const Fq& y, const Fq& blind_y, CP* Cxo, CP* Cyo) {
  t.append_protocol_name("dot product proof (log)");
  size_t n = x.size();
  ORC_CHECK(a.size() == n && gens.n == n);
  Fq d = tape.random_scalar("d");
  Fq r_delta = tape.random_scalar("r_delta");
  Fq r_beta = tape.random_scalar("r_delta");  // sic: label quirk at nizk/mod.rs:459
  size_t lg_n = log_2(n);
  FqVec v1 = tape.random_vector("blinds_vec_1", lg_n), v2 = tape.random_vector("blinds_vec_2", lg_n);
  std::vector<std::pair<Fq, Fq>> blinds_vec(lg_n);
  for (size_t i = 0; i < lg_n; i++) blinds_vec[i] = {v1[i], v2[i]};
  CP Cx = compress(commit_vec(x.data(), n, blind_x, gens.gens_n)); t.append_point("Cx", Cx.data());
  CP Cy = compress(commit_scalar(y, blind_y, gens.gens_1)); t.append_point("Cy", Cy.data());
  t.append_scalars("a", a);
  Fq r = t.challenge_scalar("r");
  MultiCommitGens g1s = gens.gens_1.scale(r);
  Fq blind_Gamma = blind_x + r * blind_y;
  BulletOut bo = bullet_prove(t, g1s.G[0], gens.gens_n.G, gens.gens_n.h, x, a, blind_Gamma, blinds_vec);
  Fq y_hat = bo.a_hat * bo.b_hat;
  MultiCommitGens ghat; ghat.n = 1; ghat.G = {bo.g_hat}; ghat.h = gens.gens_1.h;
  CP delta = compress(commit_scalar(d, r_delta, ghat)); t.append_point("delta", delta.data());
  CP beta = compress(commit_scalar(d, r_beta, g1s)); t.append_point("beta", beta.data());
  Fq c = t.challenge_scalar("c");
  DotProductProofLog p;
  p.bullet = bo.proof; p.delta = delta; p.beta = beta;
  p.z1 = d + c * y_hat;
  p.z2 = bo.b_hat * (c * bo.blind_hat + r_beta) + r_delta;
  if (Cxo) *Cxo = Cx;
  if (Cyo) *Cyo = Cy;
  return p;
}
static bool dotproductlog_verify(const DotProductProofLog& p, size_t n, const DotProductProofGens& gens, Transcript& t,
                                 const FqVec& a, const CP& Cx, const CP& Cy) {
  ORC_CHECK(gens.n == n && a.size() == n);
  t.append_protocol_name("dot product proof (log)");
  t.append_point("Cx", Cx.data()); t.append_point("Cy", Cy.data());
  t.append_scalars("a", a);
  Fq r = t.challenge_scalar("r");
  MultiCommitGens g1s = gens.gens_1.scale(r);
  Pt Gamma = pt_add(decompress(Cx), pt_mul(r, decompress(Cy)));
  Pt g_hat, Gamma_hat; Fq a_hat;
  if (!bullet_verify(p.bullet, n, a, t, Gamma, gens.gens_n.G, &g_hat, &Gamma_hat, &a_hat)) return false;
  t.append_point("delta", p.delta.data()); t.append_point("beta", p.beta.data());
  Fq c = t.challenge_scalar("c");
  Pt lhs = pt_add(pt_mul(a_hat, pt_add(pt_mul(c, Gamma_hat), decompress(p.beta))), decompress(p.delta));
  Pt rhs = pt_add(pt_mul(p.z1, pt_add(g_hat, pt_mul(a_hat, g1s.G[0]))), pt_mul(p.z2, g1s.h));
  return compress(lhs) == compress(rhs);
}

// PolyEvalProof  dense_mlpoly.rs:312-365 / 367-403
static PolyEvalProof polyeval_prove(const DensePoly& poly, const FqVec* blinds_opt, const FqVec& r, const Fq& Zr,
                                    const Fq* blind_Zr_opt, const PolyCommitmentGens& gens, Transcript& t, RandomTape& tape, CP* C_Zr) {
  t.append_protocol_name("polynomial evaluation proof");
  ORC_CHECK(poly.num_vars == r.size());
  size_t Ls = pow2(r.size() / 2), Rs = pow2(r.size() - r.size() / 2);
  FqVec zero_blinds(Ls, fq_zero());
  const FqVec& blinds = blinds_opt ? *blinds_opt : zero_blinds;
  ORC_CHECK(blinds.size() == Ls);
  Fq blind_Zr = blind_Zr_opt ? *blind_Zr_opt : fq_zero();
  FqVec L, R;
  eq_factored_evals(r, &L, &R);
  ORC_CHECK(L.size() == Ls && R.size() == Rs);
  FqVec LZ = poly.bound(L);
  Fq LZ_blind = fq_zero();
  for (size_t i = 0; i < Ls; i++) LZ_blind += blinds[i] * L[i];
  PolyEvalProof p;
  p.proof = dotproductlog_prove(gens.gens, t, tape, LZ, LZ_blind, R, Zr, blind_Zr, nullptr, C_Zr);
  return p;
}
static bool polyeval_verify(const PolyEvalProof& p, const PolyCommitmentGens& gens, Transcript& t, const FqVec& r, const CP& C_Zr,
                            const PolyCommitment& comm) {
  t.append_protocol_name("polynomial evaluation proof");
  FqVec L, R;
  eq_factored_evals(r, &L, &R);
  std::vector<Pt> C(comm.C.size());
  for (size_t i = 0; i < C.size(); i++) C[i] = decompress(comm.C[i]);
  ORC_CHECK(L.size() == C.size());
  CP C_LZ = compress(pt_msm(L.data(), C.data(), L.size()));
  return dotproductlog_verify(p.proof, R.size(), gens.gens, t, R, C_LZ, C_Zr);
}
static bool polyeval_verify_plain(const PolyEvalProof& p, const PolyCommitmentGens& gens, Transcript& t, const FqVec& r,
                                  const Fq& Zr, const PolyCommitment& comm) {  // :391-403
  CP C_Zr = compress(commit_scalar(Zr, fq_zero(), gens.gens.gens_1));
  return polyeval_verify(p, gens, t, r, C_Zr, comm);
}

// ============================== sumcheck.rs ==============================
// ZK rounds share this tail (sumcheck.rs:491-583 and :681-772): eval commit, weights, target, DotProductProof.
static void zk_round_tail(const UniPoly& poly, const Fq& r_j, size_t j, const Fq& blind_claim, const FqVec& blinds_poly,
                          const FqVec& blinds_evals, Fq& claim_per_round, CP& comm_claim_per_round, const MultiCommitGens& g1,
                          const MultiCommitGens& gn, Transcript& t, RandomTape& tape, ZKSumcheckProof& out) {
  Fq eval = poly.evaluate(r_j);
  CP comm_eval = compress(commit_scalar(eval, blinds_evals[j], g1));
  t.append_point("comm_claim_per_round", comm_claim_per_round.data());
  t.append_point("comm_eval", comm_eval.data());
  FqVec w = t.challenge_vector("combine_two_claims_to_one", 2);
  Fq target = w[0] * claim_per_round + w[1] * eval;
  Fq s2[2] = {w[0], w[1]};
  Pt p2[2] = {decompress(comm_claim_per_round), decompress(comm_eval)};
  CP comm_target = compress(pt_msm(s2, p2, 2));
  const Fq& blind_sc = (j == 0) ? blind_claim : blinds_evals[j - 1];
  Fq blind = w[0] * blind_sc + w[1] * blinds_evals[j];
  ORC_CHECK(compress(commit_scalar(target, blind, g1)) == comm_target);  // sumcheck.rs:531,722
  size_t d1 = poly.degree() + 1;
  FqVec a(d1);
  Fq pw = fq_one();
  for (size_t i = 0; i < d1; i++) {
    Fq a_sc = (i == 0) ? fq_one() + fq_one() : fq_one();
    a[i] = w[0] * a_sc + w[1] * pw;
    pw = pw * r_j;
  }
  DotProductProof proof = dotproduct_prove(g1, gn, t, tape, poly.coeffs, blinds_poly[j], a, target, blind);
  claim_per_round = eval;
  comm_claim_per_round = comm_eval;
  out.proofs.push_back(proof);
  out.comm_evals.push_back(comm_eval);
}

// prove_cubic_with_additive_term  sumcheck.rs:588-776 ; comb = A*(B*C - D)  (r1csproof.rs:87-91)
static ZKSumcheckProof zk_prove_cubic_additive(const Fq& claim, const Fq& blind_claim, size_t num_rounds, DensePoly& A, DensePoly& B,
                                               DensePoly& C, DensePoly& D, const MultiCommitGens& g1, const MultiCommitGens& gn,
                                               Transcript& t, RandomTape& tape, FqVec* r_out, Fq claims[4], Fq* blind_post) {
  FqVec blinds_poly = tape.random_vector("blinds_poly", num_rounds);
  FqVec blinds_evals = tape.random_vector("blinds_evals", num_rounds);
  Fq claim_per_round = claim;
  CP comm_claim = compress(commit_scalar(claim_per_round, blind_claim, g1));
  ZKSumcheckProof out;
  FqVec r;
  for (size_t j = 0; j < num_rounds; j++) {
    Fq e0 = fq_zero(), e2 = fq_zero(), e3 = fq_zero();
    size_t len = A.len / 2;
    for (size_t i = 0; i < len; i++) {
      e0 += A[i] * (B[i] * C[i] - D[i]);
      Fq a2 = A[len + i] + A[len + i] - A[i], b2 = B[len + i] + B[len + i] - B[i];
      Fq c2 = C[len + i] + C[len + i] - C[i], d2 = D[len + i] + D[len + i] - D[i];
      e2 += a2 * (b2 * c2 - d2);
      Fq a3 = a2 + A[len + i] - A[i], b3 = b2 + B[len + i] - B[i], c3 = c2 + C[len + i] - C[i], d3 = d2 + D[len + i] - D[i];
      e3 += a3 * (b3 * c3 - d3);
    }
    UniPoly poly = UniPoly::from_evals({e0, claim_per_round - e0, e2, e3});
    CP comm_poly = compress(commit_vec(poly.coeffs.data(), poly.coeffs.size(), blinds_poly[j], gn));
    t.append_point("comm_poly", comm_poly.data());
    out.comm_polys.push_back(comm_poly);
    Fq r_j = t.challenge_scalar("challenge_nextround");
    A.bound_poly_var_top(r_j); B.bound_poly_var_top(r_j); C.bound_poly_var_top(r_j); D.bound_poly_var_top(r_j);
    zk_round_tail(poly, r_j, j, blind_claim, blinds_poly, blinds_evals, claim_per_round, comm_claim, g1, gn, t, tape, out);
    r.push_back(r_j);
  }
  *r_out = r;
  claims[0] = A[0]; claims[1] = B[0]; claims[2] = C[0]; claims[3] = D[0];
  *blind_post = blinds_evals[num_rounds - 1];
  return out;
}
// prove_quad  sumcheck.rs:428-586 ; comb = A*B
static ZKSumcheckProof zk_prove_quad(const Fq& claim, const Fq& blind_claim, size_t num_rounds, DensePoly& A, DensePoly& B,
                                     const MultiCommitGens& g1, const MultiCommitGens& gn, Transcript& t, RandomTape& tape,
                                     FqVec* r_out, Fq claims[2], Fq* blind_post) {
  FqVec blinds_poly = tape.random_vector("blinds_poly", num_rounds);
  FqVec blinds_evals = tape.random_vector("blinds_evals", num_rounds);
  Fq claim_per_round = claim;
  CP comm_claim = compress(commit_scalar(claim_per_round, blind_claim, g1));
  ZKSumcheckProof out;
  FqVec r;
  for (size_t j = 0; j < num_rounds; j++) {
    Fq e0 = fq_zero(), e2 = fq_zero();
    size_t len = A.len / 2;
    for (size_t i = 0; i < len; i++) {
      e0 += A[i] * B[i];
      Fq a2 = A[len + i] + A[len + i] - A[i], b2 = B[len + i] + B[len + i] - B[i];
      e2 += a2 * b2;
    }
    UniPoly poly = UniPoly::from_evals({e0, claim_per_round - e0, e2});
    CP comm_poly = compress(commit_vec(poly.coeffs.data(), poly.coeffs.size(), blinds_poly[j], gn));
    t.append_point("comm_poly", comm_poly.data());
    out.comm_polys.push_back(comm_poly);
    Fq r_j = t.challenge_scalar("challenge_nextround");
    A.bound_poly_var_top(r_j); B.bound_poly_var_top(r_j);
    zk_round_tail(poly, r_j, j, blind_claim, blinds_poly, blinds_evals, claim_per_round, comm_claim, g1, gn, t, tape, out);
    r.push_back(r_j);
  }
  *r_out = r;
  claims[0] = A[0]; claims[1] = B[0];
  *blind_post = blinds_evals[num_rounds - 1];
  return out;
}
// ZKSumcheckInstanceProof::verify  sumcheck.rs:84-179
static bool zk_sumcheck_verify(const ZKSumcheckProof& p, const CP& comm_claim, size_t num_rounds, size_t degree_bound,
                               const MultiCommitGens& g1, const MultiCommitGens& gn, Transcript& t, CP* comm_last, FqVec* r_out) {
  if (gn.n != degree_bound + 1 || p.comm_polys.size() != num_rounds || p.comm_evals.size() != num_rounds) return false;
  FqVec r;
  for (size_t i = 0; i < num_rounds; i++) {
    t.append_point("comm_poly", p.comm_polys[i].data());
    Fq r_i = t.challenge_scalar("challenge_nextround");
    const CP& cc = (i == 0) ? comm_claim : p.comm_evals[i - 1];
    const CP& ce = p.comm_evals[i];
    t.append_point("comm_claim_per_round", cc.data());
    t.append_point("comm_eval", ce.data());
    FqVec w = t.challenge_vector("combine_two_claims_to_one", 2);
    Fq s2[2] = {w[0], w[1]};
    Pt p2[2] = {decompress(cc), decompress(ce)};
    CP comm_target = compress(pt_msm(s2, p2, 2));
    FqVec a(degree_bound + 1);
    Fq pw = fq_one();
    for (size_t k = 0; k < a.size(); k++) {
      Fq a_sc = (k == 0) ? fq_one() + fq_one() : fq_one();
      a[k] = w[0] * a_sc + w[1] * pw;
      pw = pw * r_i;
    }
    if (!dotproduct_verify(p.proofs[i], g1, gn, t, a, p.comm_polys[i], comm_target)) return false;
    r.push_back(r_i);
  }
  *comm_last = p.comm_evals.back();
  *r_out = r;
  return true;
}
// SumcheckInstanceProof::verify  sumcheck.rs:27-61
static bool sumcheck_verify(const SumcheckProof& p, const Fq& claim, size_t num_rounds, size_t degree_bound, Transcript& t, Fq* e_out,
                            FqVec* r_out) {
  Fq e = claim;
  FqVec r;
  if (p.compressed_polys.size() != num_rounds) return false;
  for (size_t i = 0; i < num_rounds; i++) {
    UniPoly poly = UniPoly::decompress(p.compressed_polys[i], e);
    if (poly.degree() != degree_bound) return false;
    if (poly.eval_at_zero() + poly.eval_at_one() != e) return false;
    poly.append_to_transcript(t, "poly");
    Fq r_i = t.challenge_scalar("challenge_nextround");
    r.push_back(r_i);
    e = poly.evaluate(r_i);
  }
  *e_out = e;
  *r_out = r;
  return true;
}
// prove_cubic_batched  sumcheck.rs:254-424 ; comb = A*B*C (product_tree.rs:283-286)
struct BatchedOut { SumcheckProof proof; FqVec r; FqVec prodA, prodB; Fq prodC; FqVec dotA, dotB, dotC; };
static void cubic_evals(const DensePoly& A, const DensePoly& B, const DensePoly& C, Fq* e0, Fq* e2, Fq* e3) {
  Fq s0 = fq_zero(), s2 = fq_zero(), s3 = fq_zero();
  size_t len = A.len / 2;
  for (size_t i = 0; i < len; i++) {
    s0 += A[i] * B[i] * C[i];
    Fq a2 = A[len + i] + A[len + i] - A[i], b2 = B[len + i] + B[len + i] - B[i], c2 = C[len + i] + C[len + i] - C[i];
    s2 += a2 * b2 * c2;
    Fq a3 = a2 + A[len + i] - A[i], b3 = b2 + B[len + i] - B[i], c3 = c2 + C[len + i] - C[i];
    s3 += a3 * b3 * c3;
  }
  *e0 = s0; *e2 = s2; *e3 = s3;
}
static BatchedOut prove_cubic_batched(const Fq& claim, size_t num_rounds, std::vector<DensePoly*>& Apar, std::vector<DensePoly*>& Bpar,
                                      DensePoly& Cpar, std::vector<DensePoly*>& Aseq, std::vector<DensePoly*>& Bseq,
                                      std::vector<DensePoly*>& Cseq, const FqVec& coeffs, Transcript& t) {
  BatchedOut o;
  Fq e = claim;
  for (size_t j = 0; j < num_rounds; j++) {
    std::vector<std::array<Fq, 3>> evals;
    for (size_t k = 0; k < Apar.size(); k++) {
      std::array<Fq, 3> ev;
      cubic_evals(*Apar[k], *Bpar[k], Cpar, &ev[0], &ev[1], &ev[2]);
      evals.push_back(ev);
    }
    for (size_t k = 0; k < Aseq.size(); k++) {
      std::array<Fq, 3> ev;
      cubic_evals(*Aseq[k], *Bseq[k], *Cseq[k], &ev[0], &ev[1], &ev[2]);
      evals.push_back(ev);
    }
    Fq c0 = fq_zero(), c2 = fq_zero(), c3 = fq_zero();
    for (size_t i = 0; i < evals.size(); i++) { c0 += evals[i][0] * coeffs[i]; c2 += evals[i][1] * coeffs[i]; c3 += evals[i][2] * coeffs[i]; }
    UniPoly poly = UniPoly::from_evals({c0, e - c0, c2, c3});
    poly.append_to_transcript(t, "poly");
    Fq r_j = t.challenge_scalar("challenge_nextround");
    o.r.push_back(r_j);
    for (size_t k = 0; k < Apar.size(); k++) { Apar[k]->bound_poly_var_top(r_j); Bpar[k]->bound_poly_var_top(r_j); }
    Cpar.bound_poly_var_top(r_j);
    for (size_t k = 0; k < Aseq.size(); k++) { Aseq[k]->bound_poly_var_top(r_j); Bseq[k]->bound_poly_var_top(r_j); Cseq[k]->bound_poly_var_top(r_j); }
    e = poly.evaluate(r_j);
    o.proof.compressed_polys.push_back(poly.compress());
  }
  for (size_t k = 0; k < Apar.size(); k++) { o.prodA.push_back((*Apar[k])[0]); o.prodB.push_back((*Bpar[k])[0]); }
  o.prodC = Cpar[0];
  for (size_t k = 0; k < Aseq.size(); k++) { o.dotA.push_back((*Aseq[k])[0]); o.dotB.push_back((*Bseq[k])[0]); o.dotC.push_back((*Cseq[k])[0]); }
  return o;
}

// ============================== r1csproof.rs ==============================
R1CSGens R1CSGens::make(const char* label, size_t, size_t num_vars) {  // :68-73, :48-60
  R1CSGens g;
  g.gens_pc = PolyCommitmentGens::make(log_2(num_vars), label);
  g.gens_sc.gens_1 = g.gens_pc.gens.gens_1;
  g.gens_sc.gens_3 = MultiCommitGens::make(3, label);
  g.gens_sc.gens_4 = MultiCommitGens::make(4, label);
  return g;
}

R1CSProof r1cs_prove(const R1CSShape& inst, const FqVec& vars_in, const FqVec& input, const R1CSGens& gens, Transcript& t,
                     RandomTape& tape, FqVec* rx_out, FqVec* ry_out, ProveTimes* tm) {  // :144-349
  double t0 = now_s();
  t.append_protocol_name("R1CS proof");
  ORC_CHECK(input.size() < vars_in.size());
  t.append_scalars("input", input);
  R1CSProof P;
  double tc = now_s();
  DensePoly poly_vars(vars_in);
  FqVec blinds_vars;
  P.comm_vars = poly_commit(poly_vars, gens.gens_pc, &tape, &blinds_vars);
  append_poly_commitment(t, "poly_commitment", P.comm_vars);
  if (tm) tm->polycommit = now_s() - tc;

  double t1 = now_s();
  size_t num_inputs = input.size(), num_vars = vars_in.size();
  FqVec z = vars_in;  // :177-185
  z.push_back(fq_one());
  z.insert(z.end(), input.begin(), input.end());
  z.resize(2 * num_vars, fq_zero());
  (void)num_inputs;
  size_t num_rounds_x = log_2(inst.num_cons), num_rounds_y = log_2(z.size());
  FqVec tau = t.challenge_vector("challenge_tau", num_rounds_x);
  DensePoly poly_tau(eq_evals(tau));
  DensePoly poly_Az(inst.A.multiply_vec(inst.num_cons, z.size(), z));
  DensePoly poly_Bz(inst.B.multiply_vec(inst.num_cons, z.size(), z));
  DensePoly poly_Cz(inst.C.multiply_vec(inst.num_cons, z.size(), z));
  FqVec rx;
  Fq claims1[4], blind_claim_postsc1;
  P.sc_proof_phase1 = zk_prove_cubic_additive(fq_zero(), fq_zero(), num_rounds_x, poly_tau, poly_Az, poly_Bz, poly_Cz,
                                              gens.gens_sc.gens_1, gens.gens_sc.gens_4, t, tape, &rx, claims1, &blind_claim_postsc1);
  ORC_CHECK(poly_tau.len == 1 && poly_Az.len == 1);
  if (tm) tm->sc_phase_one = now_s() - t1;

  Fq tau_claim = poly_tau[0], Az_claim = poly_Az[0], Bz_claim = poly_Bz[0], Cz_claim = poly_Cz[0];
  Fq Az_blind = tape.random_scalar("Az_blind"), Bz_blind = tape.random_scalar("Bz_blind");
  Fq Cz_blind = tape.random_scalar("Cz_blind"), prod_Az_Bz_blind = tape.random_scalar("prod_Az_Bz_blind");
  CP comm_Cz, comm_Az, comm_Bz, comm_prod;
  P.pok_Cz = knowledge_prove(gens.gens_sc.gens_1, t, tape, Cz_claim, Cz_blind, &comm_Cz);
  Fq prod = Az_claim * Bz_claim;
  P.proof_prod = product_prove(gens.gens_sc.gens_1, t, tape, Az_claim, Az_blind, Bz_claim, Bz_blind, prod, prod_Az_Bz_blind, &comm_Az,
                               &comm_Bz, &comm_prod);
  t.append_point("comm_Az_claim", comm_Az.data());
  t.append_point("comm_Bz_claim", comm_Bz.data());
  t.append_point("comm_Cz_claim", comm_Cz.data());
  t.append_point("comm_prod_Az_Bz_claims", comm_prod.data());
  P.claims_phase2[0] = comm_Az; P.claims_phase2[1] = comm_Bz; P.claims_phase2[2] = comm_Cz; P.claims_phase2[3] = comm_prod;
  Fq blind_expected_claim_postsc1 = tau_claim * (prod_Az_Bz_blind - Cz_blind);
  Fq claim_post_phase1 = (Az_claim * Bz_claim - Cz_claim) * tau_claim;
  P.proof_eq_sc_phase1 = equality_prove(gens.gens_sc.gens_1, t, tape, claim_post_phase1, blind_expected_claim_postsc1, claim_post_phase1,
                                        blind_claim_postsc1, nullptr, nullptr);

  double t2 = now_s();
  Fq r_A = t.challenge_scalar("challenge_Az"), r_B = t.challenge_scalar("challenge_Bz"), r_C = t.challenge_scalar("challenge_Cz");
  Fq claim_phase2 = r_A * Az_claim + r_B * Bz_claim + r_C * Cz_claim;
  Fq blind_claim_phase2 = r_A * Az_blind + r_B * Bz_blind + r_C * Cz_blind;
  FqVec evals_ABC;
  {
    FqVec evals_rx = eq_evals(rx);
    FqVec eA = inst.A.compute_eval_table_sparse(evals_rx, inst.num_cons, z.size());
    FqVec eB = inst.B.compute_eval_table_sparse(evals_rx, inst.num_cons, z.size());
    FqVec eC = inst.C.compute_eval_table_sparse(evals_rx, inst.num_cons, z.size());
    evals_ABC.resize(eA.size());
    for (size_t i = 0; i < eA.size(); i++) evals_ABC[i] = r_A * eA[i] + r_B * eB[i] + r_C * eC[i];
  }
  DensePoly poly_z(z), poly_ABC(evals_ABC);
  FqVec ry;
  Fq claims2[2], blind_claim_postsc2;
  P.sc_proof_phase2 = zk_prove_quad(claim_phase2, blind_claim_phase2, num_rounds_y, poly_z, poly_ABC, gens.gens_sc.gens_1,
                                    gens.gens_sc.gens_3, t, tape, &ry, claims2, &blind_claim_postsc2);
  if (tm) tm->sc_phase_two = now_s() - t2;

  double t3 = now_s();
  FqVec ry1(ry.begin() + 1, ry.end());
  Fq eval_vars_at_ry = poly_vars.evaluate(ry1);
  Fq blind_eval = tape.random_scalar("blind_eval");
  P.proof_eval_vars_at_ry = polyeval_prove(poly_vars, &blinds_vars, ry1, eval_vars_at_ry, &blind_eval, gens.gens_pc, t, tape, &P.comm_vars_at_ry);
  if (tm) tm->polyeval = now_s() - t3;

  Fq blind_eval_Z_at_ry = (fq_one() - ry[0]) * blind_eval;
  Fq blind_expected_claim_postsc2 = claims2[1] * blind_eval_Z_at_ry;
  Fq claim_post_phase2 = claims2[0] * claims2[1];
  P.proof_eq_sc_phase2 = equality_prove(gens.gens_pc.gens.gens_1, t, tape, claim_post_phase2, blind_expected_claim_postsc2,
                                        claim_post_phase2, blind_claim_postsc2, nullptr, nullptr);
  *rx_out = rx;
  *ry_out = ry;
  if (tm) tm->r1cs_sat = now_s() - t0;
  return P;
}

// SparsePolynomial::evaluate  sparse_mlpoly.rs:1566-1593
static Fq sparse_poly_evaluate(size_t num_vars, const std::vector<std::pair<size_t, Fq>>& Z, const FqVec& r) {
  ORC_CHECK(num_vars == r.size());
  Fq sum = fq_zero();
  for (auto& e : Z) {
    Fq chi = fq_one();
    for (size_t k = 0; k < num_vars; k++) {
      bool bit = (e.first >> (num_vars - k - 1)) & 1;  // math.rs:14-19 get_bits: MSB first
      chi = chi * (bit ? r[k] : fq_one() - r[k]);
    }
    sum += chi * e.second;
  }
  return sum;
}

bool r1cs_verify(const R1CSProof& P, size_t num_vars, size_t num_cons, const FqVec& input, const Fq evals[3], Transcript& t,
                 const R1CSGens& gens, FqVec* rx_out, FqVec* ry_out) {  // :351-491
  t.append_protocol_name("R1CS proof");
  t.append_scalars("input", input);
  size_t n = num_vars;
  append_poly_commitment(t, "poly_commitment", P.comm_vars);
  size_t num_rounds_x = log_2(num_cons), num_rounds_y = log_2(2 * num_vars);
  FqVec tau = t.challenge_vector("challenge_tau", num_rounds_x);
  CP claim_phase1 = compress(commit_scalar(fq_zero(), fq_zero(), gens.gens_sc.gens_1));
  CP comm_claim_post_phase1;
  FqVec rx;
  if (!zk_sumcheck_verify(P.sc_proof_phase1, claim_phase1, num_rounds_x, 3, gens.gens_sc.gens_1, gens.gens_sc.gens_4, t,
                          &comm_claim_post_phase1, &rx))
    return false;
  const CP &cAz = P.claims_phase2[0], &cBz = P.claims_phase2[1], &cCz = P.claims_phase2[2], &cProd = P.claims_phase2[3];
  if (!knowledge_verify(P.pok_Cz, gens.gens_sc.gens_1, t, cCz)) return false;
  if (!product_verify(P.proof_prod, gens.gens_sc.gens_1, t, cAz, cBz, cProd)) return false;
  t.append_point("comm_Az_claim", cAz.data());
  t.append_point("comm_Bz_claim", cBz.data());
  t.append_point("comm_Cz_claim", cCz.data());
  t.append_point("comm_prod_Az_Bz_claims", cProd.data());
  Fq taus_bound_rx = eq_evaluate(rx, tau);
  CP expected1 = compress(pt_mul(taus_bound_rx, pt_sub(decompress(cProd), decompress(cCz))));
  if (!equality_verify(P.proof_eq_sc_phase1, gens.gens_sc.gens_1, t, expected1, comm_claim_post_phase1)) return false;
  Fq r_A = t.challenge_scalar("challenge_Az"), r_B = t.challenge_scalar("challenge_Bz"), r_C = t.challenge_scalar("challenge_Cz");
  Fq s3[3] = {r_A, r_B, r_C};
  Pt p3[3] = {decompress(cAz), decompress(cBz), decompress(cCz)};
  CP comm_claim_phase2 = compress(pt_msm(s3, p3, 3));
  CP comm_claim_post_phase2;
  FqVec ry;
  if (!zk_sumcheck_verify(P.sc_proof_phase2, comm_claim_phase2, num_rounds_y, 2, gens.gens_sc.gens_1, gens.gens_sc.gens_3, t,
                          &comm_claim_post_phase2, &ry))
    return false;
  FqVec ry1(ry.begin() + 1, ry.end());
  if (!polyeval_verify(P.proof_eval_vars_at_ry, gens.gens_pc, t, ry1, P.comm_vars_at_ry, P.comm_vars)) return false;
  std::vector<std::pair<size_t, Fq>> ent;
  ent.push_back({0, fq_one()});
  for (size_t i = 0; i < input.size(); i++) ent.push_back({i + 1, input[i]});
  Fq poly_input_eval = sparse_poly_evaluate(log_2(n), ent, ry1);
  Fq s2[2] = {fq_one() - ry[0], ry[0]};
  Pt p2[2] = {decompress(P.comm_vars_at_ry), commit_scalar(poly_input_eval, fq_zero(), gens.gens_pc.gens.gens_1)};
  Pt comm_eval_Z_at_ry = pt_msm(s2, p2, 2);
  CP expected2 = compress(pt_mul(r_A * evals[0] + r_B * evals[1] + r_C * evals[2], comm_eval_Z_at_ry));
  if (!equality_verify(P.proof_eq_sc_phase2, gens.gens_sc.gens_1, t, expected2, comm_claim_post_phase2)) return false;
  *rx_out = rx;
  *ry_out = ry;
  return true;
}

#include "spartan_spark.inc"

}  // namespace orc
