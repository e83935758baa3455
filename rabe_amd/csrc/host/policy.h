// Policy language, LSSS matrix, secret sharing, pruning: the host-side (string / Fr) half of the
// reference's scheme functions, re-implemented in C++ for the engine's host layer.
//
// Reference behaviour restated (paths relative to /root/reference):
//   grammars            src/json.policy.pest, src/human.policy.pest
//   parse / PolicyValue src/utils/policy/pest/{mod,json,human}.rs   (leaf = (name, 1-based column of the
//                       first character inside the quotes), pest/json.rs:10-13)
//   serialize_policy    src/utils/policy/pest/mod.rs:68-114
//   calculate_msp / lw  src/utils/policy/msp.rs:78-147
//   traverse_policy     src/utils/tools/mod.rs:31-61
//   node_index / remove_index / gen_shares_policy / gen_shares / polynomial / calc_coefficients /
//   recover_coefficients / calc_pruned        src/utils/secretsharing/mod.rs:9-221
// Where the reference panics (single-child gates, non-binary AND in `lw`, numeric leaves) this code throws
// PolicyPanic; a pest parse error is PolicyError (-> RabeError, src/error.rs:40-48).
#pragma once
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "fr.h"

namespace rabe { namespace host {

enum class PolicyLanguage { JsonPolicy = 0, HumanPolicy = 1 };   // src/utils/policy/pest/mod.rs:18-23
enum class PolicyType { And, Or, Leaf };

struct PolicyError : std::runtime_error { using std::runtime_error::runtime_error; };
struct PolicyPanic : std::runtime_error { using std::runtime_error::runtime_error; };

struct PolicyNode {
  PolicyType type = PolicyType::Leaf;
  std::string name;          // Leaf
  size_t col = 0;            // Leaf: pest line_col().1
  std::vector<PolicyNode> children;
};

// ---------------------------------------------------------------------------------------------- parser
class PolicyParser {
 public:
  explicit PolicyParser(const std::string& s) : s_(s), n_(s.size()) {}

  PolicyNode parse(PolicyLanguage lang) {
    size_t p = skip(0);
    PolicyNode out;
    size_t end = 0;
    bool ok = (lang == PolicyLanguage::JsonPolicy) ? json_node(p, &end, &out) : human_node(p, &end, &out);
    if (!ok) throw PolicyError("could not parse policy");
    if (skip(end) != n_) throw PolicyError("trailing input after policy");
    return out;
  }

 private:
  const std::string& s_;
  size_t n_;

  // pest counts columns in characters; policies are UTF-8, so count code points since the last '\n'
  size_t col(size_t pos) const {
    size_t start = 0;
    if (pos > 0) {
      size_t nl = s_.rfind('\n', pos - 1);
      if (nl != std::string::npos) start = nl + 1;
    }
    size_t c = 0;
    for (size_t i = start; i < pos; i++)
      if (((unsigned char)s_[i] & 0xC0) != 0x80) c++;
    return c + 1;
  }
  size_t skip(size_t pos) const {
    while (pos < n_) {
      char c = s_[pos];
      if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { pos++; continue; }
      if (s_.compare(pos, 2, "/*") == 0) {
        size_t e = s_.find("*/", pos + 2);
        if (e == std::string::npos) return pos;
        pos = e + 2;
        continue;
      }
      break;
    }
    return pos;
  }
  bool lit(size_t pos, const char* const* alts, size_t n_alts, size_t* end) const {
    for (size_t i = 0; i < n_alts; i++) {
      size_t l = strlen(alts[i]);
      if (s_.compare(pos, l, alts[i]) == 0) { *end = pos + l; return true; }
    }
    return false;
  }
  // `x | QUOTE ~ x ~ QUOTE` inside a non-atomic rule
  bool quoted_or_bare(size_t pos, const char* const* alts, size_t n_alts, size_t* end) const {
    if (lit(pos, alts, n_alts, end)) return true;
    if (pos < n_ && s_[pos] == '"') {
      size_t p = skip(pos + 1), e;
      if (lit(p, alts, n_alts, &e)) {
        p = skip(e);
        if (p < n_ && s_[p] == '"') { *end = p + 1; return true; }
      }
    }
    return false;
  }
  static bool is_hex(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }
  // string = ${ QUOTE ~ inner ~ QUOTE }, inner = @{ char* }
  bool string_rule(size_t pos, size_t* end, PolicyNode* out) const {
    if (pos >= n_ || s_[pos] != '"') return false;
    size_t i = pos + 1, start = i;
    while (i < n_) {
      char c = s_[i];
      if (c == '"') break;
      if (c == '\\') {
        if (i + 1 < n_ && strchr("\"\\/bfnrt", s_[i + 1])) { i += 2; continue; }
        if (i + 5 < n_ && s_[i + 1] == 'u' && is_hex(s_[i + 2]) && is_hex(s_[i + 3]) && is_hex(s_[i + 4]) && is_hex(s_[i + 5])) { i += 6; continue; }
        return false;
      }
      i++;
    }
    if (i >= n_ || s_[i] != '"') return false;
    out->type = PolicyType::Leaf;
    out->name = s_.substr(start, i - start);
    out->col = col(start);
    out->children.clear();
    *end = i + 1;
    return true;
  }
  bool number_rule(size_t pos, size_t* end) const {
    size_t i = pos;
    if (i < n_ && s_[i] == '-') i++;
    if (i < n_ && s_[i] == '0') i++;
    else if (i < n_ && s_[i] >= '1' && s_[i] <= '9') { while (i < n_ && s_[i] >= '0' && s_[i] <= '9') i++; }
    else return false;
    if (i < n_ && s_[i] == '.') { i++; while (i < n_ && s_[i] >= '0' && s_[i] <= '9') i++; }
    if (i < n_ && (s_[i] == 'e' || s_[i] == 'E')) {
      size_t j = i + 1;
      if (j < n_ && (s_[j] == '+' || s_[j] == '-')) j++;
      if (j < n_ && s_[j] >= '0' && s_[j] <= '9') { while (j < n_ && s_[j] >= '0' && s_[j] <= '9') j++; i = j; }
    }
    *end = i;
    return true;
  }

  // ---- JSON grammar (src/json.policy.pest)
  bool json_node(size_t pos, size_t* end, PolicyNode* out) const {
    static const char* NAME[] = {"name", "NAME"};
    static const char* CHILDREN[] = {"children", "CHILDREN"};
    static const char* AND[] = {"and", "AND", "&&"};
    static const char* OR[] = {"or", "OR", "||"};
    if (pos >= n_ || s_[pos] != '{') return false;
    size_t p = skip(pos + 1), e;
    if (!quoted_or_bare(p, NAME, 2, &e)) return false;
    p = skip(e);
    if (p >= n_ || s_[p] != ':') return false;
    p = skip(p + 1);
    // alternative 1: value = string | number
    PolicyNode leaf;
    if (string_rule(p, &e, &leaf)) {
      size_t q = skip(e);
      if (q < n_ && s_[q] == '}') { *out = leaf; *end = q + 1; return true; }
    } else if (number_rule(p, &e)) {
      size_t q = skip(e);
      if (q < n_ && s_[q] == '}') throw PolicyPanic("pest/json.rs:15 unwrap on atomic `number` rule");
    }
    // alternatives 2, 3: and / or
    for (int kind = 0; kind < 2; kind++) {
      if (!quoted_or_bare(p, kind == 0 ? AND : OR, 3, &e)) continue;
      size_t q = skip(e);
      if (q >= n_ || s_[q] != ',') continue;
      q = skip(q + 1);
      if (!quoted_or_bare(q, CHILDREN, 2, &e)) continue;
      q = skip(e);
      if (q >= n_ || s_[q] != ':') continue;
      q = skip(q + 1);
      if (q >= n_ || s_[q] != '[') continue;
      q = skip(q + 1);
      PolicyNode node;
      node.type = kind == 0 ? PolicyType::And : PolicyType::Or;
      bool ok = true;
      if (q < n_ && s_[q] == ']') {
        q++;
      } else {
        while (true) {
          PolicyNode ch;
          size_t ce;
          if (!json_node(q, &ce, &ch)) { ok = false; break; }
          node.children.push_back(std::move(ch));
          q = skip(ce);
          if (q < n_ && s_[q] == ',') { q = skip(q + 1); continue; }
          break;
        }
        if (!ok || q >= n_ || s_[q] != ']') continue;
        q++;
      }
      q = skip(q);
      if (q < n_ && s_[q] == '}') { *out = std::move(node); *end = q + 1; return true; }
    }
    return false;
  }

  // ---- human grammar (src/human.policy.pest)
  bool human_term(size_t pos, size_t* end, PolicyNode* out) const {
    size_t e;
    if (string_rule(pos, end, out)) return true;
    if (number_rule(pos, &e)) throw PolicyPanic("pest/human.rs:15 unwrap on atomic `number` rule");
    if (pos < n_ && strchr("([{", s_[pos])) {
      size_t p = skip(pos + 1);
      if (human_node(p, &e, out)) {
        size_t q = skip(e);
        if (q < n_ && strchr(")]}", s_[q])) { *end = q + 1; return true; }
      }
    }
    return false;
  }
  bool human_chain(size_t pos, bool is_and, size_t* end, PolicyNode* out) const {
    static const char* AND[] = {"and", "AND", "&&"};
    static const char* OR[] = {"or", "OR", "||"};
    PolicyNode first;
    size_t p;
    if (!human_term(pos, &p, &first)) return false;
    PolicyNode node;
    node.type = is_and ? PolicyType::And : PolicyType::Or;
    node.children.push_back(std::move(first));
    while (true) {
      size_t q = skip(p), e;
      if (!quoted_or_bare(q, is_and ? AND : OR, 3, &e)) break;
      q = skip(e);
      PolicyNode nx;
      size_t ne;
      if (!human_term(q, &ne, &nx)) break;
      node.children.push_back(std::move(nx));
      p = ne;
    }
    if (node.children.size() < 2) return false;
    *out = std::move(node);
    *end = p;
    return true;
  }
  bool human_node(size_t pos, size_t* end, PolicyNode* out) const {
    // node = and | or | term : PEG ordered choice, commits to the first alternative that matches
    if (human_chain(pos, true, end, out)) return true;
    if (human_chain(pos, false, end, out)) return true;
    return human_term(pos, end, out);
  }
};

inline PolicyNode parse_policy(const std::string& policy, PolicyLanguage lang) { return PolicyParser(policy).parse(lang); }

// serialize_policy (src/utils/policy/pest/mod.rs:68-114)
inline std::string serialize_policy(const PolicyNode& n, PolicyLanguage lang) {
  if (lang == PolicyLanguage::JsonPolicy) {
    if (n.type == PolicyType::Leaf) return "{\"name\": \"" + n.name + "\"}";
    std::string inner;
    for (size_t i = 0; i < n.children.size(); i++) { if (i) inner += ", "; inner += serialize_policy(n.children[i], lang); }
    return std::string("{\"name\": \"") + (n.type == PolicyType::And ? "and" : "or") + "\", \"children\": [" + inner + "]}";
  }
  if (n.type == PolicyType::Leaf) return n.name;
  std::string inner;
  for (size_t i = 0; i < n.children.size(); i++) {
    if (i) inner += n.type == PolicyType::And ? " and " : " or ";
    inner += serialize_policy(n.children[i], lang);
  }
  return "(" + inner + ")";
}

// ---------------------------------------------------------------------------------------------- MSP
struct AbePolicy {              // src/utils/policy/msp.rs:11-15
  std::vector<std::vector<int8_t>> m;
  std::vector<std::string> pi;
  size_t c = 1;
};
namespace mspdetail {
inline void lw(AbePolicy& msp, const PolicyNode& p, const std::vector<int8_t>& v) {
  if (p.type == PolicyType::Leaf) {
    msp.m.insert(msp.m.begin(), v);
    msp.pi.insert(msp.pi.begin(), p.name);
    return;
  }
  if (p.children.size() < 2) throw PolicyPanic("lw: policy with just a single attribute is not allowed");
  if (p.type == PolicyType::Or) {
    for (const auto& ch : p.children) lw(msp, ch, v);
    return;
  }
  if (p.children.size() != 2) throw PolicyPanic("lw: Invalid policy. Number of arguments under AND != 2");
  std::vector<int8_t> right = v, left;
  right.resize(msp.c, 0);
  right.push_back(1);
  left.resize(msp.c, 0);
  left.push_back(-1);
  msp.c += 1;
  lw(msp, p.children[0], right);
  lw(msp, p.children[1], left);
}
}  // namespace mspdetail
inline AbePolicy calculate_msp(const PolicyNode& p) {
  AbePolicy msp;
  mspdetail::lw(msp, p, std::vector<int8_t>{1});
  for (auto& row : msp.m) row.resize(msp.c, 0);
  std::vector<size_t> order(msp.pi.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return msp.pi[a] < msp.pi[b]; });   // permutation::sort
  AbePolicy out;
  out.c = msp.c;
  for (size_t i : order) { out.m.push_back(msp.m[i]); out.pi.push_back(msp.pi[i]); }
  return out;
}

// ---------------------------------------------------------------------------------------------- tools
inline bool contains(const std::vector<std::string>& data, const std::string& v) {
  return std::find(data.begin(), data.end(), v) != data.end();
}
inline bool traverse_policy(const std::vector<std::string>& attr, const PolicyNode& n) {   // tools/mod.rs:31-61
  if (attr.empty()) return false;
  if (n.type == PolicyType::Leaf) return contains(attr, n.name);
  if (n.type == PolicyType::And) {
    bool ret = true;
    for (const auto& ch : n.children) ret &= traverse_policy(attr, ch);
    return ret;
  }
  bool ret = false;
  for (const auto& ch : n.children) ret |= traverse_policy(attr, ch);
  return ret;
}
inline bool is_negative(const std::string& attr) { return !attr.empty() && attr[0] == '!'; }            // tools/mod.rs:6-9
inline std::string node_index(const PolicyNode& leaf) { return leaf.name + "_" + std::to_string(leaf.col); }   // secretsharing/mod.rs:74-76
inline std::string remove_index(const std::string& s) { return s.substr(0, s.find('_')); }               // :77-80 (split('_')[0])

// ---------------------------------------------------------------------------------------------- sharing
struct FrSource {                       // stands where the reference calls rng.gen::<Fr>()
  virtual ~FrSource() {}
  virtual Fr next_fr() = 0;
};

inline Fr polynomial(const std::vector<Fr>& coeff, uint64_t x) {      // secretsharing/mod.rs:215-221
  Fr share = fr_zero();
  Fr xp = fr_one();                    // x^i, with x^0 = 1
  Fr xf = fr_from_u64(x);
  for (size_t i = 0; i < coeff.size(); i++) {
    share = fr_add(share, fr_mul(coeff[i], xp));
    xp = fr_mul(xp, xf);
  }
  return share;
}
inline std::vector<Fr> gen_shares(const Fr& secret, size_t k, size_t n, FrSource& rng) {    // :124-141
  std::vector<Fr> shares;
  if (k <= n) {
    std::vector<Fr> a{secret};
    for (size_t i = 1; i < k; i++) a.push_back(rng.next_fr());
    for (size_t i = 0; i <= n; i++) shares.push_back(polynomial(a, i));
  }
  return shares;
}
typedef std::vector<std::pair<std::string, Fr>> NamedFr;
inline void gen_shares_policy(const Fr& secret, const PolicyNode& n, FrSource& rng, NamedFr* out) {     // :82-122
  if (n.type == PolicyType::Leaf) { out->push_back({node_index(n), secret}); return; }
  size_t cnt = n.children.size();
  size_t k = n.type == PolicyType::And ? cnt : 1;
  std::vector<Fr> shares = gen_shares(secret, k, cnt, rng);
  for (size_t i = 0; i < cnt; i++) gen_shares_policy(shares[i + 1], n.children[i], rng, out);
}
// how many Fr values gen_shares_policy draws for this tree (k - 1 per gate, k = #children for AND, 1 for OR), and its
// leaf count: lets a batch pull every item's randomness from the generator in order and do the arithmetic in parallel
inline size_t count_share_draws(const PolicyNode& n) {
  if (n.type == PolicyType::Leaf) return 0;
  size_t c = (n.type == PolicyType::And) ? n.children.size() - 1 : 0;
  for (const auto& ch : n.children) c += count_share_draws(ch);
  return c;
}
inline size_t count_leaves(const PolicyNode& n) {
  if (n.type == PolicyType::Leaf) return 1;
  size_t c = 0;
  for (const auto& ch : n.children) c += count_leaves(ch);
  return c;
}
struct VecFrSource : FrSource {          // replays pre-drawn values
  const Fr* v;
  size_t n, pos = 0;
  VecFrSource(const Fr* values, size_t count) : v(values), n(count) {}
  Fr next_fr() override {
    if (pos >= n) throw PolicyPanic("pre-drawn randomness exhausted");
    return v[pos++];
  }
};
inline std::vector<Fr> recover_coefficients(const std::vector<Fr>& list) {     // Lagrange at 0, :60-72
  // res_i = prod_{j != i} (0 - j) / (i - j).  The reference inverts every (i - j) (k^2 inversions); the same field
  // element is num_i * (prod_j (i - j))^-1, and the k denominators share ONE inversion (Montgomery's trick).
  const size_t k = list.size();
  std::vector<Fr> num(k, fr_one()), den(k, fr_one());
  for (size_t a = 0; a < k; a++) {
    for (size_t b = 0; b < k; b++) {
      if (list[a] != list[b]) {
        num[a] = fr_mul(num[a], fr_sub(fr_zero(), list[b]));
        den[a] = fr_mul(den[a], fr_sub(list[a], list[b]));
      }
    }
  }
  std::vector<Fr> prefix(k + 1, fr_one());
  for (size_t a = 0; a < k; a++) prefix[a + 1] = fr_mul(prefix[a], den[a]);
  Fr inv_all;
  if (!fr_inv(prefix[k], &inv_all)) throw PolicyPanic("recover_coefficients: inverse of zero");
  std::vector<Fr> out(k);
  for (size_t a = k; a-- > 0;) {
    out[a] = fr_mul(num[a], fr_mul(inv_all, prefix[a]));
    inv_all = fr_mul(inv_all, den[a]);
  }
  return out;
}
inline void calc_coefficients(const PolicyNode& n, const Fr& coeff, NamedFr* out) {        // :9-57
  if (n.type == PolicyType::Leaf) { out->push_back({node_index(n), coeff}); return; }
  std::vector<Fr> lag;
  if (n.type == PolicyType::And) {
    std::vector<Fr> pts;
    for (size_t i = 1; i <= n.children.size(); i++) pts.push_back(fr_from_u64(i));
    lag = recover_coefficients(pts);
  } else {
    lag.assign(n.children.size(), fr_one());
  }
  for (size_t i = 0; i < n.children.size(); i++) calc_coefficients(n.children[i], fr_mul(coeff, lag[i]), out);
}
typedef std::vector<std::pair<std::string, std::string>> PrunedList;       // (name, name_col)
inline bool calc_pruned(const std::vector<std::string>& attr, const PolicyNode& n, PrunedList* out) {     // :143-199
  if (n.type == PolicyType::Leaf) {
    if (contains(attr, n.name)) { out->push_back({n.name, node_index(n)}); return true; }
    return false;
  }
  if (n.children.size() < 2)
    throw PolicyPanic(n.type == PolicyType::And ? "Error: Invalid policy (AND with just a single child)." : "Error: Invalid policy (OR with just a single child).");
  if (n.type == PolicyType::And) {
    bool ok = true;
    PrunedList acc;
    for (const auto& ch : n.children) {
      PrunedList l;
      bool f = calc_pruned(attr, ch, &l);
      ok = ok && f;
      if (ok) acc.insert(acc.end(), l.begin(), l.end());
    }
    if (ok) out->insert(out->end(), acc.begin(), acc.end());
    return ok;
  }
  for (const auto& ch : n.children) {
    PrunedList l;
    if (calc_pruned(attr, ch, &l)) { out->insert(out->end(), l.begin(), l.end()); return true; }
  }
  return false;
}


// ---------------------------------------------------------------------------------------------- DNF policies (bdabe, mke08)
// src/utils/policy/dnf.rs:203-243: rejects an OR below an AND only (an AND below an AND passes here and fails in dnf_terms)
inline bool policy_in_dnf(const PolicyNode& n, bool conjunction = false) {
  if (n.type == PolicyType::Leaf) return true;
  if (n.type == PolicyType::And) {
    bool ret = true;
    for (const auto& ch : n.children) ret &= policy_in_dnf(ch, true);
    return ret;
  }
  if (conjunction) return false;
  bool ret = true;
  for (const auto& ch : n.children) ret &= policy_in_dnf(ch, conjunction);
  return ret;
}
// One conjunction of the DNF: its attribute names and, in order, the indices of the public attribute keys whose elements were
// multiplied / added into it (a name that matches several keys takes all of them, one that matches none is dropped: dnf.rs:117-143).
struct DnfTerm {
  std::vector<std::string> attrs;
  std::vector<size_t> keys;
};
// dnf.rs:106-183.  Child k of an OR is sent to term index 2k (`i + i` with the loop's shadowing `i`, :162-164) while a missing
// index APPENDS (:133-141) -- both restated as they are.  parent: 0 = none, 1 = and, 2 = or.
inline bool dnf_walk(std::vector<DnfTerm>* terms, const std::vector<std::string>& pk_attrs, const PolicyNode& n, size_t i, int parent) {
  if (n.type == PolicyType::Leaf) {
    for (size_t k = 0; k < pk_attrs.size(); k++) {
      if (pk_attrs[k] != n.name) continue;
      if (terms->size() > i) { (*terms)[i].attrs.push_back(pk_attrs[k]); (*terms)[i].keys.push_back(k); }
      else terms->push_back(DnfTerm{{pk_attrs[k]}, {k}});
    }
    return true;
  }
  int arr_parent;
  if (parent == 0) arr_parent = (n.type == PolicyType::And) ? 1 : 2;
  else if (parent == 2) { if (n.type != PolicyType::And) return false; arr_parent = 1; }
  else return false;                 // an inner node below an AND (:172-177)
  bool ret = true;
  if (arr_parent == 1) {
    for (const auto& ch : n.children) ret = ret && dnf_walk(terms, pk_attrs, ch, i, 1);
  } else {
    for (size_t k = 0; k < n.children.size(); k++) ret = ret && dnf_walk(terms, pk_attrs, n.children[k], k + k, 2);
  }
  return ret;
}
// dnf.rs:186-201: the terms, stably sorted by their number of attributes; false = the Err the callers `.unwrap()`
inline bool json_to_dnf(const PolicyNode& n, const std::vector<std::string>& pk_attrs, std::vector<DnfTerm>* terms) {
  terms->clear();
  if (!dnf_walk(terms, pk_attrs, n, 0, 0)) return false;
  std::stable_sort(terms->begin(), terms->end(), [](const DnfTerm& a, const DnfTerm& b) { return a.attrs.size() < b.attrs.size(); });
  return true;
}

}}  // namespace rabe::host
