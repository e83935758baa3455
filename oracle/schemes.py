"""The reference's scheme functions restated in the reference's own operation order -- TEST ORACLE.

Test infrastructure only (see oracle/bn254.py header; PARITY UNPINNED for group values).
Every function follows the cited reference lines statement by statement: variable-base
`G * Fr` double-and-add for every hash-to-group and every product term, one full
`pairing()` (Miller loop + final exponentiation) per call site, string-keyed linear scans.
`rng` replaces `rand::thread_rng()`; draws happen where the reference draws (SURVEY.md 8c
"explicit-randomness contract").  KDF + AES (src/utils/aes/mod.rs) stay outside: each
encrypt returns the Gt `msg` next to the ciphertext, each decrypt returns the recovered Gt.

Keys and ciphertexts are plain dicts whose fields mirror the reference structs
(ac17/mod.rs:58-135, bsw/mod.rs:39-89, lsw/mod.rs:40-83, aw11/mod.rs:46-97).
"""
import hashlib

from . import bn254 as bn
from . import policy as pol

ASSUMPTION_SIZE = 2          # ac17/mod.rs:138


# ----------------------------------------------------------------------------- utils/hash

def sha3_hash_fr(data):
    """src/utils/hash/mod.rs:23-31."""
    return bn.fr_from_be32_reduce(hashlib.sha3_256(data.encode("utf-8")).digest())


def sha3_hash_g1(g, data):
    """src/utils/hash/mod.rs:10-20 with T = G1."""
    return bn.g1_mul(g, sha3_hash_fr(data))


def sha3_hash_g2(g, data):
    return bn.g2_mul(g, sha3_hash_fr(data))


# ============================================================================= AC17 (FAME)

def ac17_setup(rng):
    """ac17/mod.rs:141-182."""
    g = rng.g1()
    h = rng.g2()
    e_gh = bn.pairing(g, h)
    a, b = [], []
    for _ in range(ASSUMPTION_SIZE):
        a.append(rng.fr())
        b.append(rng.fr())
    k = [rng.fr() for _ in range(ASSUMPTION_SIZE + 1)]
    h_a = [bn.g2_mul(h, a[i]) for i in range(ASSUMPTION_SIZE)] + [h]
    g_k = [bn.g1_mul(g, k[i]) for i in range(ASSUMPTION_SIZE + 1)]
    e_gh_ka = [bn.gt_pow(e_gh, (k[i] * a[i] + k[ASSUMPTION_SIZE]) % bn.R) for i in range(ASSUMPTION_SIZE)]
    pk = {"g": g, "h_a": h_a, "e_gh_ka": e_gh_ka}
    msk = {"g": g, "h": h, "g_k": g_k, "a": a, "b": b}
    return pk, msk


def ac17_cp_keygen(msk, attributes, rng):
    """ac17/mod.rs:191-264."""
    if len(attributes) == 0:
        raise ValueError("empty attributes!")
    r = [rng.fr() for _ in range(ASSUMPTION_SIZE)]
    s = sum(r) % bn.R
    br = [msk["b"][i] * r[i] % bn.R for i in range(ASSUMPTION_SIZE)] + [s]
    k_0 = [bn.g2_mul(msk["h"], br[i]) for i in range(ASSUMPTION_SIZE + 1)]
    a = msk["a"]
    g = msk["g"]
    k = []
    for attr in attributes:
        key = []
        sigma_attr = rng.fr()
        for t in range(ASSUMPTION_SIZE):
            prod = None
            a_t = bn.fr_inv(a[t])
            for l in range(ASSUMPTION_SIZE + 1):
                prod = bn.g1_add(prod, bn.g1_mul(sha3_hash_g1(g, attr + str(l) + str(t)), br[l] * a_t % bn.R))
            prod = bn.g1_add(prod, bn.g1_mul(g, sigma_attr * a_t % bn.R))
            key.append(prod)
        key.append(bn.g1_mul(g, (-sigma_attr) % bn.R))
        k.append((attr, key))
    k_p = []
    sigma = rng.fr()
    for t in range(ASSUMPTION_SIZE):
        prod = msk["g_k"][t]
        a_t = bn.fr_inv(a[t])
        for l in range(ASSUMPTION_SIZE + 1):
            prod = bn.g1_add(prod, bn.g1_mul(sha3_hash_g1(g, "01" + str(l) + str(t)), br[l] * a_t % bn.R))
        prod = bn.g1_add(prod, bn.g1_mul(g, sigma * a_t % bn.R))
        k_p.append(prod)
    k_p.append(bn.g1_add(msk["g_k"][ASSUMPTION_SIZE], bn.g1_mul(g, (-sigma) % bn.R)))
    return {"attr": list(attributes), "sk": {"k_0": k_0, "k": k, "k_p": k_p}}


def ac17_cp_encrypt(pk, policy, language, rng, msg):
    """ac17/mod.rs:274-376.  `msg` is the Gt the reference samples at :362 (supplied by the
    caller); returns the ciphertext without the AES part."""
    tree = pol.parse(policy, language)
    m, pi, _c = pol.calculate_msp(tree)
    num_cols = len(m[0])
    num_rows = len(m)
    s = [rng.fr() for _ in range(ASSUMPTION_SIZE)]
    ssum = sum(s) % bn.R
    c_0 = [bn.g2_mul(pk["h_a"][i], s[i]) for i in range(ASSUMPTION_SIZE)]
    c_0.append(bn.g2_mul(pk["h_a"][ASSUMPTION_SIZE], ssum))
    g = pk["g"]
    hash_table = []
    for j in range(num_cols):
        x = []
        for l in range(ASSUMPTION_SIZE + 1):
            y = []
            for t in range(ASSUMPTION_SIZE):
                y.append(sha3_hash_g1(g, "0" + str(j + 1) + str(l) + str(t)))
            x.append(y)
        hash_table.append(x)
    c = []
    for i in range(num_rows):
        ct = []
        for l in range(ASSUMPTION_SIZE + 1):
            prod = None
            for t in range(ASSUMPTION_SIZE):
                h = sha3_hash_g1(g, pi[i] + str(l) + str(t))
                for j in range(num_cols):
                    if m[i][j] == 1:
                        h = bn.g1_add(h, hash_table[j][l][t])
                    elif m[i][j] == -1:
                        h = bn.g1_sub(h, hash_table[j][l][t])
                prod = bn.g1_add(prod, bn.g1_mul(h, s[t]))
            ct.append(prod)
        c.append((pi[i], ct))
    c_p = bn.GT_ONE
    for i in range(ASSUMPTION_SIZE):
        c_p = bn.gt_mul(c_p, bn.gt_pow(pk["e_gh_ka"][i], s[i]))
    return {"policy": (policy, language), "ct": {"c_0": c_0, "c": c, "c_p": bn.gt_mul(c_p, msg)}}


def ac17_cp_decrypt(sk, ct):
    """ac17/mod.rs:385-430.  Returns the Gt fed to `decrypt_symmetric`, or raises ValueError."""
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(sk["attr"], tree):
        raise ValueError("Error in cp_decrypt: attributes in SK do not match policy in CT.")
    ok, lst = pol.calc_pruned(sk["attr"], tree)
    if not ok:
        raise ValueError("Error: attributes in sk do not match policy in ct.")
    prod1 = bn.GT_ONE
    prod2 = bn.GT_ONE
    for i in range(ASSUMPTION_SIZE + 1):
        prod_h = None
        prod_g = None
        for cur in lst:
            for name, vec in ct["ct"]["c"]:
                if name == cur[0]:
                    prod_g = bn.g1_add(prod_g, vec[i])
            for name, vec in sk["sk"]["k"]:
                if name == cur[0]:
                    prod_h = bn.g1_add(prod_h, vec[i])
        prod1 = bn.gt_mul(prod1, bn.pairing(bn.g1_add(sk["sk"]["k_p"][i], prod_h), ct["ct"]["c_0"][i]))
        prod2 = bn.gt_mul(prod2, bn.pairing(prod_g, sk["sk"]["k_0"][i]))
    return bn.gt_mul(ct["ct"]["c_p"], bn.gt_mul(prod2, bn.gt_inv(prod1)))


def ac17_kp_keygen(msk, policy, language, rng):
    """ac17/mod.rs:439-547.  Draw order: r0, r1 (:455-459), sigma'_1..sigma'_{c-1} (:471-474), then sigma_i per row (:481)."""
    tree = pol.parse(policy, language)
    m, pi, _c = pol.calculate_msp(tree)
    num_cols, num_rows = len(m[0]), len(m)
    r = [rng.fr() for _ in range(ASSUMPTION_SIZE)]
    br = [msk["b"][i] * r[i] % bn.R for i in range(ASSUMPTION_SIZE)] + [sum(r) % bn.R]
    k_0 = [bn.g2_mul(msk["h"], br[i]) for i in range(ASSUMPTION_SIZE + 1)]
    sigma_prime = [rng.fr() for _ in range(num_cols - 1)]
    a, g = msk["a"], msk["g"]
    k = []
    for i in range(num_rows):
        key = []
        sigma_attr = rng.fr()
        for t in range(ASSUMPTION_SIZE):
            prod = None
            a_t = bn.fr_inv(a[t])
            for l in range(ASSUMPTION_SIZE + 1):
                prod = bn.g1_add(prod, bn.g1_mul(sha3_hash_g1(g, pi[i] + str(l) + str(t)), br[l] * a_t % bn.R))
            prod = bn.g1_add(prod, bn.g1_mul(g, sigma_attr * a_t % bn.R))
            if m[i][0] == 1:
                prod = bn.g1_add(prod, msk["g_k"][t])
            elif m[i][0] == -1:
                prod = bn.g1_sub(prod, msk["g_k"][t])
            temp = None                                   # NOT reset per column: accumulates across j (:496-516)
            for j in range(1, num_cols):
                for l in range(ASSUMPTION_SIZE + 1):
                    temp = bn.g1_add(temp, bn.g1_mul(sha3_hash_g1(g, "0" + str(j) + str(l) + str(t)), br[l] * a_t % bn.R))
                temp = bn.g1_add(temp, bn.g1_mul(g, (-sigma_prime[j - 1]) % bn.R))
                if m[i][j] == 1:
                    prod = bn.g1_add(prod, temp)
                elif m[i][j] == -1:
                    prod = bn.g1_sub(prod, temp)
            key.append(prod)
        sk_i3 = bn.g1_mul(g, (-sigma_attr) % bn.R)
        if m[i][0] == 1:
            sk_i3 = bn.g1_add(sk_i3, msk["g_k"][ASSUMPTION_SIZE])
        elif m[i][0] == -1:
            sk_i3 = bn.g1_sub(sk_i3, msk["g_k"][ASSUMPTION_SIZE])
        for j in range(1, num_cols):
            if m[i][j] == 1:
                sk_i3 = bn.g1_add(sk_i3, bn.g1_mul(g, (-sigma_prime[j - 1]) % bn.R))
            elif m[i][j] == -1:
                sk_i3 = bn.g1_sub(sk_i3, bn.g1_mul(g, (-sigma_prime[j - 1]) % bn.R))
        key.append(sk_i3)
        k.append((pi[i], key))
    return {"policy": (policy, language), "sk": {"k_0": k_0, "k": k, "k_p": []}}


def ac17_kp_encrypt(pk, attributes, rng, msg):
    """ac17/mod.rs:556-616."""
    s = [rng.fr() for _ in range(ASSUMPTION_SIZE)]
    ssum = sum(s) % bn.R
    c_0 = [bn.g2_mul(pk["h_a"][i], s[i]) for i in range(ASSUMPTION_SIZE)]
    c_0.append(bn.g2_mul(pk["h_a"][ASSUMPTION_SIZE], ssum))
    c = []
    for attr in attributes:
        ct = []
        for l in range(ASSUMPTION_SIZE + 1):
            prod = None
            for t in range(ASSUMPTION_SIZE):
                prod = bn.g1_add(prod, bn.g1_mul(sha3_hash_g1(pk["g"], attr + str(l) + str(t)), s[t]))
            ct.append(prod)
        c.append((attr, ct))
    c_p = bn.GT_ONE
    for i in range(ASSUMPTION_SIZE):
        c_p = bn.gt_mul(c_p, bn.gt_pow(pk["e_gh_ka"][i], s[i]))
    return {"attr": list(attributes), "ct": {"c_0": c_0, "c": c, "c_p": bn.gt_mul(c_p, msg)}}


def ac17_kp_decrypt(sk, ct):
    """ac17/mod.rs:625-675."""
    tree = pol.parse(sk["policy"][0], sk["policy"][1])
    if not pol.traverse_policy(ct["attr"], tree):
        raise ValueError("Error in kp_decrypt: attributes in ct do not match policy in sk.")
    ok, lst = pol.calc_pruned(ct["attr"], tree)
    if not ok:
        raise ValueError("Error in kp_decrypt: pruned attributes in sk do not match policy in ct.")
    prod1 = bn.GT_ONE
    prod2 = bn.GT_ONE
    for i in range(ASSUMPTION_SIZE + 1):
        prod_h = None
        prod_g = None
        for cur in lst:
            for name, vec in ct["ct"]["c"]:
                if name == cur[0]:
                    prod_g = bn.g1_add(prod_g, vec[i])
            for name, vec in sk["sk"]["k"]:
                if name == cur[0]:
                    prod_h = bn.g1_add(prod_h, vec[i])
        prod1 = bn.gt_mul(prod1, bn.pairing(prod_h, ct["ct"]["c_0"][i]))
        prod2 = bn.gt_mul(prod2, bn.pairing(prod_g, sk["sk"]["k_0"][i]))
    return bn.gt_mul(ct["ct"]["c_p"], bn.gt_mul(prod2, bn.gt_inv(prod1)))


# ============================================================================= BSW

def bsw_setup(rng):
    """bsw/mod.rs:92-114."""
    g1 = rng.g1()
    g2 = rng.g2()
    beta = rng.fr()
    alpha = rng.fr()
    h = bn.g1_mul(g1, beta)
    f = bn.g2_mul(g2, bn.fr_inv(beta))
    g2_alpha = bn.g2_mul(g2, alpha)
    e_gg_alpha = bn.pairing(g1, g2_alpha)
    return ({"g1": g1, "g2": g2, "h": h, "f": f, "e_gg_alpha": e_gg_alpha},
            {"beta": beta, "g2_alpha": g2_alpha})


def bsw_keygen(pk, msk, attributes, rng):
    """bsw/mod.rs:125-152."""
    if len(attributes) == 0:
        return None
    r = rng.fr()
    g2_r = bn.g2_mul(pk["g2"], r)
    d = bn.g2_mul(bn.g2_add(msk["g2_alpha"], g2_r), bn.fr_inv(msk["beta"]))
    d_j = []
    for j in attributes:
        r_j = rng.fr()
        d_j.append({"string": j,
                    "g1": bn.g1_mul(pk["g1"], r_j),
                    "g2": bn.g2_add(g2_r, bn.g2_mul(sha3_hash_g2(pk["g2"], j), r_j))})
    return {"d": d, "d_j": d_j}


def bsw_delegate(pk, sk, subset, rng):
    """bsw/mod.rs:162-206.  Draw order: r (:185), then r_j per delegated attribute (:190)."""
    attr_str = [v["string"] for v in sk["d_j"]]
    if not set(subset).issubset(set(attr_str)):
        return None
    if len(subset) == 0:
        return None
    r = rng.fr()
    d_j = []
    for attr in subset:
        r_j = rng.fr()
        old = next(x for x in sk["d_j"] if x["string"] == attr)
        d_j.append({"string": attr,
                    "g1": bn.g1_add(old["g1"], bn.g1_mul(pk["g1"], r_j)),
                    "g2": bn.g2_add(bn.g2_add(old["g2"], bn.g2_mul(sha3_hash_g2(pk["g2"], attr), r_j)), bn.g2_mul(pk["g2"], r))})
    return {"d": bn.g2_add(sk["d"], bn.g2_mul(pk["f"], r)), "d_j": d_j}


def bsw_encrypt(pk, policy, language, rng, msg):
    """bsw/mod.rs:217-251 (secret drawn first, the Gt `msg` second -- supplied by the caller)."""
    secret = rng.fr()
    tree = pol.parse(policy, language)
    shares = pol.gen_shares_policy(secret, tree, rng)
    c = bn.g1_mul(pk["h"], secret)
    c_p = bn.gt_mul(bn.gt_pow(pk["e_gg_alpha"], secret), msg)
    c_y = []
    for node, val in shares:
        j = pol.remove_index(node)
        c_y.append({"string": node,
                    "g1": bn.g1_mul(pk["g1"], val),
                    "g2": bn.g2_mul(sha3_hash_g2(pk["g2"], j), val)})
    return {"policy": (policy, language), "c": c, "c_p": c_p, "c_y": c_y}


def bsw_decrypt(sk, ct):
    """bsw/mod.rs:260-318."""
    attr = [v["string"] for v in sk["d_j"]]
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(attr, tree):
        raise ValueError("Error in bsw/encrypt: attributes do not match policy.")
    ok, pruned = pol.calc_pruned(attr, tree)
    if not ok:
        raise ValueError("Error in bsw/encrypt: attributes do not match policy.")
    z = pol.calc_coefficients(tree, 1)
    a = bn.GT_ONE
    for k, j in pruned:
        c_y = next((x for x in ct["c_y"] if x["string"] == j), None)
        if c_y is None:
            continue
        d_j = next((x for x in sk["d_j"] if x["string"] == k), None)
        if d_j is None:
            continue
        for zname, zval in z:
            if zname == j:
                t = bn.gt_mul(bn.pairing(c_y["g1"], d_j["g2"]), bn.gt_inv(bn.pairing(d_j["g1"], c_y["g2"])))
                a = bn.gt_mul(a, bn.gt_pow(t, zval))
    return bn.gt_mul(ct["c_p"], bn.gt_inv(bn.gt_mul(bn.pairing(ct["c"], sk["d"]), bn.gt_inv(a))))


# ============================================================================= LSW

def lsw_setup(rng):
    """lsw/mod.rs:86-110."""
    alpha1 = rng.fr()
    alpha2 = rng.fr()
    b = rng.fr()
    alpha = alpha1 * alpha2 % bn.R
    g1 = rng.g1()
    g2 = rng.g2()
    h_g1 = rng.g1()
    h_g2 = rng.g2()
    g1_b = bn.g1_mul(g1, b)
    g1_b2 = bn.g1_mul(g1_b, b)
    h_b = bn.g1_mul(h_g1, b)
    e_gg_alpha = bn.gt_pow(bn.pairing(g1, g2), alpha)
    return ({"g1": g1, "g2": g2, "g1_b": g1_b, "g1_b2": g1_b2, "h_b": h_b, "e_gg_alpha": e_gg_alpha},
            {"alpha1": alpha1, "alpha2": alpha2, "b": b, "h_g1": h_g1, "h_g2": h_g2})


def lsw_keygen(pk, msk, policy, language, rng):
    """lsw/mod.rs:121-170: shares of alpha1 first (all gate coefficients), then one `random` per share."""
    tree = pol.parse(policy, language)
    shares = pol.gen_shares_policy(msk["alpha1"], tree, rng)
    dj = []
    for share_str, share_value in shares:
        striped = pol.remove_index(share_str)
        random = rng.fr()
        if pol.is_negative(striped):
            share_hash = sha3_hash_fr(striped)
            dj.append((striped, None, None,
                       bn.g1_add(bn.g1_mul(pk["g1"], share_value), bn.g1_mul(pk["g1_b2"], random)),
                       bn.g1_add(bn.g1_mul(pk["g1_b"], share_hash * random % bn.R), bn.g1_mul(msk["h_g1"], random)),
                       bn.g1_mul(pk["g1"], (-random) % bn.R)))
        else:
            share_hash = sha3_hash_g1(pk["g1"], striped)
            dj.append((striped,
                       bn.g1_add(bn.g1_mul(pk["g1"], msk["alpha2"] * share_value % bn.R), bn.g1_mul(share_hash, random)),
                       bn.g2_mul(pk["g2"], random),
                       None, None, None))
    return {"policy": (policy, language), "dj": dj}


def lsw_encrypt(pk, attributes, rng, msg):
    """lsw/mod.rs:180-219, including the `sx[0] = sx[0] - sx[_i]` index quirk at :197-200:
    at iteration i the element just pushed is sx[i+1], but sx[i] is what gets subtracted."""
    if len(attributes) == 0:
        raise ValueError("attributes or data empty")
    ej = []
    secret = rng.fr()
    sx = [secret]
    for i, _attr in enumerate(attributes):
        sx.append(rng.fr())
        sx[0] = (sx[0] - sx[i]) % bn.R
    for i, attr in enumerate(attributes):
        ej.append((attr,
                   bn.g1_mul(sha3_hash_g1(pk["g1"], attr), secret),
                   bn.g1_mul(pk["g1_b"], sx[i]),
                   bn.g1_add(bn.g1_mul(pk["g1_b2"], sx[i] * sha3_hash_fr(attr) % bn.R), bn.g1_mul(pk["h_b"], sx[i]))))
    e1 = bn.gt_mul(bn.gt_pow(pk["e_gg_alpha"], secret), msg)
    e2 = bn.g2_mul(pk["g2"], secret)
    return {"e1": e1, "e2": e2, "ej": ej}


def lsw_decrypt(sk, ct):
    """lsw/mod.rs:228-290 (negative-attribute branch is a TODO in the reference: `_z_y` keeps
    its previous value, :265-278)."""
    attr = [a[0] for a in ct["ej"]]
    tree = pol.parse(sk["policy"][0], sk["policy"][1])
    ok, lst = pol.calc_pruned(attr, tree)
    if not ok:
        raise ValueError("Error in lsw/decrypt: attributes do not match policy.")
    prod_t = bn.GT_ONE
    z_y = bn.GT_ONE
    coeff_list = pol.calc_coefficients(tree, 1)
    for name, name_col in lst:
        sk_attr = next(a for a in sk["dj"] if a[0] == name)
        ct_attr = next(a for a in ct["ej"] if a[0] == name)
        coeff = next(c for c in coeff_list if c[0] == name_col)
        if pol.is_negative(name):
            pass
        else:
            z_y = bn.gt_mul(bn.pairing(sk_attr[1], ct["e2"]), bn.gt_inv(bn.pairing(ct_attr[1], sk_attr[2])))
        prod_t = bn.gt_mul(prod_t, bn.gt_pow(z_y, coeff[1]))
    return bn.gt_mul(ct["e1"], bn.gt_inv(prod_t))


# ============================================================================= AW11

def aw11_setup(rng):
    """aw11/mod.rs:100-108."""
    return {"g1": rng.g1(), "g2": rng.g2()}


def aw11_authgen(gk, attributes, rng):
    """aw11/mod.rs:121-151 (one `pairing(g1,g2)` per attribute in the reference)."""
    if len(attributes) == 0:
        return None
    sk, pk = [], []
    for attr in attributes:
        name = attr.upper()
        alpha_i = rng.fr()
        y_i = rng.fr()
        sk.append((name, alpha_i, y_i))
        pk.append((name, bn.gt_pow(bn.pairing(gk["g1"], gk["g2"]), alpha_i), bn.g2_mul(gk["g2"], y_i)))
    return {"attr": pk}, {"attr": sk}


def aw11_keygen(gk, msk, name, attributes):
    """aw11/mod.rs:165-231 (keygen + add_to_attribute; no randomness)."""
    if len(attributes) == 0:
        raise ValueError("empty _attributes")
    if len(name) == 0:
        raise ValueError("empty _name")
    sk = {"gid": name, "attr": []}
    for attribute in attributes:
        h = sha3_hash_g1(gk["g1"], sk["gid"])
        auth = next(a for a in msk["attr"] if a[0] == attribute)          # .unwrap() panics otherwise
        sk["attr"].append((auth[0].upper(),
                           bn.g1_add(bn.g1_mul(gk["g1"], auth[1]), bn.g1_mul(h, auth[2]))))
    return sk


def _aw11_find_pk_attr(pks, attr):
    """aw11/mod.rs:374-390."""
    for pk in pks:
        for t in pk["attr"]:
            if t[0] == attr:
                return t
    return None


def aw11_encrypt(gk, pks, policy, language, rng, msg):
    """aw11/mod.rs:241-289.  Draw order: s, gate coefficients of the s-shares, gate coefficients
    of the zero-shares, [msg -- supplied], then r_x per row.  The MSP is built (and must not
    panic) but never used (:253-255)."""
    tree = pol.parse(policy, language)
    pol.calculate_msp(tree)
    s = rng.fr()
    s_shares = pol.gen_shares_policy(s, tree, rng)
    w_shares = pol.gen_shares_policy(0, tree, rng)
    egg = bn.pairing(gk["g1"], gk["g2"])
    c_0 = bn.gt_mul(msg, bn.gt_pow(egg, s))
    c = []
    for i, (attr_name, attr_share) in enumerate(s_shares):
        r_x = rng.fr()
        pk_attr = _aw11_find_pk_attr(pks, pol.remove_index(attr_name.upper()))
        if pk_attr is None:
            continue
        c.append((attr_name.upper(),
                  bn.gt_mul(bn.gt_pow(egg, attr_share), bn.gt_pow(pk_attr[1], r_x)),
                  bn.g2_mul(gk["g2"], r_x),
                  bn.g2_add(bn.g2_mul(pk_attr[2], r_x), bn.g2_mul(gk["g2"], w_shares[i][1]))))
    return {"policy": (policy, language), "c_0": c_0, "c": c}


def aw11_decrypt(gk, sk, ct):
    """aw11/mod.rs:298-366."""
    str_attr = [a[0] for a in sk["attr"]]
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(str_attr, tree):
        raise ValueError("Error: attributes in sk do not match policy in ct.")
    ok, lst = pol.calc_pruned(str_attr, tree)
    coeff_list = pol.calc_coefficients(tree, 1)
    if not ok:
        raise ValueError("Error in aw11/decrypt: attributes in sk do not match policy in ct.")
    h = sha3_hash_g1(gk["g1"], sk["gid"])
    egg_s = bn.GT_ONE
    for name, name_col in lst:
        sk_attr = next(a for a in sk["attr"] if a[0] == name)
        ct_attr = next(a for a in ct["c"] if a[0] == name_col)
        num = bn.gt_mul(ct_attr[1], bn.pairing(h, ct_attr[3]))
        dem = bn.pairing(sk_attr[1], ct_attr[2])
        coeff = next(c[1] for c in coeff_list if c[0] == name_col)
        egg_s = bn.gt_mul(egg_s, bn.gt_pow(bn.gt_mul(num, bn.gt_inv(dem)), coeff))
    return bn.gt_mul(ct["c_0"], bn.gt_inv(egg_s))


# ============================================================================= GHW11 (outsourced decryption)

def ghw11_setup(rng):
    """ghw11/mod.rs:92-111 (draws: g1, g2, a, alpha)."""
    g1 = bn.g1_mul(bn.G1_GEN, rng.fr())
    g2 = bn.g2_mul(bn.G2_GEN, rng.fr())
    a = rng.fr()
    g1_a = bn.g1_mul(g1, a)
    g2_a = bn.g2_mul(g2, a)
    alpha = rng.fr()
    e_gg_alpha = bn.gt_pow(bn.pairing(g1, g2), alpha)
    g2_alpha = bn.g2_mul(g2, alpha)
    pk = {"g1": g1, "g2": g2, "g1_a": g1_a, "g2_a": g2_a, "e_gg_alpha": e_gg_alpha}
    return pk, {"g2_alpha": g2_alpha, "pk": pk}


def ghw11_keygen(pk, msk, attributes, rng):
    """ghw11/mod.rs:123-152; None for an empty attribute list."""
    if not attributes:
        return None
    r = rng.fr()
    g2_r = bn.g2_mul(pk["g2"], r)
    k = bn.g2_add(msk["g2_alpha"], bn.g2_mul(pk["g2_a"], r))
    attr_key = [{"string": j, "k_x": bn.g2_mul(sha3_hash_g2(pk["g2"], j), r)} for j in attributes]
    return {"k": k, "l": g2_r, "attr_key": attr_key}


def ghw11_tkgen(sk, rng):
    """ghw11/mod.rs:156-180: (transform key, retrieve key z)."""
    z = rng.fr()
    z_inv = bn.fr_inv(z)
    tk = {"k_z": bn.g2_mul(sk["k"], z_inv), "l_z": bn.g2_mul(sk["l"], z_inv),
          "attr_key_z": [{"string": a["string"], "k_x": bn.g2_mul(a["k_x"], z_inv)} for a in sk["attr_key"]]}
    return tk, {"z": z}


def ghw11_encrypt(pk, policy, language, rng, msg):
    """ghw11/mod.rs:192-228: secret first, the Gt `msg` second (supplied by the caller), the gate coefficients, then one t_i per share."""
    secret = rng.fr()
    tree = pol.parse(policy, language)
    shares = pol.gen_shares_policy(secret, tree, rng)
    c = bn.gt_mul(bn.gt_pow(pk["e_gg_alpha"], secret), msg)
    c1 = bn.g1_mul(pk["g1"], secret)
    ci_di = []
    for node, i_val in shares:
        t_i = rng.fr()
        j = pol.remove_index(node)
        ci = bn.g1_add(bn.g1_mul(pk["g1_a"], i_val), bn.g1_mul(sha3_hash_g1(pk["g1"], j), (-t_i) % bn.R))
        ci_di.append((node, ci, bn.g1_mul(pk["g1"], t_i)))
    return {"policy": (policy, language), "c": c, "c1": c1, "ci_di": ci_di}


def ghw11_transform(ct, tk):
    """ghw11/mod.rs:231-295: the outsourced part -- m + 2 pairings, 2m G1 multiplications."""
    attr = [v["string"] for v in tk["attr_key_z"]]
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(attr, tree):
        raise ValueError("Error: attributes in tk do not match policy in ct.")
    ok, lst = pol.calc_pruned(attr, tree)
    coeff_list = pol.calc_coefficients(tree, 1)
    if not ok:
        raise ValueError("Error in Ghw11/decrypt: attributes in sk do not match policy in ct.")
    t = bn.GT_ONE
    ci_wi = None          # G1::zero()
    for name, name_col in lst:
        coeff = next(c for n, c in coeff_list if n == name_col)
        tk_attr = next(a for a in tk["attr_key_z"] if a["string"] == name)
        ct_attr = next(a for a in ct["ci_di"] if a[0] == name_col)
        ci_wi = bn.g1_add(ci_wi, bn.g1_mul(ct_attr[1], coeff))
        t = bn.gt_mul(t, bn.pairing(bn.g1_mul(ct_attr[2], coeff), tk_attr["k_x"]))
    t = bn.gt_mul(t, bn.pairing(ci_wi, tk["l_z"]))
    t = bn.gt_mul(bn.pairing(ct["c1"], tk["k_z"]), bn.gt_inv(t))
    return {"c": ct["c"], "t": t}


def ghw11_decrypt_out(pct, rk):
    """ghw11/mod.rs:298-305: the client's part, one Gt power.  Returns the Gt handed to decrypt_symmetric."""
    return bn.gt_mul(pct["c"], bn.gt_inv(bn.gt_pow(pct["t"], rk["z"])))


# ============================================================================= BDABE and MKE08 (DNF policies)

_DNF_OPS = (bn.gt_mul, bn.g1_add, bn.g2_add)


def _from_authority(attr, authority):
    """bdabe/mod.rs:457-467 = mke08/mod.rs: exactly one "::" and the part before it equals the authority's name."""
    if attr.count("::") != 1:          # match_indices("::") counts non-overlapping matches, as str.count does
        return False
    return attr[:attr.index("::")] == authority


def _attr_exponent(attribute, authority_name, secret):
    return sha3_hash_fr(attribute) * sha3_hash_fr(authority_name) * secret % bn.R


def _is_satisfiable(conjunction, sk_a):
    return all(any(k[0] == a for k in sk_a) for a in conjunction)


def _calc_satisfiable(conjunction, sk_a):
    """bdabe/mod.rs:424-447 = mke08/mod.rs: sums of the attribute keys; the start value (G1::one(), G2::one()) survives only
    when the first attribute is missing, which `is_satisfiable` excludes."""
    ret = (bn.G1_GEN, bn.G2_GEN)
    for i, a in enumerate(conjunction):
        found = next((k for k in sk_a if k[0] == a), None)
        if found is None:
            continue
        ret = (found[1], found[2]) if i == 0 else (bn.g1_add(ret[0], found[1]), bn.g2_add(ret[1], found[2]))
    return ret


def bdabe_setup(rng):
    """bdabe/mod.rs:149-163."""
    g1, g2, p1, p2 = rng.g1(), rng.g2(), rng.g1(), rng.g2()
    y = rng.fr()
    return {"g1": g1, "g2": g2, "p1": p1, "p2": p2, "e_gg_y": bn.gt_pow(bn.pairing(g1, g2), y)}, {"y": y}


def bdabe_authgen(pk, msk, name, rng):
    """bdabe/mod.rs:174-189."""
    alpha = rng.fr()
    beta = (msk["y"] - alpha) % bn.R
    a1 = bn.g1_mul(pk["g1"], alpha)
    a2 = bn.g2_mul(pk["g2"], beta)
    a3 = rng.fr()
    return {"name": name, "a1": a1, "a2": a2, "a3": a3}


def bdabe_keygen(pk, ska, name, rng):
    """bdabe/mod.rs:202-224."""
    r_u = rng.fr()
    return {"sk": {"u1": bn.g1_add(ska["a1"], bn.g1_mul(pk["p1"], r_u)), "u2": bn.g2_add(ska["a2"], bn.g2_mul(pk["p2"], r_u))},
            "pk": {"u": name, "u1": bn.g1_mul(pk["g1"], r_u), "u2": bn.g2_mul(pk["g2"], r_u)},
            "sk_a": []}


def bdabe_request_attribute_pk(pk, ska, attribute):
    """bdabe/mod.rs:234-265; ValueError = the RabeError."""
    if not _from_authority(attribute, ska["name"]):
        raise ValueError("attribute %s is not from_authority() or !is_eligible()" % attribute)
    exp = _attr_exponent(attribute, ska["name"], ska["a3"])
    return {"attr": attribute, "a1": bn.g1_mul(pk["g1"], exp), "a2": bn.g2_mul(pk["g2"], exp), "a3": bn.gt_pow(pk["e_gg_y"], exp)}


def bdabe_request_attribute_sk(pk_u, ska, attribute):
    """bdabe/mod.rs:275-305 (is_eligible is constant true, :470-475)."""
    if not _from_authority(attribute, ska["name"]):
        raise ValueError("attribute %s is not from_authority() or !is_eligible()" % attribute)
    exp = _attr_exponent(attribute, ska["name"], ska["a3"])
    return {"attr": attribute, "au1": bn.g1_mul(pk_u["u1"], exp), "au2": bn.g2_mul(pk_u["u2"], exp)}


def bdabe_encrypt(pk, attr_pks, policy, language, rng):
    """bdabe/mod.rs:317-358.  Draws: the two arguments of `pairing(rng.gen(), rng.gen())` (G1 first), then r_j per term.
    Returns (ct, msg)."""
    tree = pol.parse(policy, language)
    if not pol.policy_in_dnf(tree):
        raise ValueError("Error in bdabe/encrypt: Policy not in DNF.")
    pks = [(k["attr"], k["a1"], k["a2"], k["a3"], bn.GT_ONE) for k in attr_pks]
    terms = pol.json_to_dnf(tree, pks, _DNF_OPS)
    a, b = rng.g1(), rng.g2()
    msg = bn.pairing(a, b)
    j = []
    for t in terms:
        r_j = rng.fr()
        j.append({"attr": list(t[0]), "e1": bn.gt_mul(bn.gt_pow(t[1], r_j), msg), "e2": bn.g1_mul(pk["p1"], r_j),
                  "e3": bn.g2_mul(pk["p2"], r_j), "e4": bn.g1_mul(t[3], r_j), "e5": bn.g2_mul(t[4], r_j)})
    return {"policy": (policy, language), "j": j}, msg


def bdabe_decrypt(sk, ct):
    """bdabe/mod.rs:367-399: the first satisfiable conjunction; Gt::one() when none is (the AES layer then fails)."""
    str_attr = [k["attr"] for k in sk["sk_a"]]
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(str_attr, tree):
        raise ValueError("Error in bdabe/decrypt: attributes in sk do not match policy in ct.")
    sk_a = [(k["attr"], k["au1"], k["au2"]) for k in sk["sk_a"]]
    msg = bn.GT_ONE
    for ct_j in ct["j"]:
        if _is_satisfiable(ct_j["attr"], sk_a):
            s1, s2 = _calc_satisfiable(ct_j["attr"], sk_a)
            msg = bn.gt_mul(bn.gt_mul(bn.gt_mul(ct_j["e1"], bn.pairing(ct_j["e2"], s2)), bn.pairing(s1, ct_j["e3"])),
                            bn.gt_inv(bn.gt_mul(bn.pairing(ct_j["e4"], sk["sk"]["u2"]), bn.pairing(sk["sk"]["u1"], ct_j["e5"]))))
            break
    return msg


def mke08_setup(rng):
    """mke08/mod.rs:130-149."""
    g1, g2, p1, p2 = rng.g1(), rng.g2(), rng.g1(), rng.g2()
    y1, y2 = rng.fr(), rng.fr()
    e = bn.pairing(g1, g2)
    return ({"g1": g1, "g2": g2, "p1": p1, "p2": p2, "e_gg_y1": bn.gt_pow(e, y1), "e_gg_y2": bn.gt_pow(e, y2)},
            {"g1": bn.g1_mul(g1, y1), "g2": bn.g2_mul(g2, y2)})


def mke08_keygen(pk, msk, name, rng):
    """mke08/mod.rs:159-181."""
    mk_u = rng.fr()
    return {"sk": {"g1": bn.g1_add(msk["g1"], bn.g1_mul(pk["p1"], mk_u)), "g2": bn.g2_add(msk["g2"], bn.g2_mul(pk["p2"], mk_u))},
            "pk": {"name": name, "g1": bn.g1_mul(pk["g1"], mk_u), "g2": bn.g2_mul(pk["g2"], mk_u)},
            "sk_a": []}


def mke08_authgen(name, rng):
    """mke08/mod.rs:189-197."""
    return {"name": name, "r": rng.fr()}


def mke08_request_authority_pk(pk, attribute, ska):
    """mke08/mod.rs:207-238."""
    if not _from_authority(attribute, ska["name"]):
        raise ValueError("attribute %s is not from_authority() or !is_eligible()" % attribute)
    exp = _attr_exponent(attribute, ska["name"], ska["r"])
    return {"attr": attribute, "g1": bn.g1_mul(pk["g1"], exp), "g2": bn.g2_mul(pk["g2"], exp),
            "gt1": bn.gt_pow(pk["e_gg_y1"], exp), "gt2": bn.gt_pow(pk["e_gg_y2"], exp)}


def mke08_request_authority_sk(pk_u, attr, ska):
    """mke08/mod.rs:248-278."""
    if not _from_authority(attr, ska["name"]):
        raise ValueError("attribute %s is not from_authority() or !is_eligible()" % attr)
    exp = _attr_exponent(attr, ska["name"], ska["r"])
    return {"attr": attr, "g1": bn.g1_mul(pk_u["g1"], exp), "g2": bn.g2_mul(pk_u["g2"], exp)}


def mke08_encrypt(pk, attr_pks, policy, language, rng):
    """mke08/mod.rs:290-334.  Draws: G1, G2 of msg1, the exponent of msg2, r_j per term.  Returns (ct, msg = msg1 * msg2)."""
    tree = pol.parse(policy, language)
    if not pol.policy_in_dnf(tree):
        raise ValueError("Error in mke08/encrypt: policy is not in dnf")
    pks = [(k["attr"], k["g1"], k["g2"], k["gt1"], k["gt2"]) for k in attr_pks]
    terms = pol.json_to_dnf(tree, pks, _DNF_OPS)
    a, b = rng.g1(), rng.g2()
    msg1 = bn.pairing(a, b)
    msg2 = bn.gt_pow(msg1, rng.fr())
    msg = bn.gt_mul(msg1, msg2)
    e = []
    for t in terms:
        r_j = rng.fr()
        e.append({"str": list(t[0]), "j1": bn.gt_mul(bn.gt_pow(t[1], r_j), msg1), "j2": bn.gt_mul(bn.gt_pow(t[2], r_j), msg2),
                  "j3": bn.g1_mul(pk["p1"], r_j), "j4": bn.g2_mul(pk["p2"], r_j), "j5": bn.g1_mul(t[3], r_j), "j6": bn.g2_mul(t[4], r_j)})
    return {"policy": (policy, language), "e": e}, msg


def mke08_decrypt(sk, ct):
    """mke08/mod.rs:343-380."""
    attr_str = [k["attr"] for k in sk["sk_a"]]
    tree = pol.parse(ct["policy"][0], ct["policy"][1])
    if not pol.traverse_policy(attr_str, tree):
        raise ValueError("Error in mke08/decrypt: attributes in sk do not match policy in ct.")
    sk_a = [(k["attr"], k["g1"], k["g2"]) for k in sk["sk_a"]]
    msg = bn.GT_ONE
    for e_j in ct["e"]:
        if _is_satisfiable(e_j["str"], sk_a):
            s1, s2 = _calc_satisfiable(e_j["str"], sk_a)
            msg = bn.gt_mul(bn.gt_mul(bn.gt_mul(bn.gt_mul(e_j["j1"], e_j["j2"]), bn.pairing(e_j["j3"], s2)), bn.pairing(s1, e_j["j4"])),
                            bn.gt_inv(bn.gt_mul(bn.pairing(e_j["j5"], sk["sk"]["g2"]), bn.pairing(sk["sk"]["g1"], e_j["j6"]))))
            break
    return msg
