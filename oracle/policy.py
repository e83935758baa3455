"""Policy language, LSSS matrix, secret sharing and pruning -- TEST ORACLE restatement.

Test infrastructure only (see oracle/bn254.py header).  Each function cites the
reference file:line (relative to /root/reference) whose behaviour it restates.
These host-side pieces ARE pinned by the reference's own fixed-answer tests
(msp.rs:157-199, secretsharing/mod.rs:286-324, pest/mod.rs:118-149,
tools/mod.rs:76-129) -- see tests/test_oracle_policy.py.

Tree representation (mirrors `PolicyValue`, src/utils/policy/pest/mod.rs:33-37):
    ("leaf", name, col)      PolicyValue::String((name, col))
    ("and", [children])      Object((And, Array(children)))
    ("or",  [children])      Object((Or,  Array(children)))
"""
from . import bn254 as bn

JSON = "json"
HUMAN = "human"


class PolicyError(Exception):
    """A pest parse error (`RabeError` from src/error.rs:40-48)."""


class PolicyPanic(Exception):
    """Places where the reference panics (msp.rs:121,133; secretsharing/mod.rs:107-113,167,187)."""


# ----------------------------------------------------------------------------- tiny PEG helper

class _Src:
    def __init__(self, s):
        self.s = s
        self.n = len(s)

    def col(self, pos):
        """pest `Position::line_col().1`: 1-based count of chars since the last '\\n'."""
        nl = self.s.rfind("\n", 0, pos)
        return pos - nl          # nl == -1 -> pos + 1

    def skip(self, pos):
        """implicit WHITESPACE / COMMENT between tokens of non-atomic rules."""
        s, n = self.s, self.n
        while pos < n:
            c = s[pos]
            if c in " \t\r\n":
                pos += 1
            elif s.startswith("/*", pos):
                end = s.find("*/", pos + 2)
                if end < 0:
                    return pos
                pos = end + 2
            else:
                break
        return pos

    def lit(self, pos, *alts):
        for a in alts:
            if self.s.startswith(a, pos):
                return pos + len(a)
        return None


_HEX = "0123456789abcdefABCDEF"


def _string(src, pos):
    """`string = ${QUOTE ~ inner ~ QUOTE}`, `inner = @{ char* }` (json.policy.pest:41-47,
    human.policy.pest:22-28).  Returns (end, (name, col)) or None."""
    s = src.s
    if pos >= src.n or s[pos] != '"':
        return None
    i = pos + 1
    start = i
    while i < src.n:
        c = s[i]
        if c == '"':
            break
        if c == "\\":
            if i + 1 < src.n and s[i + 1] in '"\\/bfnrt':
                i += 2
                continue
            hexd = s[i + 2:i + 6]
            if i + 1 < src.n and s[i + 1] == "u" and len(hexd) == 4 and all(ch in _HEX for ch in hexd):
                i += 6
                continue
            return None
        i += 1
    if i >= src.n or s[i] != '"':
        return None
    return i + 1, ("leaf", s[start:i], src.col(start))


def _number(src, pos):
    """`number` rule (json.policy.pest:48-53).  The reference then calls
    `into_inner().next().unwrap()` on this atomic rule (pest/json.rs:14-17) which panics."""
    s = src.s
    i = pos
    if i < src.n and s[i] == "-":
        i += 1
    if i < src.n and s[i] == "0":
        i += 1
    elif i < src.n and s[i] in "123456789":
        while i < src.n and s[i].isdigit() and s[i].isascii():
            i += 1
    else:
        return None
    if i < src.n and s[i] == ".":
        i += 1
        while i < src.n and s[i].isdigit() and s[i].isascii():
            i += 1
    if i < src.n and s[i] in "eE":
        j = i + 1
        if j < src.n and s[j] in "+-":
            j += 1
        if j < src.n and s[j].isdigit():
            while j < src.n and s[j].isdigit() and s[j].isascii():
                j += 1
            i = j
    return i


_AND = ("and", "AND", "&&")
_OR = ("or", "OR", "||")


def _quoted_or_bare(src, pos, alts):
    """`x | QUOTE ~ x ~ QUOTE` in a non-atomic rule (whitespace allowed inside the quotes)."""
    e = src.lit(pos, *alts)
    if e is not None:
        return e
    if pos < src.n and src.s[pos] == '"':
        p = src.skip(pos + 1)
        e = src.lit(p, *alts)
        if e is not None:
            p = src.skip(e)
            if p < src.n and src.s[p] == '"':
                return p + 1
    return None


# ----------------------------------------------------------------------------- JSON grammar (src/json.policy.pest)

def _json_node(src, pos):
    if pos >= src.n or src.s[pos] != "{":
        return None
    p = src.skip(pos + 1)
    p = _quoted_or_bare(src, p, ("name", "NAME"))
    if p is None:
        return None
    p = src.skip(p)
    if p >= src.n or src.s[p] != ":":
        return None
    p = src.skip(p + 1)
    # alternative 1: value = string | number
    r = _string(src, p)
    if r is not None:
        q = src.skip(r[0])
        if q < src.n and src.s[q] == "}":
            return q + 1, r[1]
    else:
        e = _number(src, p)
        if e is not None:
            q = src.skip(e)
            if q < src.n and src.s[q] == "}":
                raise PolicyPanic("pest/json.rs:15 unwrap on atomic `number` rule")
    # alternatives 2, 3: and / or
    for kind, alts in (("and", _AND), ("or", _OR)):
        e = _quoted_or_bare(src, p, alts)
        if e is None:
            continue
        q = src.skip(e)
        if q >= src.n or src.s[q] != ",":
            continue
        q = src.skip(q + 1)
        q = _quoted_or_bare(src, q, ("children", "CHILDREN"))
        if q is None:
            continue
        q = src.skip(q)
        if q >= src.n or src.s[q] != ":":
            continue
        q = src.skip(q + 1)
        if q >= src.n or src.s[q] != "[":
            continue
        q = src.skip(q + 1)
        children = []
        if q < src.n and src.s[q] == "]":
            q += 1
        else:
            ok = True
            while True:
                r = _json_node(src, q)
                if r is None:
                    ok = False
                    break
                children.append(r[1])
                q = src.skip(r[0])
                if q < src.n and src.s[q] == ",":
                    q = src.skip(q + 1)
                    continue
                break
            if not ok or q >= src.n or src.s[q] != "]":
                continue
            q += 1
        q = src.skip(q)
        if q < src.n and src.s[q] == "}":
            return q + 1, (kind, children)
    return None


# ----------------------------------------------------------------------------- human grammar (src/human.policy.pest)

def _human_term(src, pos):
    # term = value | "(" node ")" ; value = string | number | BRACEOPEN node BRACECLOSE
    r = _string(src, pos)
    if r is not None:
        return r
    e = _number(src, pos)
    if e is not None:
        raise PolicyPanic("pest/human.rs:15 unwrap on atomic `number` rule")
    if pos < src.n and src.s[pos] in "([{":
        p = src.skip(pos + 1)
        r = _human_node(src, p)
        if r is not None:
            q = src.skip(r[0])
            if q < src.n and src.s[q] in ")]}":
                return q + 1, r[1]
    return None


def _human_chain(src, pos, kind, alts):
    r = _human_term(src, pos)
    if r is None:
        return None
    children = [r[1]]
    p = r[0]
    while True:
        q = src.skip(p)
        e = _quoted_or_bare(src, q, alts)
        if e is None:
            break
        q = src.skip(e)
        r = _human_term(src, q)
        if r is None:
            break
        children.append(r[1])
        p = r[0]
    if len(children) < 2:
        return None
    return p, (kind, children)


def _human_node(src, pos):
    # node = and | or | term   (ordered choice, PEG: commits to the first alternative that matches)
    r = _human_chain(src, pos, "and", _AND)
    if r is not None:
        return r
    r = _human_chain(src, pos, "or", _OR)
    if r is not None:
        return r
    return _human_term(src, pos)


def parse(policy, language=JSON):
    """`parse(policy, language)` src/utils/policy/pest/mod.rs:40-66 (content = SOI ~ node ~ EOI)."""
    src = _Src(policy)
    p = src.skip(0)
    r = (_json_node if language == JSON else _human_node)(src, p)
    if r is None:
        raise PolicyError("could not parse policy")
    if src.skip(r[0]) != src.n:
        raise PolicyError("trailing input after policy")
    return r[1]


def serialize_policy(val, language=JSON, parent=None):
    """src/utils/policy/pest/mod.rs:68-114."""
    kind = val[0]
    if language == JSON:
        if kind == "leaf":
            return '{"name": "%s"}' % val[1]
        inner = '"children": [%s]' % ", ".join(serialize_policy(c, language) for c in val[1])
        return '{"name": "%s", %s}' % (kind, inner)
    if kind == "leaf":
        return val[1]
    return "(%s)" % (" %s " % kind).join(serialize_policy(c, language) for c in val[1])


# ----------------------------------------------------------------------------- MSP (Lewko-Waters), src/utils/policy/msp.rs

def calculate_msp(p):
    """`calculate_msp` msp.rs:78-99 + `lw` msp.rs:102-147.  Returns (m, pi, c)."""
    m, pi = [], []
    state = {"c": 1}

    def lw(node, v, parent):
        if node[0] == "leaf":
            m.insert(0, list(v))                       # msp.rs:107
            pi.insert(0, node[1])
            return True
        children = node[1]
        if len(children) < 2:
            raise PolicyPanic("lw: policy with just a single attribute is not allowed")     # msp.rs:121
        if node[0] == "or":
            ret = True
            for ch in children:
                ret &= lw(ch, v, "or")
            return ret
        if len(children) != 2:
            raise PolicyPanic("lw: Invalid policy. Number of arguments under AND != 2")      # msp.rs:133
        c = state["c"]
        right = list(v) + [0] * (c - len(v)) if len(v) <= c else list(v[:c])                 # Vec::resize
        right.append(1)
        left = [0] * c + [-1]
        state["c"] = c + 1
        return lw(children[0], right, "and") and lw(children[1], left, "and")                # msp.rs:140

    if not lw(p, [1], None):
        raise PolicyError("lewko waters algorithm failed =(")
    c = state["c"]
    m = [row + [0] * (c - len(row)) if len(row) <= c else row[:c] for row in m]
    # permutation::sort is a stable sort of indices by key (msp.rs:93-95)
    order = sorted(range(len(pi)), key=lambda i: pi[i])
    return [m[i] for i in order], [pi[i] for i in order], c


# ----------------------------------------------------------------------------- tools, src/utils/tools/mod.rs

def traverse_policy(attr, node):
    """tools/mod.rs:31-61 with policy_type = Leaf at the root."""
    if len(attr) == 0:
        return False
    if node[0] == "leaf":
        return node[1] in attr
    if node[0] == "and":
        ret = True
        for ch in node[1]:
            ret &= traverse_policy(attr, ch)
        return ret
    ret = False
    for ch in node[1]:
        ret |= traverse_policy(attr, ch)
    return ret


def is_negative(attr):
    return attr[:1] == "!"          # tools/mod.rs:6-9


def node_index(node):
    return "%s_%d" % (node[1], node[2])      # secretsharing/mod.rs:74-76


def remove_index(s):
    return s.split("_")[0]                   # secretsharing/mod.rs:77-80


# ----------------------------------------------------------------------------- secret sharing, src/utils/secretsharing/mod.rs

def polynomial(coeff, x):
    """secretsharing/mod.rs:215-221 (x.pow(i) with pow(_, 0) = 1)."""
    share = 0
    for i, c in enumerate(coeff):
        share = (share + c * pow(x, i, bn.R)) % bn.R
    return share


def gen_shares(secret, k, n, rng):
    """secretsharing/mod.rs:124-141; `rng.fr()` replaces each `rng.gen()` in draw order."""
    shares = []
    if k <= n:
        a = [secret % bn.R]
        for _ in range(1, k):
            a.append(rng.fr())
        for i in range(n + 1):
            shares.append(polynomial(a, i))
    return shares


def gen_shares_policy(secret, node, rng):
    """secretsharing/mod.rs:82-122.  Returns [(name_col, share)] in DFS order."""
    if node[0] == "leaf":
        return [(node_index(node), secret % bn.R)]
    children = node[1]
    n = len(children)
    k = n if node[0] == "and" else 1
    shares = gen_shares(secret, k, n, rng)
    out = []
    for i in range(n):
        out.extend(gen_shares_policy(shares[i + 1], children[i], rng))
    return out


def recover_coefficients(points):
    """Lagrange at 0, secretsharing/mod.rs:60-72."""
    out = []
    for i in points:
        res = 1
        for j in points:
            if i != j:
                res = res * ((0 - j) * bn.fr_inv(i - j)) % bn.R
        out.append(res % bn.R)
    return out


def calc_coefficients(node, coeff=1):
    """secretsharing/mod.rs:9-57.  Returns [(name_col, coeff)] for ALL leaves in DFS order."""
    if node[0] == "leaf":
        return [(node_index(node), coeff % bn.R)]
    children = node[1]
    if node[0] == "and":
        lag = recover_coefficients(list(range(1, len(children) + 1)))
    else:
        lag = [1] * len(children)
    out = []
    for i, ch in enumerate(children):
        out.extend(calc_coefficients(ch, coeff * lag[i] % bn.R))
    return out


def calc_pruned(attr, node):
    """secretsharing/mod.rs:143-199.  Returns (match, [(name, name_col)])."""
    if node[0] == "leaf":
        if node[1] in attr:
            return True, [(node[1], node_index(node))]
        return False, []
    children = node[1]
    if len(children) < 2:
        raise PolicyPanic("Invalid policy (%s with just a single child)" % node[0].upper())
    if node[0] == "and":
        ok, acc = True, []
        for ch in children:
            found, lst = calc_pruned(attr, ch)
            ok = ok and found
            if ok:
                acc.extend(lst)
        return (ok, acc if ok else [])
    for ch in children:
        found, lst = calc_pruned(attr, ch)
        if found:
            return True, lst
    return False, []


# ----------------------------------------------------------------------------- DNF policies (bdabe, mke08)

def policy_in_dnf(node, conjunction=False):
    """src/utils/policy/dnf.rs:203-243.  An OR below an AND is the only thing it rejects (an AND below an AND passes here and
    fails in `json_to_dnf`)."""
    kind = node[0]
    if kind == "leaf":
        return True
    if kind == "and":
        ret = True
        for child in node[1]:
            ret &= policy_in_dnf(child, True)
        return ret
    if conjunction:            # Array under Or while inside a conjunction (:225-227)
        return False
    ret = True
    for child in node[1]:
        ret &= policy_in_dnf(child, conjunction)
    return ret


def _dnf(terms, pks, node, i, parent, ops):
    """src/utils/policy/dnf.rs:106-183.  `terms` = list of [attrs, gt1, gt2, g1, g2]; `pks` = public attribute keys as
    (attr, g1, g2, gt1, gt2); `ops` = (gt_mul, g1_add, g2_add).  Child k of an OR is sent to term index 2k (`i + i` with the
    loop's shadowing `i`, :162-164) while a missing index APPENDS (:133-141) -- both restated as they are."""
    gt_mul, g1_add, g2_add = ops
    kind = node[0]
    if kind == "leaf":
        for pak in pks:
            if pak[0] == node[1]:
                if len(terms) > i:
                    t = terms[i]
                    t[0].append(pak[0])
                    t[1] = gt_mul(t[1], pak[3])
                    t[2] = gt_mul(t[2], pak[4])
                    t[3] = g1_add(t[3], pak[1])
                    t[4] = g2_add(t[4], pak[2])
                else:
                    terms.append([[pak[0]], pak[3], pak[4], pak[1], pak[2]])
        return True
    # Object((kind, Array(children)))
    if parent is None:
        arr_parent = kind
    elif parent == "or":
        if kind != "and":
            return False
        arr_parent = "and"
    else:                       # an inner node under an AND: only Leaf objects pass there and the parser makes none (:172-177)
        return False
    ret = True
    if arr_parent == "and":
        for child in node[1]:
            ret = ret and _dnf(terms, pks, child, i, "and", ops)
    else:
        for idx, child in enumerate(node[1]):
            ret = ret and _dnf(terms, pks, child, idx + idx, "or", ops)
    return ret


def json_to_dnf(node, pks, ops):
    """src/utils/policy/dnf.rs:186-201: the terms, stably sorted by their number of attributes; PolicyPanic where the
    callers' `.unwrap()` meets the Err."""
    terms = []
    if not _dnf(terms, pks, node, 0, None, ops):
        raise PolicyPanic("Error in json_to_dnf: could not parse policy as DNF")
    terms.sort(key=lambda t: len(t[0]))
    return terms
