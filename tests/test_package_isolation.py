"""The product package must not reach into the checker: no module under rabe_amd/ may import `oracle` or `tests` (nor `benchkit`,
whose CPU-baseline legs time the oracle).  An AST scan, so that a lazy import inside a function body is caught too, plus a scan of
the C / C++ / HIP sources for an #include that leaves rabe_amd/csrc and include/."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rabe_amd")
FORBIDDEN = ("oracle", "tests", "benchkit")


def _py_files():
    for root, _dirs, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(root, f)


def _imports(path):
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield node.lineno, a.name
        elif isinstance(node, ast.ImportFrom):
            if node.level == 0 and node.module:
                yield node.lineno, node.module
        elif isinstance(node, ast.Call):                       # importlib.import_module("oracle...") / __import__("oracle")
            fn = node.func
            name = fn.attr if isinstance(fn, ast.Attribute) else getattr(fn, "id", "")
            if name in ("import_module", "__import__") and node.args and isinstance(node.args[0], ast.Constant) \
                    and isinstance(node.args[0].value, str):
                yield node.lineno, node.args[0].value


def test_no_module_of_the_package_imports_the_checker():
    bad = []
    n = 0
    for path in _py_files():
        n += 1
        for line, mod in _imports(path):
            if mod.split(".")[0] in FORBIDDEN:
                bad.append("%s:%d imports %s" % (os.path.relpath(path, ROOT), line, mod))
    assert n >= 8, "package not found"
    assert not bad, "\n".join(bad)


def test_native_sources_include_nothing_of_the_checker():
    bad = []
    n = 0
    for root, _dirs, files in os.walk(os.path.join(PKG, "csrc")):
        for f in files:
            if not f.endswith((".h", ".hip", ".cpp", ".c")):
                continue
            n += 1
            path = os.path.join(root, f)
            for i, text in enumerate(open(path, errors="replace"), 1):
                m = re.match(r'\s*#\s*include\s*["<]([^">]+)[">]', text)
                if m and re.search(r"(^|/)(oracle|tests)/", m.group(1)):
                    bad.append("%s:%d includes %s" % (os.path.relpath(path, ROOT), i, m.group(1)))
    assert n >= 10
    assert not bad, "\n".join(bad)


def test_the_package_docstring_claim_holds_textually():
    """`grep -rn "oracle\\|from tests" rabe_amd/*.py rabe_amd/schemes` finds at most prose that says what is NOT done"""
    for path in _py_files():
        for i, text in enumerate(open(path), 1):
            assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", text), "%s:%d" % (path, i)
