"""A static guard for code that only runs next to a GPU (bench.py's N > 1 branches, the drivers' device paths, the visit tools):
every name a function, lambda, comprehension or class body LOADS must be bound somewhere in an enclosing scope, at module level
or in builtins.  It is a rough scope model (no flow analysis), enough to catch a misspelt variable in a branch the CPU suite never
takes -- the kind of error that would otherwise first show up on the GPU box."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bound_in(node):
    """names a function / module body binds anywhere inside it (assignments, imports, defs, handlers, global / nonlocal)"""
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and n is not node:
            out.add(n.name)
        elif isinstance(n, ast.Import):
            out.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ImportFrom):
            out.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
    return out


class _Scopes(ast.NodeVisitor):
    def __init__(self, module):
        self.scopes = [set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__spec__"} | _bound_in(module)]
        self.unbound = []

    def _function(self, node):
        a = node.args
        names = {x.arg for x in a.posonlyargs + a.args + a.kwonlyargs} | ({a.vararg.arg} if a.vararg else set()) | \
                ({a.kwarg.arg} if a.kwarg else set())
        for d in list(a.defaults) + [d for d in a.kw_defaults if d is not None]:
            self.visit(d)
        self.scopes.append(names | _bound_in(node))
        for b in (node.body if isinstance(node.body, list) else [node.body]):
            self.visit(b)
        self.scopes.pop()

    visit_FunctionDef = visit_AsyncFunctionDef = visit_Lambda = _function

    def visit_ClassDef(self, node):
        for b in node.bases + node.decorator_list:
            self.visit(b)
        self.scopes.append(_bound_in(node))
        for b in node.body:
            self.visit(b)
        self.scopes.pop()

    def _comprehension(self, node):
        names = set()
        for g in node.generators:
            names.update(m.id for m in ast.walk(g.target) if isinstance(m, ast.Name))
        self.scopes.append(names)
        self.generic_visit(node)
        self.scopes.pop()

    visit_ListComp = visit_SetComp = visit_DictComp = visit_GeneratorExp = _comprehension

    def visit_Name(self, node):
        if isinstance(node.ctx, ast.Load) and not any(node.id in s for s in self.scopes):
            self.unbound.append((node.lineno, node.id))


def unbound_names(path):
    tree = ast.parse(open(path).read(), filename=path)
    v = _Scopes(tree)
    v.visit(tree)
    return v.unbound


def test_no_unbound_names_in_python_sources():
    files = [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py")]
    for pattern in ("param_amd/*.py", "param_amd/*/*.py", "param_amd/*/*/*.py", "oracle/*.py", "tools/*.py", "tests/*.py",
                    "tests/golden/*.py", "examples/*/*.py"):
        files += sorted(glob.glob(os.path.join(ROOT, pattern)))
    assert len(files) > 80
    bad = {os.path.relpath(f, ROOT): u for f in files for u in [unbound_names(f)] if u}
    assert not bad, bad


def test_the_guard_sees_a_misspelt_name(tmp_path):
    p = tmp_path / "m.py"
    p.write_text("import os\n\ndef f(a):\n    if a:\n        return oss.getcwd()\n    return [x for x in range(a)] + [lambda q: q + b]\n")
    assert unbound_names(str(p)) == [(5, "oss"), (6, "b")]


def test_no_experiment_branches_in_the_product_kernels():
    """round 4's ablation switches (PM_FWD_EXP, PM_FWD_STORE, PM_MAIN_EXP, PM_UNIQ_EXP, PM_LB_EXP: `#if` branches inside the hot kernels
    that dropped loads / stores / ranking for timing experiments) are gone from the product sources, and nothing on a launch path
    reads the environment per call (capi.hip reads its switches once per process)"""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "param_amd", "csrc", "*")):
        if not path.endswith((".hip", ".inc", ".h")):
            continue
        text = open(path).read()
        assert not re.search(r"PM_[A-Z0-9_]*EXP\b", text), path
        assert "PM_FWD_STORE" not in text and "PM_FIX_TRACE" not in text, path
        if not path.endswith("capi.hip"):
            assert not re.search(r"(?<![a-z_])printf\s*\(", text), f"{path}: device-side printf"
    capi = open(os.path.join(root, "param_amd", "csrc", "capi.hip")).read()
    body = capi[capi.index("int make_params("):]
    assert "getenv" not in body, "capi.hip: getenv on a launch path"
