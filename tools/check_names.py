#!/usr/bin/env python3
"""A small undefined-name check for a Python file (no pyflakes in this image): every name a function loads must be a parameter or an
assignment of that function or of an enclosing one, a module-level name, or a builtin.  usage: python tools/check_names.py FILE..."""
import ast, builtins, sys


def bound_names(node):
    """names bound directly in this function / module body (not in nested functions or classes)"""
    names = set()

    def visit(n, top=True):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Lambda)) and not top:
            if not isinstance(n, ast.Lambda):
                names.add(n.name)
            return
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                names.add(arg.arg)
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                names.add((al.asname or al.name).split(".")[0])
        if isinstance(n, ast.Global):
            names.update(n.names)
        if isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            for g in n.generators:
                for t in ast.walk(g.target):
                    if isinstance(t, ast.Name):
                        names.add(t.id)
        for c in ast.iter_child_nodes(n):
            visit(c, False)
    visit(node)
    return names


def check(path):
    tree = ast.parse(open(path).read(), path)
    problems = []

    def walk(node, scopes):
        here = scopes + [bound_names(node)]
        for n in ast.iter_child_nodes(node):
            scan(n, here)

    def scan(n, scopes):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            for d in getattr(n, "decorator_list", []):
                scan(d, scopes)
            for d in n.args.defaults + [d for d in n.args.kw_defaults if d is not None]:
                scan(d, scopes)
            walk(n, scopes)
            return
        if isinstance(n, ast.ClassDef):
            walk(n, scopes[:1] + scopes[1:])          # (class bodies see the enclosing scopes; their own names are not visible to methods)
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
            if not any(n.id in s for s in scopes) and not hasattr(builtins, n.id):
                problems.append((n.lineno, n.id))
        for c in ast.iter_child_nodes(n):
            scan(c, scopes)
    walk(tree, [])
    return problems


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for line, name in sorted(set(check(p))):
            print("%s:%d: undefined name %s" % (p, line, name))
            bad += 1
    sys.exit(1 if bad else 0)
