"""Minimal undefined-name check (no pyflakes in the image): for every function, names that are loaded but bound neither in the
function (args, assignments, imports, comprehension / with / except targets, nested defs), in an enclosing function, at module level
nor in builtins.  Catches NameErrors in branches the CPU test-suite cannot execute (GPU-only paths).

  python tools/lint_names.py [paths...]      (default: cat_amd bench.py __graft_entry__.py tools tests oracle)"""
import ast
import builtins
import os
import sys


def _args(fn):
    a = fn.args
    return {x.arg for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else [])}


def _local_bindings(body_nodes):
    """Names bound directly in a scope (not inside nested function / class bodies, whose own scopes are visited separately)."""
    out = set()
    stack = list(body_nodes)
    while stack:
        x = stack.pop()
        if isinstance(x, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(x.name)
            stack.extend(x.decorator_list)
            continue          # do not descend: separate scope
        if isinstance(x, ast.Lambda):
            continue
        if isinstance(x, (ast.Import, ast.ImportFrom)):
            for al in x.names:
                out.add((al.asname or al.name).split('.')[0])
        elif isinstance(x, ast.Name) and isinstance(x.ctx, (ast.Store, ast.Del)):
            out.add(x.id)
        elif isinstance(x, ast.ExceptHandler) and x.name:
            out.add(x.name)
        elif isinstance(x, (ast.Global, ast.Nonlocal)):
            out.update(x.names)
        stack.extend(ast.iter_child_nodes(x))
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    base = set(dir(builtins)) | {'__file__', '__name__', '__doc__', '__class__'}
    bad = []

    def visit(nodes, scopes, class_scope=False):
        local = _local_bindings(nodes)
        # class bodies do not form an enclosing scope for the functions defined in them
        chain = scopes + [local]
        stack = list(nodes)
        while stack:
            x = stack.pop()
            if isinstance(x, (ast.FunctionDef, ast.AsyncFunctionDef)):
                for d in x.decorator_list + x.args.defaults + [k for k in x.args.kw_defaults if k is not None]:
                    stack.append(d)
                inner = (scopes if class_scope else chain) + [_args(x)]
                visit(x.body, inner)
                continue
            if isinstance(x, ast.Lambda):
                visit([x.body], (scopes if class_scope else chain) + [_args(x)])
                continue
            if isinstance(x, ast.ClassDef):
                for d in x.decorator_list + x.bases:
                    stack.append(d)
                visit(x.body, chain, class_scope=True)
                continue
            if isinstance(x, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                comp = set()
                for g in x.generators:
                    for y in ast.walk(g.target):
                        if isinstance(y, ast.Name):
                            comp.add(y.id)
                visit(list(ast.iter_child_nodes(x)), chain + [comp])
                continue
            if isinstance(x, ast.Name) and isinstance(x.ctx, ast.Load):
                if x.id not in base and not any(x.id in sc for sc in chain):
                    bad.append((x.lineno, x.id))
            stack.extend(ast.iter_child_nodes(x))

    visit(tree.body, [])
    return sorted(set(bad))


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    targets = sys.argv[1:] or [os.path.join(root, p) for p in ('cat_amd', 'bench.py', '__graft_entry__.py', 'tools', 'tests', 'oracle')]
    files = []
    for t in targets:
        if os.path.isdir(t):
            for d, _, fs in os.walk(t):
                files += [os.path.join(d, f) for f in fs if f.endswith('.py')]
        else:
            files.append(t)
    n = 0
    for f in sorted(files):
        for line, name in check(f):
            print(f'{os.path.relpath(f, root)}:{line}: undefined name {name!r}')
            n += 1
    print(f'{len(files)} files, {n} undefined names')
    return 1 if n else 0


if __name__ == '__main__':
    sys.exit(main())
