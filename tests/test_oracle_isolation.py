"""The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py may touch it.  Guard the rule: no module of the product, no example and no script imports it."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def imports_of(path):
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom):
            yield node.module or ""


def python_files(*dirs):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith(".py"):
                    yield os.path.join(base, f)


def test_product_examples_and_scripts_do_not_import_the_oracle():
    offenders = [p for p in python_files("adaflo_amd", "examples", "scripts")
                 if any(m == "oracle" or m.startswith("oracle.") for m in imports_of(p))]
    assert offenders == []


def test_bench_uses_the_oracle_only_in_the_cpu_baseline():
    with open(os.path.join(ROOT, "bench.py")) as f:
        tree = ast.parse(f.read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and
                       ((getattr(n, "module", None) or "").startswith("oracle") or
                        any(a.name.startswith("oracle") for a in getattr(n, "names", [])))
                       for n in ast.walk(node))
            assert not uses or node.name == "cpu_baseline", node.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any((getattr(n, "module", None) or "").startswith("oracle") for n in top)


def test_the_library_loader_has_no_cpu_fallback(monkeypatch, tmp_path):
    import pytest

    import adaflo_amd._lib as _lib
    assert "oracle" not in open(_lib.__file__).read()
    # a missing shared library is an error, not a fallback
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libadaflo_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
