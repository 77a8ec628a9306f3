"""
The drop-in seam (SURVEY 8b "What calls it"): every import statement the
reference's own modules and tests make against ``filters``, ``sliding_window``
and ``exceptions`` -- listed, with file:line, in
tests/golden/reference_imports.json (made by parsing the reference,
tests/golden/make_golden_imports.py) -- is executed in a fresh interpreter with
``hydrodem_amd/dropin`` first on ``sys.path`` and must bind objects of this
package.  No GPU, no library load: importing the operators is free.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "hydrodem_amd", "dropin")
ROWS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_imports.json"),
                      encoding="utf-8"))


def run(code, *extra_path):
    head = "import sys\n" + "".join(f"sys.path.insert(0, {p!r})\n"
                                    for p in reversed((DROPIN, ROOT) + extra_path))
    return subprocess.run([sys.executable, "-c", head + textwrap.dedent(code)],
                          capture_output=True, text=True, cwd="/")


def test_the_fixture_holds_the_three_caller_lines_of_the_verdict():
    by_where = {r["where"]: r for r in ROWS}
    assert by_where["hydrodem/image_srtm.py:7"]["names"] == [
        "DetectApplyFourier", "BinaryClosing", "GrovesCorrectionsIter"]
    assert by_where["hydrodem/image_hsheds.py:6"]["names"] == [
        "LagoonsDetection", "ClipLagoonsRivers", "ProcessRivers"]
    assert by_where["hydrodem/hydro_dem_process.py:20"]["names"] == [
        "SubtractionFilter", "ProductFilter", "AdditionFilter", "PostProcessingFinal"]


@pytest.mark.parametrize("row", ROWS, ids=[r["where"] for r in ROWS])
def test_reference_import_statement_resolves_to_this_package(row):
    stmt = f"from {row['module']} import ({', '.join(row['names'])})"
    code = stmt + "\n" + "\n".join(
        f"assert {n}.__module__.startswith('hydrodem_amd.'), ({n!r}, {n}.__module__)"
        for n in row["names"]) + "\nprint('ok')\n"
    out = run(code)
    assert out.returncode == 0 and out.stdout.strip() == "ok", f"{stmt}\n{out.stderr}"


def test_all_statements_in_one_interpreter_share_one_set_of_classes():
    lines = []
    for row in ROWS:
        lines.append(f"from {row['module']} import ({', '.join(row['names'])})")
        lines += [f"seen.setdefault({n!r}, {n}); assert seen[{n!r}] is {n}, {n!r}"
                  for n in row["names"]]
    out = run("seen = {}\n" + "\n".join(lines) + "\nimport hydrodem_amd\n"
              "assert seen['GrovesCorrectionsIter'] is hydrodem_amd.GrovesCorrectionsIter\n"
              "assert seen['Filter'] is hydrodem_amd.Filter\nprint(len(seen))\n")
    assert out.returncode == 0 and int(out.stdout) >= 45, out.stderr


def test_other_cguerrero_modules_still_come_from_the_reference_tree(tmp_path):
    """``cguerrero.hydrodem.utils_dem`` (GDAL I/O, not rebuilt) must keep
    resolving to the tree behind the drop-in; a stand-in tree plays the
    reference here (the GPU box has none)."""
    pkg = tmp_path / "cguerrero" / "hydrodem"
    (pkg / "filters").mkdir(parents=True)
    (tmp_path / "cguerrero" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "utils_dem.py").write_text("WHO = 'reference tree'\n")
    (pkg / "sliding_window.py").write_text("raise ImportError('shadowed module was imported')\n")
    (pkg / "filters" / "__init__.py").write_text("raise ImportError('shadowed package')\n")
    out = run("""
        from cguerrero.hydrodem.utils_dem import WHO
        from cguerrero.hydrodem.filters.custom_filters import RouteRivers, QuadraticFilter
        from cguerrero.hydrodem.sliding_window import SlidingWindow
        from cguerrero.hydrodem.exceptions import WindowSizeEvenError
        import hydrodem_amd
        assert WHO == 'reference tree'
        assert QuadraticFilter is hydrodem_amd.QuadraticFilter
        assert SlidingWindow is hydrodem_amd.sliding_window.SlidingWindow
        print('ok')
        """, str(tmp_path))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_custom_filters_namespace_covers_the_reference_module():
    """Every public class the reference's custom_filters binds at module level
    (its own 22 + what custom_filters.py:9-19 imports) exists here."""
    names = sorted({n for r in ROWS if r["where"].startswith("hydrodem/filters/custom_filters.py")
                    for n in r["names"]})
    own = ["MajorityFilter", "ExpandFilter", "RouteRivers", "QuadraticFilter", "CorrectNANValues",
           "IsolatedPoints", "BlanksFourier", "DetectBlanksFourier", "MaskNegatives",
           "MaskPositives", "MaskTallGroves", "MaskFourier", "TidyingLagoons", "LagoonsDetection",
           "GrovesCorrection", "GrovesCorrectionsIter", "ProcessRivers", "ClipLagoonsRivers",
           "FourierInitial", "FourierProcessQuarters", "DetectApplyFourier", "PostProcessingFinal"]
    out = run("import filters.custom_filters as m\n"
              f"missing = [n for n in {names + own!r} if not hasattr(m, n)]\n"
              "assert not missing, missing\nprint('ok')\n")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


REF_TESTS = "/root/reference/cguerrero/tests"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_TESTS, "test_sliding_window.py")),
                    reason="the reference tree is only present in the build container")
def test_the_references_own_sliding_window_tests_behave_the_same_against_the_dropin():
    """`cguerrero/tests/test_sliding_window.py` -- the reference's own 16 test methods with
    their known-answer literals (`tests/constants.py`) -- run unmodified, in fresh
    interpreters, once against the reference tree and once with `dropin/` first on `sys.path`
    (its `from cguerrero.hydrodem.sliding_window import ...` then binds this package's
    classes): the same tests pass and the same tests fail.  (With NumPy 2 two of the
    reference's tests raise on its own code -- `array == array` of different shapes -- and do
    the same here.  Runs where the reference tree is; the committed goldens of
    tests/test_sliding_window.py carry the same windows to the GPU box.)"""
    body = f"""
        import unittest
        sys.path.append({REF_TESTS!r})                 # `from constants import ...`
        sys.dont_write_bytecode = True
        import test_sliding_window as t
        print('BOUND', t.SlidingWindow.__module__)
        suite = unittest.defaultTestLoader.loadTestsFromModule(t)
        result = unittest.TextTestRunner(verbosity=0, stream=open(os.devnull, 'w')).run(suite)
        bad = sorted(str(test).split()[0] for test, _ in result.failures + result.errors)
        print('RAN', result.testsRun, 'BAD', ','.join(bad))
        """
    mine = run("import os\n" + textwrap.dedent(body))
    ref_head = ("import sys, os\nsys.path[:0] = ['/root/reference', "
                "'/root/reference/cguerrero/hydrodem']\n")
    theirs = subprocess.run([sys.executable, "-c", ref_head + textwrap.dedent(body)],
                            capture_output=True, text=True, cwd="/")
    assert mine.returncode == 0 and theirs.returncode == 0, mine.stderr + theirs.stderr
    pick = lambda out, key: [ln for ln in out.stdout.splitlines() if ln.startswith(key)][0]
    assert pick(mine, "BOUND") == "BOUND hydrodem_amd.sliding_window"
    assert pick(theirs, "BOUND") == "BOUND cguerrero.hydrodem.sliding_window"
    assert pick(mine, "RAN") == pick(theirs, "RAN")
    ran, bad = pick(mine, "RAN").split(" BAD ")
    assert ran == "RAN 16" and len([b for b in bad.split(",") if b]) <= 2


REF_SLIDING = "/root/reference/cguerrero/hydrodem/sliding_window.py"


@pytest.mark.skipif(not os.path.exists(REF_SLIDING),
                    reason="the reference tree is only present in the build container")
def test_the_88_doctest_examples_of_the_references_sliding_window_hold_for_the_dropin():
    """The known-answer literals of SURVEY 8c: every `>>>` example in the docstrings of the
    reference's `sliding_window.py` (read as text, not imported), executed with `dropin/`
    first on `sys.path` -- `from sliding_window import SlidingWindow` in the examples binds
    this package's class -- prints exactly what the docstring says."""
    code = f"""
        import ast, doctest, io
        tree = ast.parse(open({REF_SLIDING!r}).read())
        docs = [(getattr(n, 'name', 'module'), ast.get_docstring(n, clean=False))
                for n in ast.walk(tree) if isinstance(n, (ast.Module, ast.ClassDef, ast.FunctionDef))]
        runner = doctest.DocTestRunner(verbose=False,
                                       optionflags=doctest.NORMALIZE_WHITESPACE | doctest.ELLIPSIS)
        sink = io.StringIO()
        for name, doc in docs:
            if doc and '>>>' in doc:
                runner.run(doctest.DocTestParser().get_doctest(doc, {{}}, name, None, 0), out=sink.write)
        import sliding_window
        assert sliding_window.SlidingWindow.__module__ == 'hydrodem_amd.sliding_window'
        print(runner.tries, runner.failures)
        assert runner.tries == 88 and runner.failures == 0, sink.getvalue()[-3000:]
        """
    out = run(code)
    assert out.returncode == 0, out.stdout + out.stderr
