"""
Regenerates tests/golden/reference_imports.json.  RUNS ONLY IN THE BUILD
CONTAINER: parses (does not import) every module of the reference and its test
suite and lists the names each one imports from the modules the drop-in
provides -- ``filters``, ``filters.*``, ``sliding_window``, ``exceptions``, flat
or under ``cguerrero.hydrodem``.  The list is data (module + names + where), the
contract `tests/test_dropin_seam.py` executes against ``hydrodem_amd/dropin``.

    python tests/golden/make_golden_imports.py
"""
import ast
import glob
import json
import os

REF = "/root/reference/cguerrero"
HERE = os.path.dirname(os.path.abspath(__file__))
PROVIDED = ("filters", "sliding_window", "exceptions")


def provided(module):
    flat = module[len("cguerrero.hydrodem."):] if module.startswith("cguerrero.hydrodem.") else module
    return flat.split(".")[0] in PROVIDED


def main():
    rows = []
    for path in sorted(glob.glob(os.path.join(REF, "hydrodem", "**", "*.py"), recursive=True) +
                       glob.glob(os.path.join(REF, "tests", "*.py"))):
        tree = ast.parse(open(path, encoding="utf-8").read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.level == 0 \
                    and provided(node.module):
                rows.append({"where": f"{os.path.relpath(path, REF)}:{node.lineno}",
                             "module": node.module,
                             "names": [a.name for a in node.names]})
    with open(os.path.join(HERE, "reference_imports.json"), "w", encoding="utf-8") as fh:
        json.dump(rows, fh, indent=1)
    for r in rows:
        print(r["where"], r["module"], len(r["names"]))


if __name__ == "__main__":
    main()
