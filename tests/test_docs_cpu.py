"""The documents point at files and symbols that exist (stale references are the first thing a reader trips over)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["README.md", "DESIGN.md", "INTEGRATION.md", "profiles/README.md"]
PATH = re.compile(r"`((?:tests|tools|profiles|oracle|include|lars_image_processing_amd)/[A-Za-z0-9_./-]+?\.(?:py|md|h|hip|cpp|json|csv|txt|sh|npz))(?:::[A-Za-z0-9_]+)?`")
SYMBOL = re.compile(r"`(lars_[dh]_[a-z0-9_]+|lars_comm_[a-z0-9_]+)`")


def test_referenced_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(PATH.findall(text)):
            if "*" in path or "<" in path:
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing


def test_referenced_entry_points_are_declared():
    # the product's header, and the laboratory library's for the experiments the evidence files still name
    header = open(os.path.join(ROOT, "include", "lars_hip.h")).read() + open(os.path.join(ROOT, "include", "lars_lab.h")).read()
    unknown = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for sym in set(SYMBOL.findall(text)):
            if not re.search(r"\b" + re.escape(sym) + r"\b", header):
                unknown.append((doc, sym))
    assert not unknown, unknown


def test_referenced_tests_exist():
    named = re.compile(r"`(?:tests/)?(test_[a-z0-9_]+\.py)::(test_[A-Za-z0-9_]+)`")
    bad = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for fname, func in set(named.findall(text)):
            path = os.path.join(ROOT, "tests", fname)
            if not os.path.exists(path) or ("def " + func) not in open(path).read():
                bad.append((doc, fname, func))
    assert not bad, bad


def test_bare_test_names_exist():
    import glob
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")))
    bad = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"`(test_[a-z0-9_]+)`", text)):
            if "def " + name not in src:
                bad.append((doc, name))
    assert not bad, bad
