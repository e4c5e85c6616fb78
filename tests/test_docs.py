"""DESIGN.md quotes numbers from the round's tracked evidence under profiles/: the block
`<!-- tracked-numbers rNN ... -->` lists (file | JSON path | value) for every one of them, and this test holds
each to the tracked file, so that the text cannot drift from what was measured."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lookup(obj, path):
    for key in path.split("."):
        obj = obj[key]
    return obj


def test_design_numbers_match_the_tracked_profiles():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- tracked-numbers (r\d+):[^>]*-->\s*```\n(.*?)```", text, re.S)
    assert m, "DESIGN.md has no tracked-numbers block"
    rows = [l.split("|") for l in m.group(2).strip().splitlines()]
    assert len(rows) >= 10
    for fname, path, quoted in rows:
        fname, path, quoted = fname.strip(), path.strip(), quoted.strip()
        assert fname.startswith("profiles/") and os.path.exists(os.path.join(ROOT, fname)), fname
        data = json.loads(open(os.path.join(ROOT, fname)).read().strip().splitlines()[-1]) \
            if fname.endswith("bench.json") or fname.endswith("config5.json") else json.load(open(os.path.join(ROOT, fname)))
        actual = _lookup(data, path)
        decimals = len(quoted.split(".")[1]) if "." in quoted else 0
        assert round(float(actual), decimals) == float(quoted), (fname, path, quoted, actual)
    # and the round's headline figures appear in the prose as the block has them
    by = {(r[0].strip(), r[1].strip()): r[2].strip() for r in rows}
    frac = by[("profiles/%s_bench.json" % m.group(1), "roofline.frac")]
    assert frac in text.replace("**", ""), "the tracked roofline.frac is not what the prose says"


def test_profiles_named_in_design_exist():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for name in set(re.findall(r"profiles/((?:r0\d|history)_[A-Za-z0-9_.{},]+?\.(?:json|md|txt|csv))", text)):
        if "{" in name:
            continue
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name


def test_readme_quotes_the_tracked_headline():
    """README.md's measured paragraph is generated from the same tracked files (tools/design_numbers.py)."""
    text = open(os.path.join(ROOT, "README.md")).read()
    m = re.search(r"<!-- gen:measured -->(.*?)<!-- /gen -->", text, re.S)
    assert m
    rnd = re.search(r"profiles/(r\d+)_bench.json", m.group(1)).group(1)
    line = json.loads(open(os.path.join(ROOT, "profiles", rnd + "_bench.json")).read().strip().splitlines()[-1])
    assert "%.0f Msamples/s" % line["value"] in m.group(1) and "%.3f" % line["roofline"]["frac"] in m.group(1)


def test_design_is_short_and_narrow():
    """The design document describes the current design (at most 400 lines of at most 120 columns,
    tables excepted); history lives under profiles/."""
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 400, len(lines)
    wide = [i + 1 for i, l in enumerate(lines) if len(l) > 120 and not l.startswith("|") and not l.startswith("profiles/")]
    assert not wide, wide
