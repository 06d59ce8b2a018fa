"""DESIGN.md's measurement tables are generated from the tracked files under profiles/r4/ (tools/design_tables.py): the document
must contain exactly what the generator prints today, so a figure cannot drift from the file it cites."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_tables_are_the_generators_output():
    import design_tables
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert design_tables.render(text) == text, "run `python tools/design_tables.py --write`"
    for name in design_tables.BLOCKS:
        assert f"<!-- BEGIN {name} -->" in text


def test_every_profile_file_the_design_cites_exists():
    import re
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    cited = set(re.findall(r"`(profiles/r[0-9]+/[\w.{},*-]+)`", text))
    missing = [c for c in cited if "{" not in c and "*" not in c and not os.path.exists(os.path.join(ROOT, c))]
    assert not missing, missing
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for c in set(re.findall(r"profiles/r[0-9]+/[\w.]+\.json", bench)):
        assert os.path.exists(os.path.join(ROOT, c)), c
