"""The driver's entry points are part of the suite: `__graft_entry__.smoke()` runs on the GPU box after the tests, and round 4 showed that a rename
in the package can break it while 198 GPU tests stay green (VERDICT r4 next #1b).  CPU: every attribute smoke() reads from a DecisionPicture exists
on the class; GPU: smoke() itself."""
import ast
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_reads_only_attributes_the_decision_picture_has():
    from turingcodec_amd.decisions import DecisionPicture
    import inspect
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    tree = ast.parse(src)
    used = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "dp"}
    assert used, "smoke() no longer drives a DecisionPicture: update this test"
    body = inspect.getsource(DecisionPicture)
    for base in DecisionPicture.__mro__[1:-1]:
        body += inspect.getsource(base)
    missing = [a for a in sorted(used) if not hasattr(DecisionPicture, a) and f"self.{a} =" not in body and f"self.{a}," not in body and f", self.{a} =" not in body]
    assert not missing, f"__graft_entry__.smoke() reads dp.{missing} which DecisionPicture never sets"


@pytest.mark.gpu
def test_smoke_entry_point_runs():
    import __graft_entry__
    __graft_entry__.smoke()
