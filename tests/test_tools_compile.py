"""The measurement / sweep scripts under tools/ are not imported by anything: keep them at least syntactically alive (CPU)."""
import glob
import os
import py_compile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_tool_script_compiles(tmp_path):
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "**", "*.py"), recursive=True))
    assert len(scripts) >= 10
    for path in scripts:
        py_compile.compile(path, cfile=str(tmp_path / (os.path.basename(path) + "c")), doraise=True)
