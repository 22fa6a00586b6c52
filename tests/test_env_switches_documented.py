"""CPU: every PWG_* environment variable the package, the library sources or bench.py read is listed in
INTEGRATION.md section 4 (and nothing is listed that no longer exists)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_vars():
    files = [os.path.join(ROOT, "bench.py")]
    for ext in ("py", "hip", "h"):
        files += glob.glob(os.path.join(ROOT, "parallelwavegan_amd", "**", f"*.{ext}"), recursive=True)
    pat = re.compile(r'(?:getenv\(\s*|environ(?:\.get|\.setdefault|\.pop)?\s*[\(\[]\s*)"(PWG_[A-Z0-9_]+)"')
    found = set()
    for f in files:
        found.update(pat.findall(open(f, errors="replace").read()))
    return found


def test_every_switch_is_documented():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = txt[txt.index("## 4. Environment switches"):]
    listed = set(re.findall(r"`(PWG_[A-Z0-9_]+)`", section))
    used = _read_vars()
    assert len(used) >= 40
    assert used - listed == set(), f"read by the code but not in INTEGRATION.md section 4: {sorted(used - listed)}"
    assert listed - used == set(), f"listed in INTEGRATION.md section 4 but not read anywhere: {sorted(listed - used)}"
