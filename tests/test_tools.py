"""The GPU-less developer tools keep working: tools/isa_wait_scan.py compiles a unit to gfx950 ISA and prints its table."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_isa_wait_scan_runs_on_the_smallest_unit():
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'isa_wait_scan.py'), 'records.hip'],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith('| serialised loads |') and lines[1].startswith('|---')
    for row in lines[2:]:                      # whatever it lists parses as the six numeric-or-text columns of the header
        cells = [c.strip() for c in row.strip('|').split('|')]
        assert len(cells) == 7 and all(c.isdigit() for c in cells[:5])
