"""The gfx950 assembly of the env kernels keeps what round 6 fixed (no GPU needed: hipcc cross-compiles): no vector load inside the
observation wave's loop of any role-specialised instantiation (a conditional lvalue on kernel-argument members once put a
`global_load` + `s_waitcnt vmcnt(0)` there, twice per env-step: DESIGN.md 4.1.3), and the paced flush sites are present."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_observation_wave_loops_hold_no_vector_load():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [l for l in r.stdout.split("\n") if l.startswith("dcc_env_roles_kernel<")]
    assert len(rows) >= 12                                   # 3 action sources x force on / off x the BASELINE sizes
    for l in rows:
        obs = l.split("|")[2].split()
        assert len(obs) == 3 and int(obs[0]) == 0 and int(obs[1]) == 0 and int(obs[2]) > 0, l      # vload, compiler drains, paced sites
    head = [l for l in rows if l.startswith("dcc_env_roles_kernel<0, false, 8, 64>")]
    assert head and int(re.split(r"\s+", head[0].split("|")[0].strip())[-2]) >= 4, head            # the headline kernel keeps 4 waves per SIMD
