"""The checked build of the kernels (-DJ2P_DEBUG, jpeg2png_amd/libjpeg2png_amd_debug.so): every global load and
store of the two phase kernels is compared on the device with the byte range it is meant to stay in — the
counterpart of the reference's DEBUG=1 build, whose pixel indexer p() asserts every access (utils.h:68-81).
The kernels' loads are unconditional by design (clamped addresses + masks), so an off-by-one would be silent
in the release build; here it is counted."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEBUG_LIB = os.path.join(ROOT, "jpeg2png_amd", "libjpeg2png_amd_debug.so")


def _sweep(library):
    env = dict(os.environ)
    env.pop("J2P_LIBRARY", None)
    if library:
        env["J2P_LIBRARY"] = library
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug_sweep.py"), "16", "4"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("checked build")]
    assert last, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.match(r"checked build: (\w+); violations: (\d+); digest (\w+)", last[-1])
    return r.returncode, m.group(1) == "True", int(m.group(2)), m.group(3), r.stdout


@pytest.mark.gpu
def test_checked_build_sees_no_stray_access_and_computes_the_same_bits(lib):
    # the checked library does not travel with the lease (.gpurunignore): built here, once, where it is used
    # (hipcc is part of the image; only the device translation unit is compiled a second time)
    from jpeg2png_amd.buildlib import build_debug
    assert build_debug() == DEBUG_LIB and os.path.exists(DEBUG_LIB)
    rc_d, checked_d, viol, digest_d, out_d = _sweep(DEBUG_LIB)
    assert checked_d, "the debug library does not report a J2P_DEBUG build"
    assert rc_d == 0 and viol == 0, out_d[-3000:]
    rc_r, checked_r, _, digest_r, out_r = _sweep(None)
    assert rc_r == 0 and not checked_r, out_r[-3000:]
    assert digest_d == digest_r, "checked and release builds disagree on the result bits"


def test_release_build_has_the_checks_compiled_out(lib):
    import jpeg2png_amd as j
    assert j.debug_build() is False
