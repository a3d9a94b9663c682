"""CPU tests of the C-ABI boundary: the library builds for gfx950 without a GPU, loads, and
exports every symbol the public headers declare; without a device the product path fails
loudly instead of falling back."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("jpeg2png_amd.h", "jpeg2png_amd_compute.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?(?:int|void|char)\s*\*?\s*(j2p_\w+|compute)\s*\(", text, flags=re.M):
            names.add(m.group(1))
    return names


def test_headers_and_binding_list_agree():
    import jpeg2png_amd
    assert declared_symbols() == set(jpeg2png_amd.C_ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"


def test_exported_symbols_have_c_linkage(lib):
    import jpeg2png_amd
    out = subprocess.run(["nm", "-D", "--defined-only", jpeg2png_amd.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert declared_symbols() <= exported


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.j2p_version()
    assert isinstance(lib.j2p_last_error(), bytes)


def has_gpu():
    import jpeg2png_amd
    try:
        return jpeg2png_amd.device_count() > 0
    except Exception:
        return False


def test_no_gpu_means_loud_failure(lib):
    """the product path has no CPU fallback"""
    if has_gpu():
        pytest.skip("a GPU is present")
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    p = synth.make_planes(16, 16, "444", 50, seed=1, y_only=True)[0]
    p.fdata = np.zeros((16, 16), np.float32)
    with pytest.raises(j.J2PError, match="no HIP device|no CPU fallback"):
        j.Solver([p], 0.3, [0.001], 4)
    with pytest.raises(j.J2PError):
        j.decode_plane(p)
    with pytest.raises(j.J2PError):
        j.dct8x8_blocks(np.zeros((1, 64), np.float32))


def test_argument_validation_happens_before_device_use(lib):
    import jpeg2png_amd as j
    from jpeg2png_amd import synth
    p = synth.make_planes(16, 16, "444", 50, seed=1, y_only=True)[0]
    p.fdata = np.zeros((16, 16), np.float32)
    bad = synth.Plane(12, 16, 1, 1, p.data, p.quant_table, p.fdata)       # not a multiple of 8 (box.c:6-7)
    with pytest.raises(j.J2PError, match="multiple of 8"):
        j.Solver([bad], 0.3, [0.001], 4)
    zq = synth.Plane(16, 16, 1, 1, p.data, np.zeros(64, np.uint16), p.fdata)   # jpeg.c:41-45
    with pytest.raises(j.J2PError, match="quantization table"):
        j.Solver([zq], 0.3, [0.001], 4)
    with pytest.raises(j.J2PError, match="nchannel"):
        j.Solver([p, p, p, p], 0.3, [0.001] * 4, 4)


def test_missing_library_raises(monkeypatch, tmp_path):
    import jpeg2png_amd as j
    monkeypatch.setattr(j, "_lib", None)
    monkeypatch.setattr(j, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(j.J2PError, match="missing"):
        j.load_library()


def test_drop_in_compute_struct_layout(lib):
    """struct coef of include/jpeg2png_amd_compute.h must match the reference's layout
    (jpeg2png.h:7-20): 4 unsigned, two pointers, 64 uint16 = 160 bytes on LP64."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "jpeg2png_amd_compute.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu\n", sizeof(struct coef), offsetof(struct coef, w_samp),
               offsetof(struct coef, data), offsetof(struct coef, fdata), offsetof(struct coef, quant_table),
               sizeof(struct logger));
        return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [160, 12, 16, 24, 32, 24]


def test_struct_layouts_against_the_reference_headers(lib, tmp_path):
    """where /root/reference is present (this container): the same offsets printed once from the reference's own
    headers (jpeg2png.h, logger.h, progressbar.h) and once from include/jpeg2png_amd_compute.h must agree — and a
    translation unit that includes the reference's headers FIRST must still compile with ours on top (the include
    guards then skip the re-definitions), with identical prototypes for compute()."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "jpeg2png.h")):
        pytest.skip("/root/reference not present")
    body = r'''
    #include <stdio.h>
    #include <stddef.h>
    HEADERS
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu|", sizeof(struct coef), offsetof(struct coef, h), offsetof(struct coef, w),
               offsetof(struct coef, h_samp), offsetof(struct coef, w_samp), offsetof(struct coef, data),
               offsetof(struct coef, fdata), offsetof(struct coef, quant_table));
        printf("%zu %zu %zu %zu %zu|", sizeof(struct logger), offsetof(struct logger, f), offsetof(struct logger, filename),
               offsetof(struct logger, channel), offsetof(struct logger, iteration));
        printf("%zu %zu %zu\n", sizeof(struct progressbar), offsetof(struct progressbar, current), offsetof(struct progressbar, max));
        return 0;
    }'''
    outs = []
    variants = {
        "ref": '#include "jpeg2png.h"\n#include "logger.h"\n#include "progressbar.h"\n#include "compute.h"',
        "ours": '#include "jpeg2png_amd_compute.h"',
        # both: the reference's first, ours on top — also checks that the two compute() prototypes are compatible
        "both": '#include "jpeg2png.h"\n#include "logger.h"\n#include "progressbar.h"\n#include "compute.h"\n#include "jpeg2png_amd_compute.h"',
    }
    for name, headers in variants.items():
        c = tmp_path / f"{name}.c"
        c.write_text(body.replace("HEADERS", headers))
        exe = tmp_path / name
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-iquote", ref, "-I", ref, str(c), "-o", str(exe)],
                       check=True)
        outs.append(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1] == outs[2], outs


def test_gradient_launch_map_covers_every_row_of_every_strip_exactly_once():
    """k_gradient's workgroups decode their work from their number (j2p_kernels.hip.h: grad_item): units of four strips x a
    pair of tile rows, every XCD a contiguous run, dealt as double / whole / half / quarter tile-row items.  Whatever the
    shares, the canvas and the walking direction, every row of every strip must be marched exactly once, a half or quarter
    item must stay inside its tile row, a double item must be two whole tile rows of one strip — evaluated on the host
    through j2p_debug_grad_items (no GPU needed)."""
    import ctypes
    import numpy as np
    import jpeg2png_amd as j
    lib = j.load_library()
    lib.j2p_debug_grad_items.argtypes = [ctypes.c_uint] * 7 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint,
                                         ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
    rng = np.random.default_rng(606)
    cases = [(4096, 4096, 16, 1, 0, 32, 10, 0), (4096, 4096, 16, 1, 200, 24, 8, 1), (16384, 2048, 16, 1, 200, 24, 8, 0),
             (1920, 1080, 8, 1, 0, 32, 0, 0), (8, 8, 4, 1, 0, 0, 0, 0), (520, 136, 16, 3, 0, 0, 0, 0), (264, 200, 8, 2, 0, 0, 0, 1)]
    for _ in range(60):
        rpw = int(rng.choice([4, 8, 16]))
        zd, zb, zc = (int(v) for v in rng.integers(0, 257, 3))
        if rpw < 16:
            zc = 0
        if rpw < 8:
            zd = zb = 0
        if zb + zc > 256:
            zc = 256 - zb
        if zd + zb + zc > 256:
            zd = 256 - zb - zc
        cases.append((int(rng.integers(1, 300)) * 8, int(rng.integers(1, 80)) * 8, rpw, 1, zd, zb, zc, int(rng.integers(0, 2))))
    buf = np.zeros((1 << 18, 5), np.uint32)
    for W, rows, rpw, cw, zd, zb, zc, rev in cases:
        n, wgs = ctypes.c_uint(), ctypes.c_uint()
        rc = lib.j2p_debug_grad_items(W, rows, rpw, cw, zd, zb, zc, rev, buf.ctypes.data, buf.shape[0], ctypes.byref(n), ctypes.byref(wgs))
        assert rc == 0 and n.value <= buf.shape[0], (W, rows, rpw)
        it = buf[:n.value].astype(np.int64)
        ntx = 1 if W <= 4 else (W - 4 + 123) // 124
        cover = np.zeros((ntx, rows), np.int32)
        for strip, t0, nr, tr, kind in it:
            assert 0 <= strip < ntx and nr > 0 and t0 + nr <= rows
            cover[strip, t0:t0 + nr] += 1
            if kind == 3:
                assert t0 == tr * rpw and (nr == 2 * rpw or t0 + nr == rows)
            else:
                assert tr * rpw <= t0 and t0 + nr <= min((tr + 1) * rpw, rows), "an item leaves its tile row"
                assert nr <= rpw >> (0 if kind == 0 else kind)
        assert (cover == 1).all(), f"{W}x{rows} rpw {rpw} zones {zd}/{zb}/{zc} reverse {rev}: rows marched {np.unique(cover)} times"
        # one workgroup = four wavefronts (one channel) or one strip (joint): the grid holds the items with little to spare
        per_wg = 4 if cw == 1 else 1
        assert n.value <= wgs.value * per_wg


@pytest.mark.parametrize("flags", [[], ["-DJ2P_DEBUG"]], ids=["release", "checked"])
def test_no_kernel_spills_to_scratch(flags):
    """No kernel of either build may need scratch memory.  Besides the speed: kernels that use scratch on several streams
    which wait for each other's events — the bands of a row-tiled run that share a GPU — hung the queues every other run
    (round 6: the checked build's k_gradient had grown to 128 VGPRs + 132 bytes of scratch at four wavefronts per SIMD;
    tests/test_debug_build_gpu.py timed out on the GPU box).  Read from the device assembly's metadata; no GPU needed."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), *flags], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    rows = [ln for ln in r.stdout.splitlines() if " scratch " in ln]
    assert len(rows) > 40, "kernel_resources.py found too few kernels"
    bad = [ln for ln in rows if not re.search(r"scratch 0$", ln)]
    assert not bad, "kernels with scratch:\n" + "\n".join(bad)
