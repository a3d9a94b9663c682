"""The command-line driver cli/jpeg2png_gpu.c: option handling on the CPU, and on the GPU an
end-to-end comparison with the UNMODIFIED reference program (oracle/_ref/jpeg2png_ref, built by
`make -C oracle ref-cli` from /root/reference): same JPEG in, byte-identical PNG out."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "jpeg2png_ref")


@pytest.fixture(scope="module")
def cli():
    sys.path.insert(0, ROOT)
    from jpeg2png_amd.buildlib import build_cli
    exe = build_cli()
    if exe is None:
        pytest.skip("libjpeg / libpng headers not available")
    return exe


def make_jpeg(path, w, h, quality, subsampling, seed):
    from PIL import Image
    from jpeg2png_amd import synth
    rgb = synth.synth_rgb(w, h, seed).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(path, "JPEG", quality=quality, subsampling=subsampling)


def run(exe, *args):
    return subprocess.run([exe, *args], capture_output=True, text=True)


def test_version_exits_with_failure_like_the_reference(cli):
    r = run(cli, "-V")
    assert r.returncode == 1 and "version" in r.stdout            # jpeg2png.c:195-198


def test_usage_without_arguments(cli):
    r = run(cli)
    assert r.returncode == 1 and r.stdout.startswith("usage:")


@pytest.mark.parametrize("args,msg", [
    (["x.jpg", "-w", "abc"], "invalid weight"),
    (["x.jpg", "-w", "1,2,3"], "different weights are only possible when using separated components"),
    (["x.jpg", "-i", "1,2,3"], "different iteration counts are only possible when using separated components"),
    (["x.jpg", "-i", "x"], "invalid number of iterations"),
    (["x.jpg", "-p", "x"], "invalid probability weight"),
    (["x.jpg", "-t", "0"], "invalid number of threads"),
    (["a.jpg", "b.jpg", "-o", "a.png"], "must give output file names for all input files or none"),
    (["/nonexistent/x.jpg"], "could not open input file `/nonexistent/x.jpg`"),
])
def test_option_errors_match_the_reference_messages(cli, args, msg):
    r = run(cli, *args)
    assert r.returncode == 1
    assert r.stderr.strip() == "jpeg2png: " + msg


def test_refuses_to_overwrite_default_output(cli, tmp_path):
    jpg = tmp_path / "pic.jpg"
    make_jpeg(jpg, 32, 24, 50, 0, 1)
    (tmp_path / "pic.png").write_bytes(b"x")
    r = run(cli, str(jpg))
    assert r.returncode == 1 and "not overwriting output file" in r.stderr      # jpeg2png.c:303-307


def test_rejects_greyscale_jpeg(cli, tmp_path):
    from PIL import Image
    jpg = tmp_path / "grey.jpg"
    Image.fromarray(np.zeros((16, 16), np.uint8), "L").save(jpg, "JPEG")
    r = run(cli, str(jpg), "-q", "-o", str(tmp_path / "o.png"))
    assert r.returncode == 1 and r.stderr.strip() == "jpeg2png: only 3 component jpegs are supported"   # jpeg.c:34


CASES = [
    ("420_joint", 200, 136, 10, 2, ["-i", "12"]),
    ("444_joint_16bit", 96, 64, 25, 0, ["-i", "8", "-1"]),
    ("420_separate", 160, 120, 10, 2, ["-s", "-i", "10,6,4", "-w", "0.3,0.1,0"]),
    ("422_tv_only", 120, 72, 50, 1, ["-i", "6", "-w", "0", "-p", "0.002"]),
    ("odd_size_420", 101, 67, 10, 2, ["-i", "7"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_png_identical_to_reference_program(cli, tmp_path, case):
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    name, w, h, q, sub, flags = case
    jpg = tmp_path / "in.jpg"
    make_jpeg(jpg, w, h, q, sub, seed=len(name))
    ref_png, gpu_png = tmp_path / "ref.png", tmp_path / "gpu.png"
    ref_csv, gpu_csv = tmp_path / "ref.csv", tmp_path / "gpu.csv"
    r = run(REF_CLI, str(jpg), "-o", str(ref_png), "-q", "-c", str(ref_csv), "-t", "1", *flags)
    assert r.returncode == 0, r.stderr
    g = run(cli, str(jpg), "-o", str(gpu_png), "-q", "-c", str(gpu_csv), *flags)
    assert g.returncode == 0, g.stderr
    assert ref_png.read_bytes() == gpu_png.read_bytes()
    ref_rows = np.loadtxt(ref_csv, delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
    gpu_rows = np.loadtxt(gpu_csv, delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
    ref_rows = ref_rows[np.lexsort((ref_rows[:, 1], ref_rows[:, 0]))]
    gpu_rows = gpu_rows[np.lexsort((gpu_rows[:, 1], gpu_rows[:, 0]))]
    assert ref_rows.shape == gpu_rows.shape
    np.testing.assert_allclose(gpu_rows, ref_rows, rtol=0, atol=2e-6 * max(1.0, np.abs(ref_rows).max()))


@pytest.mark.gpu
def test_multiple_files_and_default_names(cli, tmp_path):
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    names = []
    for i in range(3):
        p = tmp_path / f"img{i}.jpeg"
        make_jpeg(p, 64 + 16 * i, 48, 10, 2, seed=10 + i)
        names.append(str(p))
    g = run(cli, *names, "-i", "5", "-t", "2")
    assert g.returncode == 0, g.stderr
    for i in range(3):
        out = tmp_path / f"img{i}.png"
        assert out.exists()
        ref = tmp_path / f"ref{i}.png"
        r = run(REF_CLI, names[i], "-o", str(ref), "-q", "-i", "5")
        assert r.returncode == 0
        assert out.read_bytes() == ref.read_bytes()


@pytest.mark.gpu
def test_randomised_cli_sweep(cli):
    """tools/sweep_cli.py as a test: 16 random JPEGs (sizes from 1x1, qualities 3..100, 4:4:4/4:2:2/4:2:0,
    progressive / optimised entropy coding, flat areas) with random -s/-i/-w/-p/-1 flags; PNGs byte-identical
    to the reference program's (120 of 120 in the long run of seed 1)."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_cli.py"), "16", "4"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_default_and_long_iteration_counts(cli):
    """tools/cli_long.py: no -i (the default 50), more iterations than one 32-iteration host chunk, per-component
    counts in -s mode, with and without the CSV log — PNG bytes and CSV values against the reference program"""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_long.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


REF_DROPIN = os.path.join(ROOT, "oracle", "_ref", "jpeg2png_ref_dropin")


def _csv_rows(path):
    rows = np.loadtxt(path, delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
    names = [line.split(",")[0] for line in open(path).read().splitlines()[1:]]
    order = sorted(range(len(names)), key=lambda i: (names[i], rows[i, 0], rows[i, 1]))
    return [names[i] for i in order], rows[order]


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["-i", "21"], ["-s", "-i", "40,9,5", "-w", "0.3,0.1,0"], ["-i", "6", "-w", "0", "-1"]],
                         ids=["joint", "separate", "tv_only_16bit"])
@pytest.mark.parametrize("threads", ["1", "4"])
def test_the_reference_program_itself_runs_on_the_library(tmp_path, flags, threads):
    """THE DROP-IN PROOF.  oracle/_ref/jpeg2png_ref_dropin is the reference's own main(), option parser, JPEG
    reader, PNG writer, logger.c and progressbar.c — every reference source except compute.c — linked against
    libjpeg2png_amd.so (the change INTEGRATION.md §2 shows for the reference Makefile:32-33).  Same JPEGs
    through it and through the unmodified reference program: PNG bytes identical, CSV rows identical to the 6
    decimals logger.c:23 prints.  With -t 4 the reference's nested OpenMP regions (jpeg2png.c:147 components,
    :330 files) call the library's compute() from several threads at once, sharing its progress bar."""
    if not (os.path.exists(REF_CLI) and os.path.exists(REF_DROPIN)):
        pytest.skip("oracle/_ref/jpeg2png_ref(_dropin) not built (needs /root/reference)")
    sys.path.insert(0, ROOT)
    names = []
    for i, (w, h, q, sub) in enumerate([(200, 136, 10, 2), (96, 64, 30, 0), (133, 77, 12, 1), (64, 48, 50, 2)]):
        p = tmp_path / f"img{i}.jpg"
        make_jpeg(p, w, h, q, sub, seed=40 + i)
        names.append(str(p))
    outs = {}
    for tag, exe in (("ref", REF_CLI), ("dropin", REF_DROPIN)):
        d = tmp_path / tag
        d.mkdir()
        args = []
        for i, n in enumerate(names):
            args += ["-o", str(d / f"o{i}.png")]
        r = run(exe, *names, *args, "-c", str(d / "log.csv"), "-t", threads, *flags)
        assert r.returncode == 0, r.stderr
        assert "100%" in r.stdout                      # the reference's own progress bar ran to the end
        outs[tag] = d
    for i in range(len(names)):
        assert (outs["ref"] / f"o{i}.png").read_bytes() == (outs["dropin"] / f"o{i}.png").read_bytes(), f"file {i}"
    rn, rr = _csv_rows(outs["ref"] / "log.csv")
    dn, dr = _csv_rows(outs["dropin"] / "log.csv")
    assert rn == dn and rr.shape == dr.shape
    np.testing.assert_allclose(dr, rr, rtol=0, atol=2e-6 * max(1.0, np.abs(rr).max()))


@pytest.mark.gpu
def test_a_failure_inside_the_library_reads_like_the_reference_dying(tmp_path):
    """compute() has no return value: any failure ends like die() (utils.c:11-28) — the host's own
    die_message_start() wipes the progress bar off the line, then `jpeg2png: <message>` and EXIT_FAILURE.  Provoked
    here by asking the reference program, linked against the library, for a GPU that does not exist."""
    if not os.path.exists(REF_DROPIN):
        pytest.skip("oracle/_ref/jpeg2png_ref_dropin not built (needs /root/reference)")
    jpg = tmp_path / "a.jpg"
    make_jpeg(jpg, 64, 48, 30, 2, seed=9)
    env = dict(os.environ, J2P_DEVICE="99")
    env.pop("J2P_DEVICES", None)
    r = subprocess.run([REF_DROPIN, str(jpg), "-o", str(tmp_path / "a.png"), "-i", "3"], capture_output=True, text=True, env=env)
    assert r.returncode == 1
    assert r.stderr.startswith("jpeg2png: ") and "device" in r.stderr
    assert not (tmp_path / "a.png").exists()
    # the bar was drawn and then wiped by the host's own die_message_start (progressbar_clear, progressbar.c:57-66:
    # carriage return, 77 blanks, carriage return — which text mode hands over as newlines)
    assert "0%" in r.stdout and r.stdout.rstrip("\n").endswith(" " * 77)


@pytest.mark.gpu
def test_truncated_jpeg_warns_and_carries_on_like_the_reference(cli, tmp_path):
    """libjpeg reports a premature end of file through output_message; the reference prints it
    (`jpeg2png: libjpeg error: ...`, jpeg.c:14-19) and decodes what is there.  So must the CLI — and the drop-in."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    jpg = tmp_path / "full.jpg"
    make_jpeg(jpg, 160, 120, 30, 2, seed=5)
    data = jpg.read_bytes()
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(data[: len(data) * 2 // 3])
    r = run(REF_CLI, str(cut), "-o", str(tmp_path / "ref.png"), "-q", "-i", "5")
    g = run(cli, str(cut), "-o", str(tmp_path / "gpu.png"), "-q", "-i", "5")
    assert r.returncode == 0 and g.returncode == 0, (r.stderr, g.stderr)
    assert "libjpeg error" in r.stderr
    assert g.stderr == r.stderr
    assert (tmp_path / "ref.png").read_bytes() == (tmp_path / "gpu.png").read_bytes()
    if os.path.exists(REF_DROPIN):
        d = run(REF_DROPIN, str(cut), "-o", str(tmp_path / "dropin.png"), "-q", "-i", "5")
        assert d.returncode == 0 and d.stderr == r.stderr
        assert (tmp_path / "ref.png").read_bytes() == (tmp_path / "dropin.png").read_bytes()


def test_truncated_jpeg_message_without_a_gpu(cli, tmp_path):
    """CPU half of the above: the warning is printed in the reference's format and is not fatal by itself (what
    follows — creating the solver — is, on a box without a GPU)"""
    jpg = tmp_path / "full.jpg"
    make_jpeg(jpg, 160, 120, 30, 2, seed=5)
    data = jpg.read_bytes()
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(data[: len(data) * 2 // 3])
    g = run(cli, str(cut), "-o", str(tmp_path / "gpu.png"), "-q", "-i", "2")
    lines = g.stderr.strip().splitlines()
    assert lines and lines[0].startswith("jpeg2png: libjpeg error: ")
    if g.returncode != 0:      # no GPU here: the next line must be the loud "no device" failure, not the warning
        assert len(lines) >= 2 and "HIP device" in lines[-1]


TILE_CASES = [
    ("tall_420_joint", 136, 400, 10, 2, ["-i", "9"]),
    ("tall_420_separate_16bit", 120, 328, 20, 2, ["-s", "-i", "8,5,3", "-w", "0.3,0.1,0", "-1"]),
    ("tall_444_joint_odd", 77, 301, 10, 0, ["-i", "6"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", TILE_CASES, ids=[c[0] for c in TILE_CASES])
def test_one_tall_image_is_row_tiled_over_every_listed_gpu(cli, tmp_path, case):
    """fewer files than GPUs in J2P_DEVICES: the image is solved as one band per listed device (the four "devices"
    are this box's GPU four times) — decode_file -> compute of one image, jpeg2png.c:141-152 — with the bands
    converting their own rows to RGB: PNG bytes and CSV rows must equal the reference program's"""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    name, w, h, q, sub, flags = case
    jpg = tmp_path / "in.jpg"
    make_jpeg(jpg, w, h, q, sub, seed=40 + len(name))
    ref_png, gpu_png = tmp_path / "ref.png", tmp_path / "gpu.png"
    ref_csv, gpu_csv = tmp_path / "ref.csv", tmp_path / "gpu.csv"
    r = run(REF_CLI, str(jpg), "-o", str(ref_png), "-q", "-c", str(ref_csv), "-t", "1", *flags)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, J2P_DEVICES="0,0,0,0", J2P_TILE_MIN_BAND_PIXELS="0")     # (small test images: lower the pixel gate)
    g = subprocess.run([cli, str(jpg), "-o", str(gpu_png), "-q", "-c", str(gpu_csv), *flags], capture_output=True, text=True, env=env)
    assert g.returncode == 0, g.stderr
    assert ref_png.read_bytes() == gpu_png.read_bytes()
    ref_rows = np.loadtxt(ref_csv, delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
    gpu_rows = np.loadtxt(gpu_csv, delimiter=",", skiprows=1, usecols=(1, 2, 3, 4, 5, 6), ndmin=2)
    ref_rows = ref_rows[np.lexsort((ref_rows[:, 1], ref_rows[:, 0]))]
    gpu_rows = gpu_rows[np.lexsort((gpu_rows[:, 1], gpu_rows[:, 0]))]
    assert ref_rows.shape == gpu_rows.shape
    np.testing.assert_allclose(gpu_rows, ref_rows, rtol=0, atol=2e-6 * max(1.0, np.abs(ref_rows).max()))


@pytest.mark.gpu
def test_two_files_share_out_four_listed_gpus(cli, tmp_path):
    """more than one file, still fewer than GPUs: file i is tiled over ITS share of J2P_DEVICES (two "GPUs" each here)
    instead of every file over all of them; PNG bytes equal the reference program's"""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/jpeg2png_ref not built (needs /root/reference)")
    names = []
    for i in range(2):
        p = tmp_path / f"tall{i}.jpg"
        make_jpeg(p, 72 + 8 * i, 200 + 56 * i, 12, 2, seed=70 + i)
        names.append(str(p))
    env = dict(os.environ, J2P_DEVICES="0,0,0,0", J2P_TILE_MIN_BAND_PIXELS="0")
    g = subprocess.run([cli, *names, "-q", "-i", "7"], capture_output=True, text=True, env=env)
    assert g.returncode == 0, g.stderr
    for i in range(2):
        ref = tmp_path / f"ref{i}.png"
        r = run(REF_CLI, names[i], "-o", str(ref), "-q", "-i", "7", "-t", "1")
        assert r.returncode == 0, r.stderr
        assert (tmp_path / f"tall{i}.png").read_bytes() == ref.read_bytes()


def test_threads_default_is_the_core_count_like_openmp(cli):
    """-t is optional and its default is the online core count (the reference leaves it to OpenMP,
    jpeg2png.c:246-257): the usage text says so"""
    r = run(cli, "-h")
    assert "default: online cores" in r.stdout
