"""The C command-line tools (som_lvq_pak_amd/host): same flags, files and output text as the
reference's vsom / lvq1..olvq1 / qerror / accuracy / vcal.  GPU tests replay the CLI chains
whose outputs the REAL reference produced (tests/golden/cli) and compare bytes."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "som_lvq_pak_amd", "host", "bin")
DATA = os.path.join(GOLDEN, "data")
CLI = os.path.join(GOLDEN, "cli")
EXPECTED = json.load(open(os.path.join(CLI, "expected.json")))


@pytest.fixture(scope="module")
def tools():
    if not all(os.path.exists(os.path.join(BIN, t)) for t in ("vsom", "knntest", "classify", "eveninit", "propinit", "balance", "cmatr", "setlabel", "elimin", "vfind")):
        subprocess.check_call(["make", "-s", "-C", ROOT, "lib"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "som_lvq_pak_amd", "host")])
    return BIN


def run(tool, *args, cwd=None, check=True):
    p = subprocess.run([os.path.join(BIN, tool)] + [str(a) for a in args], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=cwd)
    if check:
        assert p.returncode == 0, (tool, args, p.stderr)
    return p


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


# ------------------------------------------------------------------ CPU side
def test_tools_build_and_usage(tools):
    for t in ("vsom", "lvqtrain", "qerror", "accuracy", "vcal", "lvq1", "olvq1", "lvq2", "lvq3",
              "eveninit", "propinit", "knntest", "classify", "balance", "cmatr", "setlabel", "elimin"):
        p = run(t, "-help")
        assert "MI355X" in p.stdout
    p = run("qerror", "-din", "x", check=False)          # required flag missing: message + exit(-1)
    assert p.returncode == 255 and "Can't find asked option -cin" in p.stderr
    p = run("lvqtrain", "-type", "nosuch", "-din", "a", "-cin", "b", "-cout", "c", "-rlen", 1, check=False)
    assert p.returncode == 1 and "Unknown LVQ type nosuch" in p.stderr


def test_fast_number_parser_equals_sscanf(tools, tmp_path):
    """the .dat/.cod reader's fast path for plain decimals must give the float sscanf("%f") gives
    (reference datafile.c:627, 664) on every token: generated decimals, shortest and 9-digit forms of
    random floats, values placed on and next to float ties, exponents, junk"""
    exe = str(tmp_path / "parse_check")
    host = os.path.join(ROOT, "som_lvq_pak_amd", "host")
    subprocess.check_call(["gcc", "-O2", "-I", host, "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "parse_check.c"), os.path.join(host, "paklib.c"),
                           "-L", os.path.join(ROOT, "som_lvq_pak_amd"), "-lsomhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "som_lvq_pak_amd"), "-lm"])
    p = subprocess.run([exe, "400000"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("mismatches 0"), p.stdout


def test_tools_refuse_without_gpu(tools, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = run("vsom", "-din", os.path.join(DATA, "ex.dat"), "-cin", os.path.join(CLI, "som_init_hexa_bubble.cod"),
            "-cout", tmp_path / "o.cod", "-rlen", 10, "-alpha", 0.05, "-radius", 3, "-v", 0, check=False)
    assert p.returncode == 1 and "no CPU path" in p.stderr
    assert not os.path.exists(tmp_path / "o.cod")
    p = run("vsom", "-din", os.path.join(DATA, "ex1.dat"), "-cin", os.path.join(CLI, "som_init_hexa_bubble.cod"),
            "-cout", tmp_path / "o.cod", "-rlen", 10, "-alpha", 0.05, "-radius", 3, check=False)
    assert p.returncode == 1 and "different dimensions" in p.stderr


# ------------------------------------------------------------------ GPU side
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["hexa_bubble", "hexa_gaussian", "rect_bubble", "rect_gaussian"])
def test_vsom_and_qerror_match_reference_cli(tools, tmp_path, tag):
    ex = EXPECTED["som"][tag]
    out = tmp_path / "out.cod"
    run("vsom", "-din", os.path.join(DATA, "ex.dat"), "-cin", os.path.join(CLI, ex["init"]), "-cout", out,
        "-rlen", ex["rlen"], "-alpha", ex["alpha"], "-radius", ex["radius"], "-v", 0)
    assert md5(out) == ex["md5"]
    p = run("qerror", "-din", os.path.join(DATA, "ex.dat"), "-cin", out, "-v", 0)
    assert p.stdout == ex["qerror_stdout"]


@pytest.mark.gpu
def test_vfind_matches_reference_cli(tools, tmp_path):
    """vfind.c:244-306: three trials (seeds 3, 2, 1) of randinit -> two som_training runs -> qerror
    (or -qetype 1); per-trial errors, the winning seed and the saved map must equal the reference's"""
    answers = ["3", "{data}", "{data}", "{out}", "hexa", "bubble", "6", "5", "800", "0.05", "5", "2000", "0.02", "2"]
    for tag, ex in EXPECTED["som"]["vfind"].items():
        out = tmp_path / (tag + ".cod")
        ans = "\n".join(a.format(data=os.path.join(DATA, "ex.dat"), out=out) for a in answers) + "\n"
        p = subprocess.run([os.path.join(BIN, "vfind")] + ex["args"], input=ans, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        assert p.returncode == 0, p.stderr
        trials = [ln for ln in p.stderr.splitlines() if ": " in ln and ln.strip()[:1].isdigit()]
        assert trials == ex["trials_stderr"], tag
        assert p.stdout.strip().splitlines()[-1] == ex["last_stdout_line"], tag
        assert md5(out) == ex["md5"], tag


@pytest.mark.gpu
def test_qerror_qetype1_matches_reference_cli(tools):
    """qerror -qetype 1 -radius 2 on the reference's own map (find_qerror2, som_rout.c:823)"""
    p = run("qerror", "-din", os.path.join(DATA, "ex.dat"), "-cin", os.path.join(CLI, "som_hexa_bubble.cod"),
            "-qetype", 1, "-radius", 2, "-v", 0)
    assert p.stdout == EXPECTED["som"]["qerror2_r2"]


@pytest.mark.gpu
def test_vsom_variants(tools, tmp_path):
    d, init = os.path.join(DATA, "ex.dat"), os.path.join(CLI, "som_init_hexa_bubble.cod")
    out = tmp_path / "o.cod"
    run("vsom", "-din", d, "-cin", init, "-cout", out, "-rlen", 5000, "-alpha", 0.05, "-radius", 10,
        "-alpha_type", "inverse_t", "-selfuncs", "hip", "-v", 0)
    assert md5(out) == EXPECTED["som"]["inverse_t"]["md5"]
    run("vsom", "-din", d, "-cin", init, "-cout", out, "-rlen", 5000, "-alpha", 0.05, "-radius", 10,
        "-rand", 7, "-v", 0)
    assert md5(out) == EXPECTED["som"]["rand7"]["md5"]
    p = run("vsom", "-din", d, "-cin", init, "-cout", out, "-rlen", 100, "-alpha", 0.05, "-radius", 10,
            "-selfuncs", "nosuch", "-v", 0)
    assert "functions for 'nosuch' not found, using defaults" in p.stderr


@pytest.mark.gpu
def test_somexample_chain(tools, tmp_path):
    """reference Makefile:195-205 -> qerror 3.571006, then vcal"""
    d = os.path.join(DATA, "ex.dat")
    cod = tmp_path / "ex.cod"
    shutil.copyfile(os.path.join(CLI, "som_init_hexa_bubble.cod"), cod)
    run("vsom", "-din", d, "-cin", cod, "-cout", cod, "-rlen", 1000, "-alpha", 0.05, "-radius", 10, "-v", 0)
    run("vsom", "-din", d, "-cin", cod, "-cout", cod, "-rlen", 10000, "-alpha", 0.02, "-radius", 3, "-v", 0)
    assert md5(cod) == EXPECTED["som"]["somexample"]["md5"]
    p = run("qerror", "-din", d, "-cin", cod, "-v", 0)
    assert p.stdout == "3.571006\n"
    p = run("qerror", "-din", d, "-cin", cod)
    assert p.stdout.endswith("is 3.571006 per sample (3840 samples)\n")
    lab = tmp_path / "lab.cod"
    run("vcal", "-din", os.path.join(DATA, "ex_fts.dat"), "-cin", cod, "-cout", lab, "-v", 0)
    assert md5(lab) == EXPECTED["som"]["somexample_vcal_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["lvq1_10000", "lvq1", "lvq2", "lvq3", "olvq1", "olvq1_default"])
def test_lvq_tools_match_reference_cli(tools, tmp_path, tag):
    ex = EXPECTED["lvq"][tag]
    out = tmp_path / "out.cod"
    run(ex["tool"], "-din", os.path.join(DATA, "ex1.dat"), "-cin", os.path.join(CLI, "lvq_init.cod"),
        "-cout", out, *ex["args"], "-v", 0)
    assert md5(out) == ex["md5"]
    assert not os.path.exists(tmp_path / "out.lra")            # lvqtrain.c:249 removes it
    p = run("accuracy", "-din", os.path.join(DATA, "ex2.dat"), "-cin", out, "-v", 0)
    assert p.stdout == ex["accuracy_stdout"]
    # the same through `lvqtrain -type`
    out2 = tmp_path / "out2.cod"
    run("lvqtrain", "-type", ex["tool"], "-din", os.path.join(DATA, "ex1.dat"),
        "-cin", os.path.join(CLI, "lvq_init.cod"), "-cout", out2, *ex["args"], "-v", 0)
    assert md5(out2) == ex["md5"]


@pytest.mark.gpu
def test_lvq_init_knntest_classify_match_reference_cli(tools, tmp_path):
    """the k-NN consumers around the LVQ loops, byte for byte: eveninit / propinit (k-NN vote of
    every entry over the whole data set, second picking pass included), knntest, classify"""
    t = EXPECTED["lvq"]["tools"]
    out = tmp_path / "init.cod"
    run("eveninit", "-din", os.path.join(DATA, "ex1.dat"), "-cout", out, "-noc", 200, "-v", 0)
    assert md5(out) == EXPECTED["lvq"]["init_md5"]
    for tag in ("propinit_200", "eveninit_knn3_100", "propinit_knn1_60", "eveninit_800"):
        run(t[tag]["tool"], "-din", os.path.join(DATA, "ex1.dat"), "-cout", out, *t[tag]["args"], "-v", 0)
        assert md5(out) == t[tag]["md5"], tag
    run("initlvq", "-type", "propinit", "-din", os.path.join(DATA, "ex1.dat"), "-cout", out, "-noc", 200, "-v", 0)
    assert md5(out) == t["propinit_200"]["md5"]
    cod = os.path.join(CLI, "lvq_olvq1.cod")
    for knn in (1, 3, 5):
        p = run("knntest", "-din", os.path.join(DATA, "ex2.dat"), "-cin", cod, "-knn", knn, "-v", 0)
        assert p.stdout == t["knntest_%d" % knn]
    run("classify", "-din", os.path.join(DATA, "ex2.dat"), "-cin", cod, "-dout", tmp_path / "cls.dat",
        "-cfout", tmp_path / "cls.cfo", "-v", 0)
    assert md5(tmp_path / "cls.dat") == t["classify_dout_md5"]
    assert md5(tmp_path / "cls.cfo") == t["classify_cfout_md5"]
    p = run("knntest", "-din", os.path.join(DATA, "ex2.dat"), "-cin", cod, "-knn", 9, check=False)
    assert p.returncode == 1 and "at most 8" in p.stderr


@pytest.mark.gpu
def test_cmatr_setlabel_elimin_match_reference_cli(tools, tmp_path):
    t = EXPECTED["lvq"]["tools"]
    cod = os.path.join(CLI, "lvq_olvq1.cod")
    p = run("cmatr", "-din", os.path.join(DATA, "ex2.dat"), "-cin", cod, "-cfout", tmp_path / "cm.cfo", "-v", 0)
    assert p.stdout == t["cmatr"]
    assert md5(tmp_path / "cm.cfo") == t["cmatr_cfout_md5"]
    for knn in (3, 5):
        run("setlabel", "-din", os.path.join(DATA, "ex2.dat"), "-cin", cod, "-cout", tmp_path / "sl.cod", "-knn", knn, "-v", 0)
        assert md5(tmp_path / "sl.cod") == t["setlabel_%d_md5" % knn], knn
    for knn in (3, 5, 8):
        run("elimin", "-din", os.path.join(DATA, "ex1.dat"), "-cout", tmp_path / "el.cod", "-knn", knn, "-v", 0)
        assert md5(tmp_path / "el.cod") == t["elimin_%d_md5" % knn], knn


@pytest.mark.gpu
def test_balance_matches_reference_cli(tools, tmp_path):
    """balance (balance.c:44-226): medians, removal, k-NN picking, one OLVQ1 pass -- codebook, .lra
    and the printed class table byte for byte"""
    t = EXPECTED["lvq"]["tools"]
    e400 = tmp_path / "even400.cod"
    run("eveninit", "-din", os.path.join(DATA, "ex1.dat"), "-cout", e400, "-noc", 400, "-v", 0)
    for tag, cin in (("balance_even", os.path.join(CLI, "lvq_init.cod")), ("balance_even400_knn3", e400)):
        out = tmp_path / (tag + ".cod")
        p = run("balance", "-din", os.path.join(DATA, "ex1.dat"), "-cin", cin, "-cout", out, *t[tag]["args"], "-v", 0)
        assert p.stdout == t[tag]["stdout"], tag
        assert md5(out) == t[tag]["md5"], tag
        assert md5(tmp_path / (tag + ".lra")) == t[tag]["lra_md5"], tag


@pytest.mark.gpu
def test_snapshots_and_cfout(tools, tmp_path):
    d, init = os.path.join(DATA, "ex.dat"), os.path.join(CLI, "som_init_hexa_bubble.cod")
    out = tmp_path / "o.cod"
    run("vsom", "-din", d, "-cin", init, "-cout", out, "-rlen", 5000, "-alpha", 0.05, "-radius", 10,
        "-snapinterval", 2000, "-snapfile", str(tmp_path / "snap_%ld.cod"), "-v", 0)
    assert md5(out) == EXPECTED["som"]["hexa_bubble"]["md5"]     # segments change nothing
    for it in (2000, 4000):
        txt = open(tmp_path / ("snap_%d.cod" % it)).read().splitlines()
        assert txt[0] == "5 hexa 12 8 bubble" and txt[1] == "#SNAPSHOT FILE" and txt[2] == "#iterations: %d/5000" % it
        assert len(txt) == 3 + 96
    cf = tmp_path / "cf.txt"
    run("accuracy", "-din", os.path.join(DATA, "ex2.dat"), "-cin", os.path.join(CLI, "lvq_lvq1.cod"),
        "-cfout", cf, "-v", 0)
    flags = open(cf).read().split()
    assert len(flags) == 1962 and 100.0 * flags.count("1") / 1962 == pytest.approx(87.56, abs=0.005)


@pytest.mark.gpu
def test_vsom_minibatch_flag(tools, tmp_path, oracle, exdata):
    """-batch B runs the mini-batch schedule: equals the batch oracle, bytes of the .cod included"""
    from conftest import read_cod
    from som_lvq_pak_amd import textio
    ini = read_cod("som_init_hexa_bubble.cod")
    want, _, _ = oracle.som_train(ini.points, 12, 8, 3, 1, exdata["ex"].points, 5000, 0.05, 10.0, batch=64, trace=False)
    ini.points = want
    ref = tmp_path / "want.cod"
    textio.write_entries(str(ref), ini)
    out = tmp_path / "o.cod"
    run("vsom", "-din", os.path.join(DATA, "ex.dat"), "-cin", os.path.join(CLI, "som_init_hexa_bubble.cod"),
        "-cout", out, "-rlen", 5000, "-alpha", 0.05, "-radius", 10, "-batch", 64, "-v", 0)
    assert md5(out) == md5(ref)


def test_randinit_matches_reference(tools, tmp_path):
    """host-only step before the path: same LCG, same bounding-box rule, same bytes"""
    out = tmp_path / "init.cod"
    run("randinit", "-din", os.path.join(DATA, "ex.dat"), "-cout", out, "-xdim", 12, "-ydim", 8,
        "-topol", "hexa", "-neigh", "bubble", "-rand", 123, "-v", 0)
    assert md5(out) == EXPECTED["som"]["randinit_md5"]


@pytest.mark.gpu
def test_buffer_with_rand_matches_reference_cli(tools, tmp_path):
    """-buffer N -rand: the reference reshuffles every buffer as it is (re)loaded and rewinds the
    file after the last one (datafile.c:237-344, 754-830); SOM and LVQ loops, N below and above the
    file length"""
    for tag, ex in EXPECTED["buffer_rand"].items():
        out = tmp_path / (tag + ".cod")
        run(ex["tool"], "-din", os.path.join(DATA, ex["data"]), "-cin", os.path.join(CLI, ex["cin"]), "-cout", out,
            *ex["args"], "-v", 0)
        assert md5(out) == ex["md5"], tag


@pytest.mark.gpu
def test_lininit_matches_reference(tools, tmp_path):
    """lininit_codes (som_rout.c:322): mean and centred product sums from the GPU (fp32, rows in
    order), eigenvector iteration on the host -- the reference's bytes, masked components included"""
    for tag, ex in EXPECTED["som"]["lininit"].items():
        out = tmp_path / (tag + ".cod")
        run("lininit", "-din", os.path.join(DATA, ex["data"]), "-cout", out, *ex["args"], "-v", 0)
        assert md5(out) == ex["md5"], tag
    out = tmp_path / "m.cod"
    ex = EXPECTED["som"]["lininit"]["ex_hexa"]
    run("mapinit", "-init", "lin", "-din", os.path.join(DATA, ex["data"]), "-cout", out, *ex["args"], "-v", 0)
    assert md5(out) == ex["md5"]


@pytest.mark.gpu
def test_whole_somexample_on_these_tools(tools, tmp_path):
    """reference Makefile:195-205 end to end with these binaries only:
    randinit -> vsom -> vsom -> qerror -> vcal -> visual"""
    d = os.path.join(DATA, "ex.dat")
    cod = tmp_path / "ex.cod"
    run("randinit", "-din", d, "-cout", cod, "-xdim", 12, "-ydim", 8, "-topol", "hexa", "-neigh", "bubble",
        "-rand", 123, "-v", 0)
    run("vsom", "-din", d, "-cin", cod, "-cout", cod, "-rlen", 1000, "-alpha", 0.05, "-radius", 10, "-v", 0)
    run("vsom", "-din", d, "-cin", cod, "-cout", cod, "-rlen", 10000, "-alpha", 0.02, "-radius", 3, "-v", 0)
    assert run("qerror", "-din", d, "-cin", cod, "-v", 0).stdout == "3.571006\n"
    run("vcal", "-din", os.path.join(DATA, "ex_fts.dat"), "-cin", cod, "-cout", cod, "-v", 0)
    assert md5(cod) == EXPECTED["som"]["somexample_vcal_md5"]
    vis = tmp_path / "ex.vis"
    run("visual", "-din", os.path.join(DATA, "ex_fts.dat"), "-cin", cod, "-dout", vis, "-v", 0)
    assert md5(vis) == EXPECTED["som"]["somexample_vis_md5"]
