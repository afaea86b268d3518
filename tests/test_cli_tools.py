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


def test_registry_row_dist_and_adapt_equal_the_oracle(tools, tmp_path):
    """the "hip" row fills all three pointers of the reference's registry (datafile.c:1217-1220); its per-row
    dist / vector_adapt give the bits of vector_dist_euc / adapt_vector (oracle restatement) on masked rows"""
    exe = str(tmp_path / "row_funcs_check")
    host = os.path.join(ROOT, "som_lvq_pak_amd", "host")
    orc = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", orc, "oracle"])
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-I", host, "-I", os.path.join(ROOT, "include"), "-I", orc, "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "row_funcs_check.c"), os.path.join(host, "paklib.c"),
                           "-L", os.path.join(ROOT, "som_lvq_pak_amd"), "-lsomhip", "-L", orc, "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "som_lvq_pak_amd"), "-Wl,-rpath," + orc, "-lm"])
    p = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and p.stdout.strip().endswith("mismatches 0"), p.stdout


def test_a_failing_rank_ends_the_others(tools, tmp_path):
    """pak_run_ranks (paklib.c; `vsom/lvqtrain/vfind -gpus G`): a rank that exits non-zero or dies of a signal while
    the others block (rank 0 in a read from a live peer, the rest in pause() -- as a rank inside an RCCL collective
    would) ends them all: the call returns 1 within seconds and no child is left"""
    exe = str(tmp_path / "ranks_fail")
    host = os.path.join(ROOT, "som_lvq_pak_amd", "host")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-I", host, "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "ranks_fail.c"), os.path.join(host, "paklib.c"),
                           "-L", os.path.join(ROOT, "som_lvq_pak_amd"), "-lsomhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "som_lvq_pak_amd"), "-lm"])
    for mode, world, want in (("ok", 4, 0), ("exit", 3, 1), ("signal", 4, 1), ("exit", 2, 1)):
        p = subprocess.run([exe, mode, str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        assert p.stdout.startswith("returned %d after" % want) and p.stdout.strip().endswith("children left 0"), (mode, p.stdout, p.stderr)
        secs = float(p.stdout.split("after")[1].split("s;")[0])
        assert secs < 10.0, p.stdout


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
def test_vfind_gpus_replicas_equal_the_sequential_run(tools, tmp_path):
    """vfind -gpus G runs the trials as independent replicas, one process per GPU (two ranks sharing this box's GPU
    here); rank 0 prints the trials in the reference's order and keeps the first smallest error: same lines, same map
    as the reference's sequential loop (vfind.c:244-306)."""
    answers = ["3", "{data}", "{data}", "{out}", "hexa", "bubble", "6", "5", "800", "0.05", "5", "2000", "0.02", "2"]
    ex = EXPECTED["som"]["vfind"]["q0"]
    out = tmp_path / "g2.cod"
    ans = "\n".join(a.format(data=os.path.join(DATA, "ex.dat"), out=out) for a in answers) + "\n"
    p = subprocess.run([os.path.join(BIN, "vfind"), "-gpus", "2"] + ex["args"], input=ans, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    trials = [ln for ln in p.stderr.splitlines() if ": " in ln and ln.strip()[:1].isdigit()]
    assert trials == ex["trials_stderr"]
    assert p.stdout.strip().splitlines()[-1] == ex["last_stdout_line"]
    assert md5(out) == ex["md5"]


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
def test_buffer_with_snapshots_matches_reference_cli(tools, tmp_path):
    """-buffer N -snapinterval N (and interval 1): a segment cut by the buffer feed can start exactly on a snapshot
    iteration; the reference saves after every le % interval == 0, le > 0 (som_rout.c:650, lvq_rout.c:559).
    Snapshot files and the final codebook byte for byte."""
    for tag, ex in EXPECTED["buffer_snap"].items():
        out = tmp_path / (tag + ".cod")
        run(ex["tool"], "-din", os.path.join(DATA, ex["data"]), "-cin", os.path.join(CLI, ex["cin"]), "-cout", out,
            *ex["args"], "-snapfile", str(tmp_path / (tag + "_%ld.snap")), "-v", 0)
        assert md5(out) == ex["md5"], tag
        made = sorted(f for f in os.listdir(tmp_path) if f.startswith(tag + "_") and f.endswith(".snap"))
        assert made == sorted("%s_%s.snap" % (tag, i) for i in ex["snapshots"]), (tag, made)
        for it, want in ex["snapshots"].items():
            assert md5(tmp_path / ("%s_%s.snap" % (tag, it))) == want, (tag, it)


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


# ------------------------------------------------------------------ raw fp32 side format + generator source
def _f32_payload(path):
    raw = open(path, "rb").read()
    head_end = raw.index(b"\n", raw.index(b"\n") + 1) + 1
    first = raw[:raw.index(b"\n")].split()
    dim = int(raw[raw.index(b"\n") + 1:head_end].split()[0])
    n = int(first[1])
    import numpy as np
    x = np.frombuffer(raw[head_end:head_end + 4 * n * dim], dtype=np.float32).reshape(n, dim)
    return x, raw[head_end + 4 * n * dim:].decode()


def test_datconv_raw_fp32_roundtrip(tools, tmp_path):
    """text -> raw fp32 keeps every number the text reader produced (sscanf("%f") semantics), masks ride
    as NaN, labels / weight= / fixed= in the trailing text section; raw -> raw is the identity and raw -> text prints what text -> text prints."""
    import numpy as np
    ex = os.path.join(DATA, "ex.dat")
    run("datconv", "-din", ex, "-dout", tmp_path / "ex.f32")
    x, tail = _f32_payload(tmp_path / "ex.f32")
    want = np.array([[np.float32(t) for t in ln.split()[:5]] for ln in open(ex) if not ln.startswith("#")][1:], dtype=np.float32)
    assert tail == "" and np.array_equal(x.view(np.uint32), want.view(np.uint32))
    for name in ("ex_masked.dat", "ex_fts.dat", "ex1.dat"):
        src = os.path.join(DATA, name)
        run("datconv", "-din", src, "-dout", tmp_path / "a.txt", "-text", "-noskip")
        run("datconv", "-din", src, "-dout", tmp_path / "a.f32", "-noskip")
        run("datconv", "-din", tmp_path / "a.f32", "-dout", tmp_path / "b.txt", "-text", "-noskip")
        run("datconv", "-din", tmp_path / "a.f32", "-dout", tmp_path / "b.f32", "-noskip")
        assert md5(tmp_path / "a.txt") == md5(tmp_path / "b.txt"), name          # raw -> text == text -> text ("%g")
        assert md5(tmp_path / "a.f32") == md5(tmp_path / "b.f32"), name          # raw -> raw is the identity
    xm, tailm = _f32_payload(tmp_path / "a.f32")
    assert len(tailm.splitlines()) == xm.shape[0]                    # ex1.dat: one label line per row
    # a map file keeps its header (topology, size, neighbourhood); a -cout name ending in .f32 selects the raw format
    run("randinit", "-din", ex, "-cout", tmp_path / "i.f32", "-xdim", 12, "-ydim", 8, "-topol", "hexa", "-neigh", "bubble", "-rand", 5, "-v", 0)
    run("randinit", "-din", ex, "-cout", tmp_path / "i.cod", "-xdim", 12, "-ydim", 8, "-topol", "hexa", "-neigh", "bubble", "-rand", 5, "-v", 0)
    run("datconv", "-din", tmp_path / "i.f32", "-dout", tmp_path / "i.txt", "-text")
    want_lines = [ln for ln in open(tmp_path / "i.cod").read().split("\n") if not ln.startswith("#")]
    assert open(tmp_path / "i.txt").read().split("\n") == want_lines and want_lines[0] == "5 hexa 12 8 bubble"


def test_generator_source_equals_its_restatement(tools, tmp_path):
    """-din gen:... (paklib.c pak_gen_row) against the numpy restatement in engine.gen_rows, bit for bit, and
    the stream does not depend on where a window starts."""
    import numpy as np
    from som_lvq_pak_amd import engine as E
    run("datconv", "-din", "gen:k=7,dim=13,n=300,seed=4242,labels=1", "-dout", tmp_path / "g.f32")
    x, tail = _f32_payload(tmp_path / "g.f32")
    gx, gc = E.gen_rows(4242, 7, 13, 0, 300)
    assert np.array_equal(x.view(np.uint32), gx.view(np.uint32))
    assert [int(t[1:]) for t in tail.split()] == gc.tolist()
    wx, wc = E.gen_rows(4242, 7, 13, 100, 50)
    assert np.array_equal(wx.view(np.uint32), gx[100:150].view(np.uint32)) and np.array_equal(wc, gc[100:150])
    assert abs(float(gx.mean())) < 1.5 and 2.5 < float(gx.std()) < 5.5     # 4 z centres + unit noise
    assert run("datconv", "-din", "gen:dim=3", "-dout", tmp_path / "bad", check=False).returncode != 0


@pytest.mark.gpu
def test_tools_read_raw_fp32_like_text(tools, tmp_path):
    """vsom / qerror on ex.f32 give the bytes the reference gave on ex.dat; lvq1 / accuracy on the labelled
    ex1/ex2 pair likewise."""
    run("datconv", "-din", os.path.join(DATA, "ex.dat"), "-dout", tmp_path / "ex.f32")
    ex = EXPECTED["som"]["hexa_bubble"]
    out = tmp_path / "out.cod"
    run("vsom", "-din", tmp_path / "ex.f32", "-cin", os.path.join(CLI, ex["init"]), "-cout", out,
        "-rlen", ex["rlen"], "-alpha", ex["alpha"], "-radius", ex["radius"], "-v", 0)
    assert md5(out) == ex["md5"]
    assert run("qerror", "-din", tmp_path / "ex.f32", "-cin", out, "-v", 0).stdout == ex["qerror_stdout"]
    for name in ("ex1", "ex2"):
        run("datconv", "-din", os.path.join(DATA, name + ".dat"), "-dout", tmp_path / (name + ".f32"))
    run("eveninit", "-din", tmp_path / "ex1.f32", "-cout", tmp_path / "a.cod", "-noc", 200, "-v", 0)
    run("eveninit", "-din", os.path.join(DATA, "ex1.dat"), "-cout", tmp_path / "b.cod", "-noc", 200, "-v", 0)
    assert md5(tmp_path / "a.cod") == md5(tmp_path / "b.cod")
    run("lvq1", "-din", tmp_path / "ex1.f32", "-cin", tmp_path / "a.cod", "-cout", tmp_path / "a1.cod", "-rlen", 2000, "-alpha", 0.05, "-v", 0)
    run("lvq1", "-din", os.path.join(DATA, "ex1.dat"), "-cin", tmp_path / "b.cod", "-cout", tmp_path / "b1.cod", "-rlen", 2000, "-alpha", 0.05, "-v", 0)
    assert md5(tmp_path / "a1.cod") == md5(tmp_path / "b1.cod")
    pa = run("accuracy", "-din", tmp_path / "ex2.f32", "-cin", tmp_path / "a1.cod", "-v", 0).stdout
    pb = run("accuracy", "-din", os.path.join(DATA, "ex2.dat"), "-cin", tmp_path / "b1.cod", "-v", 0).stdout
    assert pa == pb and "Total accuracy" in pa or pa == pb


@pytest.mark.gpu
def test_vsom_on_generated_source(tools, tmp_path):
    """`-din gen:...` is a data source like any file: training on it equals training on the same rows written out."""
    spec = "gen:k=6,dim=16,n=1500,seed=31"
    run("datconv", "-din", spec, "-dout", tmp_path / "g.f32")
    run("randinit", "-din", spec, "-cout", tmp_path / "i.cod", "-xdim", 8, "-ydim", 6, "-topol", "hexa", "-neigh", "bubble", "-rand", 5, "-v", 0)
    run("vsom", "-din", spec, "-cin", tmp_path / "i.cod", "-cout", tmp_path / "a.cod", "-rlen", 3000, "-alpha", 0.05, "-radius", 4, "-v", 0)
    run("vsom", "-din", tmp_path / "g.f32", "-cin", tmp_path / "i.cod", "-cout", tmp_path / "b.cod", "-rlen", 3000, "-alpha", 0.05, "-radius", 4, "-v", 0)
    assert md5(tmp_path / "a.cod") == md5(tmp_path / "b.cod")


@pytest.mark.gpu
def test_c2_full_size_matches_reference_cli(tools, tmp_path):
    """BASELINE.json configs[1] at FULL size (32x32 hexa bubble map, 100 000 vectors x 128, -rlen 100000): the
    REAL reference trained on the text form of the generator stream (tests/golden/make_golden.py c2_full_golden);
    here randinit / vsom / qerror read the same stream from `-din gen:...` and must give the reference's bytes --
    100 000 online iterations on the GPU, bit-exact."""
    ex = EXPECTED["c2_full"]
    init, out = tmp_path / "init.cod", tmp_path / "out.cod"
    run("randinit", "-din", ex["gen"], "-cout", init, "-xdim", ex["xdim"], "-ydim", ex["ydim"], "-topol", "hexa",
        "-neigh", "bubble", "-rand", ex["rand"], "-v", 0)
    assert md5(init) == ex["init_md5"]
    run("vsom", "-din", ex["gen"], "-cin", init, "-cout", out, "-rlen", ex["rlen"], "-alpha", ex["alpha"],
        "-radius", ex["radius"], "-v", 0)
    assert md5(out) == ex["md5"]
    assert run("qerror", "-din", ex["gen"], "-cin", out, "-v", 0).stdout == ex["qerror_stdout"]
    # `-batch auto` on this map is the reference's own schedule (the engine's rule does not vouch for mini-batches on
    # 1024 units / 100 000 vectors: somhip_som_auto_batch answers batch 1), so it meets north_star's tolerance against the
    # committed reference qerror by being the reference's result: the same bytes, the same "11.168620"
    auto = tmp_path / "auto.cod"
    run("vsom", "-din", ex["gen"], "-cin", init, "-cout", auto, "-rlen", ex["rlen"], "-alpha", ex["alpha"],
        "-radius", ex["radius"], "-batch", "auto", "-v", 0)
    assert md5(auto) == ex["md5"]
    q = run("qerror", "-din", ex["gen"], "-cin", auto, "-v", 0).stdout
    assert abs(float(q) - float(ex["qerror_stdout"])) <= 1e-4 and q == ex["qerror_stdout"]


def test_raw_fp32_reader_rejects_truncated_files(tools, tmp_path):
    """a payload or a text section shorter than the header promises is an error message and exit 1, not a crash
    (the reader runs clean under -fsanitize=address,undefined on these inputs as well)"""
    import struct
    short = tmp_path / "short.f32"
    short.write_bytes(b"#!somf32 5 0\n3\n" + struct.pack("<6f", *range(6)))
    p = run("datconv", "-din", short, "-dout", tmp_path / "o.f32", check=False)
    assert p.returncode == 1 and "shorter than its header says" in p.stderr
    text = tmp_path / "text.f32"
    text.write_bytes(b"#!somf32 2 1\n2\n" + struct.pack("<4f", 1, 2, 3, 4) + b"lab1\n")
    p = run("datconv", "-din", text, "-dout", tmp_path / "o.f32", check=False)
    assert p.returncode == 1 and "text section ends at row 1" in p.stderr
    ok = tmp_path / "ok.f32"
    ok.write_bytes(b"#!somf32 2 1\n2\n" + struct.pack("<4f", 1, 2, float("nan"), 4) + b"a\nb weight=3\n")
    run("datconv", "-din", ok, "-dout", tmp_path / "ok.txt", "-text")
    # (the text writer prints numbers and labels only, as the reference's write_entry does, datafile.c:420-447)
    assert open(tmp_path / "ok.txt").read().split("\n")[:3] == ["2", "1 2 a ", "x 4 b "]
    run("datconv", "-din", ok, "-dout", tmp_path / "ok2.f32")
    assert open(tmp_path / "ok2.f32", "rb").read().endswith(b"a \nb weight=3 \n")


@pytest.mark.gpu
def test_vsom_gpus_flag_equals_one_gpu(tools, tmp_path):
    """vsom -gpus G (one process per GPU, sharded codebook, all-reduce of winner keys) must write the bytes of the same
    run on one GPU.  Three ranks sharing this box's GPU exchange their keys over host sockets (RCCL refuses duplicate
    devices); SOMHIP_COMM=rccl with one rank sends the same loop through ncclAllReduce(ncclUint64, ncclMin) on the
    engine's stream.  Interleaved 8x8 patches (16x24 map) and contiguous row blocks (12x8 map)."""
    d = os.path.join(DATA, "ex.dat")
    for init_args, tag in ((["-xdim", 16, "-ydim", 24], "patch"), (["-xdim", 12, "-ydim", 8], "rows")):
        init = tmp_path / (tag + "_init.cod")
        run("randinit", "-din", d, "-cout", init, *init_args, "-topol", "hexa", "-neigh", "bubble", "-rand", 5, "-v", 0)
        args = ["-din", d, "-cin", init, "-rlen", 3000, "-alpha", 0.05, "-radius", 6, "-batch", 64, "-v", 0]
        one = tmp_path / (tag + "_one.cod")
        run("vsom", *args, "-cout", one)
        three = tmp_path / (tag + "_three.cod")
        run("vsom", *args, "-cout", three, "-gpus", 3)
        assert md5(three) == md5(one), tag
        rccl = tmp_path / (tag + "_rccl.cod")
        p = subprocess.run([os.path.join(BIN, "vsom")] + [str(a) for a in args] + ["-cout", str(rccl), "-gpus", "1"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SOMHIP_COMM="rccl"))
        assert p.returncode == 0, p.stderr
        assert md5(rccl) == md5(one), tag
    # a generated source: every rank makes the stream in its own HBM
    g = "gen:k=8,dim=24,n=4000,seed=77"
    init = tmp_path / "g_init.cod"
    run("randinit", "-din", g, "-cout", init, "-xdim", 16, "-ydim", 16, "-topol", "rect", "-neigh", "bubble", "-rand", 2, "-v", 0)
    a, b = tmp_path / "g1.cod", tmp_path / "g2.cod"
    run("vsom", "-din", g, "-cin", init, "-cout", a, "-rlen", 4000, "-alpha", 0.04, "-radius", 8, "-batch", 128, "-v", 0)
    run("vsom", "-din", g, "-cin", init, "-cout", b, "-rlen", 4000, "-alpha", 0.04, "-radius", 8, "-batch", 128, "-gpus", 2, "-v", 0)
    assert md5(a) == md5(b)


@pytest.mark.gpu
def test_vsom_gpus_with_exchanged_bounds(tools, tmp_path):
    """vsom -gpus on a shape the two-level pre-filter takes (dim a multiple of 32, batches of 256): the ranks exchange
    the pre-filter's bounds inside every winner search (somhip_shard_winner_begin/refine/finish + the float MIN
    all-reduce of the C host; on by itself from 8 ranks on, asked for here with SOMHIP_SHARD_EXCHANGE=1).  Same bytes
    as one GPU, and as the ranks without the exchange; the verbose log says which path ran."""
    g = "gen:k=10,dim=32,n=6000,seed=21"
    init = tmp_path / "x_init.cod"
    run("randinit", "-din", g, "-cout", init, "-xdim", 24, "-ydim", 24, "-topol", "hexa", "-neigh", "bubble", "-rand", 3, "-v", 0)
    common = ["-din", g, "-cin", init, "-rlen", 6000, "-alpha", 0.05, "-radius", 9, "-batch", 256]
    one, two, plain = tmp_path / "x1.cod", tmp_path / "x2.cod", tmp_path / "x2p.cod"
    run("vsom", *common, "-cout", one, "-v", 0)
    p = subprocess.run([os.path.join(BIN, "vsom")] + [str(a) for a in common] + ["-cout", str(two), "-gpus", "3", "-v", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SOMHIP_SHARD_EXCHANGE="1"))
    assert p.returncode == 0, p.stderr
    assert "bounds exchanged" in p.stderr, p.stderr
    p = subprocess.run([os.path.join(BIN, "vsom")] + [str(a) for a in common] + ["-cout", str(plain), "-gpus", "3", "-v", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SOMHIP_NO_SHARD_EXCHANGE="1"))
    assert p.returncode == 0, p.stderr
    assert "bounds exchanged" not in p.stderr
    assert md5(one) == md5(two) == md5(plain)


@pytest.mark.gpu
def test_vsom_batch_auto(tools, tmp_path):
    """vsom -batch auto: the engine's own batch boundaries (somhip_som_auto_batch, a rule in (units, radius, alpha)).  On a
    map the rule vouches for (128 x 128, 100 long batches' worth of iterations): one GPU == two ranks (every rank asks the library for
    the same boundaries), and != a fixed -batch 32768 run (the schedule really differs).  On a small map `auto` is the
    reference's own online schedule: the bytes of -batch 1 (= no -batch flag at all)."""
    g = "gen:k=6,dim=8,n=40000,seed=5"          # (at least one long batch of rows: the fast update kernels take no run longer than the data)
    init = tmp_path / "a_init.cod"
    run("randinit", "-din", g, "-cout", init, "-xdim", 128, "-ydim", 128, "-topol", "hexa", "-neigh", "bubble", "-rand", 4, "-v", 0)
    L = 100 * 32768 + 5000                       # (the rule wants >= 64 long batches before its short ones: with 80 it answers batch 1)
    common = ["-din", g, "-cin", init, "-rlen", L, "-alpha", 0.05, "-radius", 64, "-v", 0]
    a, b, c = tmp_path / "auto1.cod", tmp_path / "auto2.cod", tmp_path / "fixed.cod"
    run("vsom", *common, "-batch", "auto", "-cout", a)
    run("vsom", *common, "-batch", "auto", "-cout", b, "-gpus", 2)
    run("vsom", *common, "-batch", 32768, "-cout", c)
    assert md5(a) == md5(b)
    assert md5(a) != md5(c)
    small = tmp_path / "s_init.cod"
    run("randinit", "-din", g, "-cout", small, "-xdim", 16, "-ydim", 16, "-topol", "hexa", "-neigh", "bubble", "-rand", 4, "-v", 0)
    common = ["-din", g, "-cin", small, "-rlen", 30000, "-alpha", 0.05, "-radius", 8, "-v", 0]
    d, e = tmp_path / "s_auto.cod", tmp_path / "s_online.cod"
    run("vsom", *common, "-batch", "auto", "-cout", d)
    run("vsom", *common, "-cout", e)
    assert md5(d) == md5(e)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["lvq1", "lvq2", "lvq3", "olvq1", "olvq1_default"])
def test_lvqtrain_gpus_flag_gives_the_reference_bytes(tools, tmp_path, tag):
    """lvqtrain -gpus 3: the codebook's rows cut into three blocks, one process each (sharing this box's GPU: candidate
    lists and rows travel over host sockets; SOMHIP_COMM=rccl with one rank goes through ncclAllGather / ncclAllReduce).
    The LVQ loops over a sharded codebook are exact: the reference's own bytes."""
    ex = EXPECTED["lvq"][tag]
    for extra, env in ((["-gpus", 3], {}), (["-gpus", 1], {"SOMHIP_COMM": "rccl"})):
        out = tmp_path / ("out%s.cod" % extra[1])
        p = subprocess.run([os.path.join(BIN, ex["tool"]), "-din", os.path.join(DATA, "ex1.dat"), "-cin", os.path.join(CLI, "lvq_init.cod"),
                            "-cout", str(out)] + [str(a) for a in ex["args"]] + [str(a) for a in extra] + ["-v", "0"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        assert md5(out) == ex["md5"], extra
        assert not os.path.exists(str(out)[:-4] + ".lra")


# ------------------------------------------------------------------ INTEGRATION.md, proven on the reference's own tools
REF = os.path.join(ROOT, "oracle", "_ref")


def _glued(tool):
    exe = os.path.join(REF, tool)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s not built (needs /root/reference at build time)" % tool)
    return exe


def _run_ref(exe, *args, env=None):
    p = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, (exe, args, p.stderr)
    return p


def test_glued_reference_tools_keep_their_cpu_path(tmp_path):
    """oracle/_ref/vsom_hip = the reference's unmodified vsom.o + library objects + som_lvq_pak_amd/host/glue/somhip_glue.c.
    Without -selfuncs hip the glue must hand every call on to the reference's own loops: the reference's bytes."""
    ex = EXPECTED["som"]["hexa_bubble"]
    out = tmp_path / "o.cod"
    _run_ref(_glued("vsom_hip"), "-din", os.path.join(DATA, "ex.dat"), "-cin", os.path.join(CLI, ex["init"]), "-cout", out,
             "-rlen", ex["rlen"], "-alpha", ex["alpha"], "-radius", ex["radius"], "-v", 0)
    assert md5(out) == ex["md5"]
    p = _run_ref(_glued("qerror_hip"), "-din", os.path.join(DATA, "ex.dat"), "-cin", out, "-v", 0)
    assert p.stdout == ex["qerror_stdout"]


@pytest.mark.gpu
def test_reference_tools_linked_with_the_glue(tmp_path):
    """The reference's OWN host code (argument parsing, .dat/.cod I/O, labels, snapshots: vsom.o, qerror.o, lvqtrain.o and
    the library objects compiled from /root/reference, unmodified) with `-selfuncs hip`: som_training / find_qerror /
    lvq*_training go to the MI355X engine through the committed glue and must reproduce what the reference's CPU loops
    wrote (tests/golden/cli/expected.json)."""
    vsom, qerr, lvq = _glued("vsom_hip"), _glued("qerror_hip"), _glued("lvqtrain_hip")
    d = os.path.join(DATA, "ex.dat")
    for tag in ("hexa_bubble", "hexa_gaussian", "rect_bubble", "rect_gaussian"):
        ex = EXPECTED["som"][tag]
        out = tmp_path / (tag + ".cod")
        p = _run_ref(vsom, "-din", d, "-cin", os.path.join(CLI, ex["init"]), "-cout", out, "-rlen", ex["rlen"], "-alpha", ex["alpha"],
                     "-radius", ex["radius"], "-selfuncs", "hip", "-v", 0)
        assert "not found" not in p.stderr                   # the name reached the glue, not the reference's "unknown row" warning
        assert md5(out) == ex["md5"], tag
        assert _run_ref(qerr, "-din", d, "-cin", out, "-selfuncs", "hip", "-v", 0).stdout == ex["qerror_stdout"], tag
    out = tmp_path / "invt.cod"
    _run_ref(vsom, "-din", d, "-cin", os.path.join(CLI, "som_init_hexa_bubble.cod"), "-cout", out, "-rlen", 5000, "-alpha", 0.05,
             "-radius", 10, "-alpha_type", "inverse_t", "-rand", 7, "-selfuncs", "hip", "-v", 0)
    ref = tmp_path / "invt_ref.cod"
    _run_ref(vsom, "-din", d, "-cin", os.path.join(CLI, "som_init_hexa_bubble.cod"), "-cout", ref, "-rlen", 5000, "-alpha", 0.05,
             "-radius", 10, "-alpha_type", "inverse_t", "-rand", 7, "-v", 0)
    assert md5(out) == md5(ref)                              # -rand shuffle + inverse_t: engine == the same binary's CPU loop
    # snapshots through the reference's own save_snapshot
    ex = EXPECTED["buffer_snap"]["vsom_snap1"]
    out = tmp_path / "snap.cod"
    _run_ref(vsom, "-din", d, "-cin", os.path.join(CLI, ex["cin"]), "-cout", out, *ex["args"], "-snapfile",
             str(tmp_path / "s1_%ld.snap"), "-selfuncs", "hip", "-v", 0)
    assert md5(out) == ex["md5"]
    for it, want in ex["snapshots"].items():
        assert md5(tmp_path / ("s1_%s.snap" % it)) == want, it
    # the LVQ loops (lvqtrain.c never reads -selfuncs: the row is selected through the environment)
    for tag in ("lvq1", "lvq2", "lvq3", "olvq1", "olvq1_default"):
        ex = EXPECTED["lvq"][tag]
        out = tmp_path / (tag + ".cod")
        _run_ref(lvq, "-type", ex["tool"], "-din", os.path.join(DATA, "ex1.dat"), "-cin", os.path.join(CLI, "lvq_init.cod"),
                 "-cout", out, *ex["args"], "-v", 0, env={"SOMHIP_SELFUNCS": "hip"})
        assert md5(out) == ex["md5"], tag


@pytest.mark.gpu
def test_reference_scanners_and_buffers_through_the_glue(tmp_path):
    """VERDICT r2 item 6.  (a) The reference's scanners that call teach->winner once per data vector -- accuracy.o
    (compute_accuracy, accuracy.c:80-113), vcal.o (find_labels, vcal.c:106-129), visual.o, classify.o, knntest.o, unmodified
    -- with `-selfuncs hip`: the glue's winner answers from ONE GPU scan of the data list (stderr at -v 2 says so) and the
    tools print / write the reference's bytes, with and without -buffer.  (b) -buffer N training through the glue: one data
    set per loaded buffer, the reference's own loads and -rand shuffles: the reference's .cod and snapshot bytes
    (tests/golden/cli/expected.json buffer_rand / buffer_snap).  (c) -batch B read by the glue with extract_parameter."""
    t = EXPECTED["lvq"]["tools"]
    cod = os.path.join(CLI, "lvq_olvq1.cod")
    ex2 = os.path.join(DATA, "ex2.dat")
    # ---- (a) scanners
    acc = _glued("accuracy_hip")
    cpu = _run_ref(acc, "-din", ex2, "-cin", cod, "-v", 0).stdout                   # the same binary's CPU row
    for extra in ([], ["-buffer", 300]):
        p = _run_ref(acc, "-din", ex2, "-cin", cod, "-selfuncs", "hip", "-v", 2, *extra)
        assert p.stdout == cpu and "Total accuracy" in cpu, extra
        assert "data vectors against 200 codes on the GPU" in p.stderr, p.stderr
        assert p.stderr.count("on the GPU") == (1 if not extra else 7)                # 1962 vectors: one scan, or one per buffer of 300
    p = _run_ref(acc, "-din", ex2, "-cin", os.path.join(CLI, "lvq_init.cod"), "-selfuncs", "hip", "-v", 0)
    assert p.stdout == _run_ref(acc, "-din", ex2, "-cin", os.path.join(CLI, "lvq_init.cod"), "-v", 0).stdout
    kn = _glued("knntest_hip")
    for knn in (1, 3, 5):
        p = _run_ref(kn, "-din", ex2, "-cin", cod, "-knn", knn, "-selfuncs", "hip", "-v", 0)
        assert p.stdout == t["knntest_%d" % knn], knn
    cl = _glued("classify_hip")
    _run_ref(cl, "-din", ex2, "-cin", cod, "-dout", tmp_path / "cls.dat", "-cfout", tmp_path / "cls.cfo", "-selfuncs", "hip", "-v", 0)
    assert md5(tmp_path / "cls.dat") == t["classify_dout_md5"] and md5(tmp_path / "cls.cfo") == t["classify_cfout_md5"]
    som = os.path.join(CLI, EXPECTED["som"]["hexa_bubble"]["init"])
    exd = os.path.join(DATA, "ex.dat")
    vcal, vis = _glued("vcal_hip"), _glued("visual_hip")
    fts = os.path.join(DATA, "ex_fts.dat")                                          # labelled data (vcal needs labels)
    for extra in ([], ["-buffer", 100]):
        _run_ref(vcal, "-din", fts, "-cin", som, "-cout", tmp_path / "cal_cpu.cod", "-v", 0, *extra)
        p = _run_ref(vcal, "-din", fts, "-cin", som, "-cout", tmp_path / "cal_hip.cod", "-selfuncs", "hip", "-v", 2, *extra)
        assert md5(tmp_path / "cal_cpu.cod") == md5(tmp_path / "cal_hip.cod") and "on the GPU" in p.stderr, extra
    _run_ref(vis, "-din", exd, "-cin", som, "-dout", tmp_path / "vis_cpu.out", "-v", 0)
    _run_ref(vis, "-din", exd, "-cin", som, "-dout", tmp_path / "vis_hip.out", "-selfuncs", "hip", "-v", 0)
    assert md5(tmp_path / "vis_cpu.out") == md5(tmp_path / "vis_hip.out")
    # ---- (b) -buffer N training: the reference's bytes
    vsom, lvq = _glued("vsom_hip"), _glued("lvqtrain_hip")
    for group in ("buffer_rand", "buffer_snap"):
        for tag, ex in EXPECTED[group].items():
            if "-buffer" not in [str(a) for a in ex["args"]]:
                continue
            out = tmp_path / (tag + ".cod")
            snap = ["-snapfile", str(tmp_path / (tag + "_%ld.snap"))] if "snapshots" in ex else []
            if ex["tool"] == "vsom":
                _run_ref(vsom, "-din", os.path.join(DATA, ex["data"]), "-cin", os.path.join(CLI, ex["cin"]), "-cout", out, *ex["args"],
                         *snap, "-selfuncs", "hip", "-v", 0)
            else:
                _run_ref(lvq, "-type", ex["tool"], "-din", os.path.join(DATA, ex["data"]), "-cin", os.path.join(CLI, ex["cin"]), "-cout", out,
                         *ex["args"], *snap, "-v", 0, env={"SOMHIP_SELFUNCS": "hip"})
            assert md5(out) == ex["md5"], tag
            for it, want in ex.get("snapshots", {}).items():
                assert md5(tmp_path / ("%s_%s.snap" % (tag, it))) == want, (tag, it)
    # qerror over a buffered data file
    ex = EXPECTED["som"]["hexa_bubble"]
    trained = tmp_path / "hb.cod"
    _run_ref(vsom, "-din", exd, "-cin", os.path.join(CLI, ex["init"]), "-cout", trained, "-rlen", ex["rlen"], "-alpha", ex["alpha"],
             "-radius", ex["radius"], "-selfuncs", "hip", "-v", 0)
    assert _run_ref(_glued("qerror_hip"), "-din", exd, "-cin", trained, "-buffer", 700, "-selfuncs", "hip", "-v", 0).stdout == ex["qerror_stdout"]
    # ---- (c) -batch B through the glue == the product's vsom -batch B (same engine call)
    a, b = tmp_path / "gb.cod", tmp_path / "pb.cod"
    _run_ref(vsom, "-din", exd, "-cin", os.path.join(CLI, ex["init"]), "-cout", a, "-rlen", 2000, "-alpha", 0.05, "-radius", 8,
             "-batch", 64, "-selfuncs", "hip", "-v", 0)
    _run_ref(os.path.join(BIN, "vsom"), "-din", exd, "-cin", os.path.join(CLI, ex["init"]), "-cout", b, "-rlen", 2000, "-alpha", 0.05,
             "-radius", 8, "-batch", 64, "-v", 0)
    assert md5(a) == md5(b) and md5(a) != ex["md5"]
