#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref, built from /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The fixtures it writes are data (inputs + the reference's outputs); they are committed so
the GPU box -- which has no /root/reference -- can check against them.

  data/            the reference's bundled example data sets (inputs)
  cli/*.cod        codebooks written by the reference's own tools ("%g" text)
  cli/expected.json  what the tools printed (qerror, accuracy) + md5 of each .cod
  traces/*.npz     in-memory fp32 results of som_training / lvq*_training driven through
                   oracle/ref_harness.c, with the (index, diff) of every winner call
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import RefHarness, build, ref_tool  # noqa: E402
from som_lvq_pak_amd import textio  # noqa: E402

REF_SRC = "/root/reference"
DATA = os.path.join(HERE, "data")
CLI = os.path.join(HERE, "cli")
TR = os.path.join(HERE, "traces")


def run(tool, *args):
    cmd = [ref_tool(tool)] + [str(a) for a in args] + ["-v", "0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=CLI)
    if p.returncode != 0:
        raise RuntimeError("%s failed: %s" % (cmd, p.stderr))
    return p.stdout


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def synth(seed, n, d, k=6, spread=4.0):
    """Seeded Gaussian mixture (legacy RandomState: stable across numpy versions)."""
    rs = np.random.RandomState(seed)
    centres = (spread * rs.standard_normal((k, d))).astype(np.float32)
    which = rs.randint(0, k, size=n)
    x = centres[which] + rs.standard_normal((n, d)).astype(np.float32)
    return x.astype(np.float32), which.astype(np.int32) + 1


VFIND_ANSWERS = ["3", "{data}", "{data}", "{out}", "hexa", "bubble", "6", "5", "800", "0.05", "5", "2000", "0.02", "2"]


def vfind_golden(exp, d):
    """vfind: 3 trials of randinit -> vsom -> vsom -> qerror, best map kept (answers on stdin)"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for tag, extra in (("q0", []), ("q1", ["-qetype", "1"])):
            out = os.path.join(tmp, tag + ".cod")
            ans = "\n".join(a.format(data=d("ex.dat"), out=out) for a in VFIND_ANSWERS) + "\n"
            p = subprocess.run([ref_tool("vfind")] + extra, input=ans, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               text=True, cwd=CLI)
            assert p.returncode == 0, p.stderr
            res[tag] = {"args": extra, "md5": md5(out), "trials_stderr": [l for l in p.stderr.splitlines() if ": " in l and l.strip()[0].isdigit()],
                        "last_stdout_line": p.stdout.strip().splitlines()[-1]}
        exp["som"]["vfind"] = res


def buffer_golden(exp, d):
    """-buffer N with -rand: every buffer is reshuffled when it is (re)loaded (datafile.c:237-344)"""
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, tool, data, cin, args in (
                ("vsom_b500_r7", "vsom", "ex.dat", "som_init_hexa_bubble.cod",
                 ["-rlen", 9000, "-alpha", 0.05, "-radius", 10, "-rand", 7, "-buffer", 500]),
                ("vsom_b3000_r2_gauss", "vsom", "ex.dat", "som_init_hexa_gaussian.cod",
                 ["-rlen", 4000, "-alpha", 0.04, "-radius", 6, "-rand", 2, "-buffer", 3000]),
                ("lvq1_b300_r3", "lvq1", "ex1.dat", "lvq_init.cod", ["-rlen", 5000, "-alpha", 0.05, "-rand", 3, "-buffer", 300]),
                ("olvq1_b777_r9", "olvq1", "ex1.dat", "lvq_init.cod", ["-rlen", 4000, "-rand", 9, "-buffer", 777]),
                ("lvq3_b5000_r4", "lvq3", "ex1.dat", "lvq_init.cod",      # buffer > file: shuffled once
                 ["-rlen", 3000, "-alpha", 0.05, "-win", 0.3, "-epsilon", 0.1, "-rand", 4, "-buffer", 5000])):
            out = os.path.join(tmp, tag + ".cod")
            run(tool, "-din", d(data), "-cin", cin, "-cout", out, *args)
            res[tag] = {"tool": tool, "data": data, "cin": cin, "args": [str(a) for a in args], "md5": md5(out)}
    exp["buffer_rand"] = res
    # -buffer N together with -snapinterval: segments cut by the buffer feed may start exactly on a snapshot
    # iteration (som_rout.c:650, lvq_rout.c:559 save after every le % interval == 0, le > 0)
    snaps = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, tool, data, cin, args, its in (
                ("vsom_b1000_snap1000", "vsom", "ex.dat", "som_init_hexa_bubble.cod",
                 ["-rlen", 4500, "-alpha", 0.05, "-radius", 10, "-rand", 5, "-buffer", 1000, "-snapinterval", 1000], (1000, 2000, 3000, 4000)),
                ("vsom_snap1", "vsom", "ex.dat", "som_init_hexa_bubble.cod",
                 ["-rlen", 6, "-alpha", 0.05, "-radius", 10, "-snapinterval", 1], (1, 2, 3, 4, 5)),
                ("lvq1_b500_snap250", "lvq1", "ex1.dat", "lvq_init.cod",
                 ["-rlen", 1200, "-alpha", 0.05, "-rand", 3, "-buffer", 500, "-snapinterval", 250], (250, 500, 750, 1000))):
            out = os.path.join(tmp, tag + ".cod")
            run(tool, "-din", d(data), "-cin", cin, "-cout", out, *args, "-snapfile", os.path.join(tmp, tag + "_%ld.snap"))
            files = sorted(f for f in os.listdir(tmp) if f.startswith(tag + "_") and f.endswith(".snap"))
            assert files == sorted("%s_%d.snap" % (tag, i) for i in its), files
            snaps[tag] = {"tool": tool, "data": data, "cin": cin, "args": [str(a) for a in args], "md5": md5(out),
                          "snapshots": {str(i): md5(os.path.join(tmp, "%s_%d.snap" % (tag, i))) for i in its}}
    exp["buffer_snap"] = snaps


def lininit_golden(exp, d):
    """lininit on ex.dat and on a copy of its first 500 rows with every 7th value masked ('x')"""
    import tempfile
    lines = open(d("ex.dat")).read().splitlines()
    rs = np.random.RandomState(99)
    out_lines = [lines[0]]
    for ln in lines[1:501]:
        toks = ln.split()
        out_lines.append(" ".join("x" if rs.random_sample() < 0.14 else t for t in toks))
    open(d("ex_masked.dat"), "w").write("\n".join(out_lines) + "\n")
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, data, args in (("ex_hexa", "ex.dat", ["-xdim", 12, "-ydim", 8, "-topol", "hexa", "-neigh", "bubble", "-rand", 123]),
                                ("ex_rect_seed5", "ex.dat", ["-xdim", 7, "-ydim", 9, "-topol", "rect", "-neigh", "gaussian", "-rand", 5]),
                                ("masked", "ex_masked.dat", ["-xdim", 10, "-ydim", 6, "-topol", "hexa", "-neigh", "bubble", "-rand", 11])):
            out = os.path.join(tmp, tag + ".cod")
            run("lininit", "-din", d(data), "-cout", out, *args)
            res[tag] = {"data": data, "args": [str(a) for a in args], "md5": md5(out)}
    exp["som"]["lininit"] = res


def lvq_tool_goldens(exp, d):
    """the k-NN consumers around the LVQ loops: propinit / eveninit -knn, knntest, classify"""
    import tempfile
    t = {}
    with tempfile.TemporaryDirectory() as tmp:
        o = lambda f: os.path.join(tmp, f)  # noqa: E731
        for tag, tool, args in (("propinit_200", "propinit", ["-noc", 200]),
                                ("eveninit_knn3_100", "eveninit", ["-noc", 100, "-knn", 3]),
                                ("propinit_knn1_60", "propinit", ["-noc", 60, "-knn", 1]),
                                ("eveninit_800", "eveninit", ["-noc", 800])):     # needs the second pass
            run(tool, "-din", d("ex1.dat"), "-cout", o(tag + ".cod"), *args)
            t[tag] = {"tool": tool, "args": [str(a) for a in args], "md5": md5(o(tag + ".cod"))}
        for knn in (1, 3, 5):
            t["knntest_%d" % knn] = run("knntest", "-din", d("ex2.dat"), "-cin", "lvq_olvq1.cod", "-knn", knn)
        run("classify", "-din", d("ex2.dat"), "-cin", "lvq_olvq1.cod", "-dout", o("cls.dat"), "-cfout", o("cls.cfo"))
        t["classify_dout_md5"] = md5(o("cls.dat"))
        t["classify_cfout_md5"] = md5(o("cls.cfo"))
        # cmatr, setlabel (codes relabelled by their 3 and 5 nearest data vectors), elimin
        t["cmatr"] = run("cmatr", "-din", d("ex2.dat"), "-cin", "lvq_olvq1.cod", "-cfout", o("cm.cfo"))
        t["cmatr_cfout_md5"] = md5(o("cm.cfo"))
        for knn in (3, 5):
            run("setlabel", "-din", d("ex2.dat"), "-cin", "lvq_olvq1.cod", "-cout", o("sl.cod"), "-knn", knn)
            t["setlabel_%d_md5" % knn] = md5(o("sl.cod"))
        for knn in (3, 5, 8):
            run("elimin", "-din", d("ex1.dat"), "-cout", o("el.cod"), "-knn", knn)
            t["elimin_%d_md5" % knn] = md5(o("el.cod"))
        # balance.  The reference never counts the codes it appends (balance.c:188), so its
        # olvq1_training indexes its learning-rate array past the end for them: most inputs give
        # results that change with MALLOC_PERTURB_ (or abort in malloc).  These two do not -- the
        # appended codes only ever win their own sample, which moves nothing.
        run("eveninit", "-din", d("ex1.dat"), "-cout", o("even400.cod"), "-noc", 400)
        for tag, cin, args in (("balance_even", os.path.join(CLI, "lvq_init.cod"), []),
                               ("balance_even400_knn3", o("even400.cod"), ["-knn", 3])):
            out = run("balance", "-din", d("ex1.dat"), "-cin", cin, "-cout", o(tag + ".cod"), *args)
            t[tag] = {"args": [str(a) for a in args], "stdout": out, "md5": md5(o(tag + ".cod")),
                      "lra_md5": md5(o(tag + ".lra"))}
    exp["lvq"]["tools"] = t


def c2_full_golden(exp):
    """BASELINE.json configs[1] at full size through the REAL reference: 100 000 vectors x 128 of the seeded
    generator stream (engine.gen_rows == paklib.c pak_gen_row == k_gen_mixture), written as text with %.9g (exact
    round trip through sscanf("%f")), randinit 32x32 hexa bubble -rand 7, vsom -rlen 100000 -alpha 0.05 -radius 10,
    qerror.  Only hashes and stdout are kept; the GPU test regenerates the data from the same spec."""
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from som_lvq_pak_amd import engine as E
    spec = {"k": 16, "dim": 128, "n": 100000, "seed": 1234}
    x, _ = E.gen_rows(spec["seed"], spec["k"], spec["dim"], 0, spec["n"])
    with tempfile.TemporaryDirectory() as td:
        dat = os.path.join(td, "c2.dat")
        with open(dat, "w") as f:
            f.write("%d\n" % spec["dim"])
            for row in x:
                f.write(" ".join("%.9g" % v for v in row) + "\n")
        init, out = os.path.join(td, "init.cod"), os.path.join(td, "out.cod")
        run("randinit", "-din", dat, "-cout", init, "-xdim", 32, "-ydim", 32, "-topol", "hexa", "-neigh", "bubble", "-rand", 7)
        run("vsom", "-din", dat, "-cin", init, "-cout", out, "-rlen", 100000, "-alpha", 0.05, "-radius", 10)
        exp["c2_full"] = {"gen": "gen:k=%(k)d,dim=%(dim)d,n=%(n)d,seed=%(seed)d" % spec, "xdim": 32, "ydim": 32, "rand": 7,
                          "rlen": 100000, "alpha": 0.05, "radius": 10, "init_md5": md5(init), "md5": md5(out),
                          "qerror_stdout": run("qerror", "-din", dat, "-cin", out)}


def c3_prefix_golden(exp):
    """BASELINE.json configs[2] (OLVQ1, 10 000 codes x 256, alpha0 0.3) through the REAL reference's olvq1_training
    (oracle/ref_harness.c over the reference's own objects) on the first 20 000 vectors of the generator stream
    gen:k=100,dim=256,seed=2345 -- OLVQ1 has no global schedule (lvq_rout.c:596), so this IS the state of the full
    1 M-vector run after 20 000 iterations.  Initial codes = the first 100 samples of each class.  Only
    hashes are kept; the GPU test regenerates stream and codes from the same spec."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from som_lvq_pak_amd import engine as E
    spec = {"k": 100, "dim": 256, "seed": 2345, "codes_per_class": 100, "prefix": 20000, "head": 40000}
    x, cen = E.gen_rows(spec["seed"], spec["k"], spec["dim"], 0, spec["head"])
    pick = np.concatenate([np.where(cen == c)[0][:spec["codes_per_class"]] for c in range(spec["k"])])
    assert len(pick) == spec["k"] * spec["codes_per_class"]
    codes, clab = x[pick].copy(), cen[pick].astype(np.int32)
    ref = RefHarness()
    oc, ol, oi, od = ref.lvq_train(2, codes, clab, x[:spec["prefix"]], cen[:spec["prefix"]].astype(np.int32), spec["prefix"], 0.3)
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    # the reference hands the rates over as the "%g" text of its .lra file (datafile.c:1081)
    rates = hashlib.sha256(" ".join(ol).encode()).hexdigest()
    exp["c3_prefix"] = dict(spec, alpha=0.3, codes_sha256=h(oc.astype(np.float32)), rates_g_sha256=rates,
                            winners_sha256=h(oi.astype(np.int32)), diffs_sha256=h(od.astype(np.float32)),
                            init_sha256=h(codes))


def main():
    if "--c3" in sys.argv:
        build()
        exp = json.load(open(os.path.join(CLI, "expected.json")))
        c3_prefix_golden(exp)
        json.dump(exp, open(os.path.join(CLI, "expected.json"), "w"), indent=1, sort_keys=True)
        return
    if "--c2" in sys.argv:                 # refresh only that section of expected.json
        build()
        exp = json.load(open(os.path.join(CLI, "expected.json")))
        c2_full_golden(exp)
        json.dump(exp, open(os.path.join(CLI, "expected.json"), "w"), indent=1, sort_keys=True)
        return
    if "--lvq-tools" in sys.argv:          # refresh only that section of expected.json
        build()
        exp = json.load(open(os.path.join(CLI, "expected.json")))
        lvq_tool_goldens(exp, lambda f: os.path.join(DATA, f))
        vfind_golden(exp, lambda f: os.path.join(DATA, f))
        lininit_golden(exp, lambda f: os.path.join(DATA, f))
        buffer_golden(exp, lambda f: os.path.join(DATA, f))
        json.dump(exp, open(os.path.join(CLI, "expected.json"), "w"), indent=1, sort_keys=True)
        return
    build()
    for p in (DATA, CLI, TR):
        os.makedirs(p, exist_ok=True)
    for f in ("ex.dat", "ex_fts.dat", "ex1.dat", "ex2.dat"):
        shutil.copyfile(os.path.join(REF_SRC, f), os.path.join(DATA, f))
        os.chmod(os.path.join(DATA, f), 0o644)
    d = lambda f: os.path.join(DATA, f)  # noqa: E731
    exp = {"som": {}, "lvq": {}}

    # ---------------- SOM chains through the reference's own CLI ----------------
    for topol in ("hexa", "rect"):
        for neigh in ("bubble", "gaussian"):
            tag = "%s_%s" % (topol, neigh)
            init = "som_init_%s.cod" % tag
            run("randinit", "-din", d("ex.dat"), "-cout", init, "-xdim", 12, "-ydim", 8,
                "-topol", topol, "-neigh", neigh, "-rand", 123)
            out = "som_%s.cod" % tag
            run("vsom", "-din", d("ex.dat"), "-cin", init, "-cout", out, "-rlen", 5000,
                "-alpha", 0.05, "-radius", 10)
            q = run("qerror", "-din", d("ex.dat"), "-cin", out)
            exp["som"][tag] = {"init": init, "out": out, "rlen": 5000, "alpha": 0.05, "radius": 10,
                               "qerror_stdout": q, "md5": md5(os.path.join(CLI, out))}
    init = "som_init_hexa_bubble.cod"
    run("vsom", "-din", d("ex.dat"), "-cin", init, "-cout", "som_invt.cod", "-rlen", 5000,
        "-alpha", 0.05, "-radius", 10, "-alpha_type", "inverse_t")
    exp["som"]["inverse_t"] = {"init": init, "out": "som_invt.cod",
                               "qerror_stdout": run("qerror", "-din", d("ex.dat"), "-cin", "som_invt.cod"),
                               "md5": md5(os.path.join(CLI, "som_invt.cod"))}
    run("vsom", "-din", d("ex.dat"), "-cin", init, "-cout", "som_rand7.cod", "-rlen", 5000,
        "-alpha", 0.05, "-radius", 10, "-rand", 7)
    exp["som"]["rand7"] = {"init": init, "out": "som_rand7.cod",
                           "qerror_stdout": run("qerror", "-din", d("ex.dat"), "-cin", "som_rand7.cod"),
                           "md5": md5(os.path.join(CLI, "som_rand7.cod"))}
    exp["som"]["qerror2_r2"] = run("qerror", "-din", d("ex.dat"), "-cin", "som_hexa_bubble.cod",
                                   "-qetype", 1, "-radius", 2)
    # the reference Makefile's `somexample` (Makefile:195-205)
    shutil.copyfile(os.path.join(CLI, init), os.path.join(CLI, "somexample.cod"))
    run("vsom", "-din", d("ex.dat"), "-cin", "somexample.cod", "-cout", "somexample.cod",
        "-rlen", 1000, "-alpha", 0.05, "-radius", 10)
    run("vsom", "-din", d("ex.dat"), "-cin", "somexample.cod", "-cout", "somexample.cod",
        "-rlen", 10000, "-alpha", 0.02, "-radius", 3)
    exp["som"]["somexample"] = {"qerror_stdout": run("qerror", "-din", d("ex.dat"), "-cin", "somexample.cod"),
                                "md5": md5(os.path.join(CLI, "somexample.cod"))}
    shutil.copyfile(os.path.join(CLI, "somexample.cod"), os.path.join(CLI, "somexample_vcal.cod"))
    run("vcal", "-din", d("ex_fts.dat"), "-cin", "somexample.cod", "-cout", "somexample_vcal.cod")
    exp["som"]["somexample_vcal_md5"] = md5(os.path.join(CLI, "somexample_vcal.cod"))
    run("visual", "-din", d("ex_fts.dat"), "-cin", "somexample_vcal.cod", "-dout", "somexample_fts.vis")
    exp["som"]["somexample_vis_md5"] = md5(os.path.join(CLI, "somexample_fts.vis"))
    exp["som"]["randinit_md5"] = md5(os.path.join(CLI, "som_init_hexa_bubble.cod"))

    # ---------------- LVQ chains ----------------
    run("eveninit", "-din", d("ex1.dat"), "-cout", "lvq_init.cod", "-noc", 200)
    exp["lvq"]["init_md5"] = md5(os.path.join(CLI, "lvq_init.cod"))
    runs = {
        "lvq1_10000": ("lvq1", ["-rlen", 10000, "-alpha", 0.05]),
        "lvq1": ("lvq1", ["-rlen", 5000, "-alpha", 0.05]),
        "lvq2": ("lvq2", ["-rlen", 5000, "-alpha", 0.05, "-win", 0.3]),
        "lvq3": ("lvq3", ["-rlen", 5000, "-alpha", 0.05, "-win", 0.3, "-epsilon", 0.1]),
        "olvq1": ("olvq1", ["-rlen", 5000, "-alpha", 0.05]),
        "olvq1_default": ("olvq1", ["-rlen", 5000]),
    }
    for tag, (tool, args) in runs.items():
        out = "lvq_%s.cod" % tag
        run(tool, "-din", d("ex1.dat"), "-cin", "lvq_init.cod", "-cout", out, *args)
        acc = run("accuracy", "-din", d("ex2.dat"), "-cin", out)
        exp["lvq"][tag] = {"tool": tool, "args": [str(a) for a in args], "out": out,
                           "accuracy_stdout": acc, "md5": md5(os.path.join(CLI, out))}
    lvq_tool_goldens(exp, d)
    vfind_golden(exp, d)
    lininit_golden(exp, d)
    buffer_golden(exp, d)
    c2_full_golden(exp)
    c3_prefix_golden(exp)
    json.dump(exp, open(os.path.join(CLI, "expected.json"), "w"), indent=1, sort_keys=True)

    # ---------------- in-memory traces through the harness ----------------
    ref = RefHarness()
    ex, _ = textio.read_entries(d("ex.dat"))
    for topol in ("hexa", "rect"):
        for neigh in ("bubble", "gaussian"):
            tag = "%s_%s" % (topol, neigh)
            ini, _ = textio.read_entries(os.path.join(CLI, "som_init_%s.cod" % tag))
            for at, atname in ((1, "linear"), (2, "inverse_t")):
                if at == 2 and tag != "hexa_bubble":
                    continue
                codes, ti, td = ref.som_train(ini.points, 12, 8, ini.topol, ini.neigh, ex.points,
                                              5000, 0.05, 10.0, alpha_type=at)
                q, qi, qd = ref.find_qerror(codes, ex.points)
                np.savez_compressed(os.path.join(TR, "som_ex_%s_%s.npz" % (tag, atname)),
                                    codes=codes, trace_index=ti.astype(np.int32), trace_diff=td,
                                    qerror_sum=np.float32(q), q_index=qi.astype(np.int32), q_diff=qd,
                                    params=np.array([12, 8, ini.topol, ini.neigh, 5000, at]),
                                    alpha=np.float32(0.05), radius=np.float32(10.0))
    # qerror -qetype 1
    hb, _ = textio.read_entries(os.path.join(CLI, "som_hexa_bubble.cod"))
    hg, _ = textio.read_entries(os.path.join(CLI, "som_hexa_gaussian.cod"))
    np.savez_compressed(os.path.join(TR, "som_qerror2.npz"),
                        bubble_r2=np.float32(ref.find_qerror2(hb.points, 12, 3, 1, ex.points, 2.0)),
                        gaussian_r2=np.float32(ref.find_qerror2(hg.points, 12, 3, 2, ex.points, 2.0)))

    # synthetic: masks + weights + fixed points (the bundled data has none of them)
    x, _ = synth(11, 300, 6)
    rs = np.random.RandomState(12)
    mask = (rs.rand(300, 6) < 0.15).astype(np.uint8)
    mask[7, :] = 1                       # a fully masked sample -> skipped (som_rout.c:635-640)
    weight = rs.randint(0, 4, size=300).astype(np.int16)
    fixed = np.full((300, 2), -1, dtype=np.int16)
    for r in rs.choice(300, 20, replace=False):
        fixed[r] = (rs.randint(0, 7), rs.randint(0, 5))
    ini = ref.randinit(x, 7, 5, 99)
    for neigh in (1, 2):
        codes, ti, td = ref.som_train(ini, 7, 5, 3, neigh, x, 1500, 0.08, 4.0, weight=weight,
                                      fixed_xy=fixed, mask=mask, fixed_on=1, weights_on=1)
        np.savez_compressed(os.path.join(TR, "som_masked_%d.npz" % neigh), init=ini, codes=codes,
                            trace_index=ti.astype(np.int32), trace_diff=td, mask=mask,
                            weight=weight, fixed=fixed, x=x,
                            params=np.array([7, 5, 3, neigh, 1500, 1]),
                            alpha=np.float32(0.08), radius=np.float32(4.0))

    # synthetic mid-size map: only hashes + index trace (data regenerated from the seed)
    x, _ = synth(21, 4000, 48, k=8)
    ini = ref.randinit(x, 24, 16, 5)
    for topol, neigh in ((3, 1), (4, 2)):
        codes, ti, td = ref.som_train(ini, 24, 16, topol, neigh, x, 6000, 0.05, 8.0)
        q, _, _ = ref.find_qerror(codes, x)
        np.savez_compressed(os.path.join(TR, "som_synth_%d_%d.npz" % (topol, neigh)),
                            init_sha=hashlib.sha256(ini.tobytes()).hexdigest(),
                            codes_sha=hashlib.sha256(codes.tobytes()).hexdigest(),
                            trace_index=ti.astype(np.int32),
                            trace_diff_sha=hashlib.sha256(td.tobytes()).hexdigest(),
                            qerror_sum=np.float32(q),
                            params=np.array([24, 16, topol, neigh, 6000, 1, 21, 4000, 48, 8, 5]),
                            alpha=np.float32(0.05), radius=np.float32(8.0))

    # LVQ on ex1.dat
    tab = textio.LabelTable()
    e1, _ = textio.read_entries(d("ex1.dat"), tab)
    ci, _ = textio.read_entries(os.path.join(CLI, "lvq_init.cod"), tab)
    e2, _ = textio.read_entries(d("ex2.dat"), tab)
    for tag, kind, kw in (("lvq1", 1, {}), ("olvq1", 2, {}), ("lvq2", 3, {"winlen": 0.3}),
                          ("lvq3", 4, {"winlen": 0.3, "epsilon": 0.1}),
                          ("lvq1_invt", 1, {"alpha_type": 2})):
        codes, tal, ti, td = ref.lvq_train(kind, ci.points, ci.first_label, e1.points,
                                           e1.first_label, 5000, 0.05, **kw)
        wi, _, _ = ref.winners(codes, e2.points)
        acc = int((ci.first_label[wi[:, 0]] == e2.first_label).sum())
        extra = {}
        if tal is not None:
            extra["lra"] = np.array(tal)
        np.savez_compressed(os.path.join(TR, "lvq_ex1_%s.npz" % tag), codes=codes,
                            trace_index=ti.astype(np.int32), trace_diff=td,
                            correct_on_ex2=np.int64(acc), kind=np.int64(kind), **extra)
    # k-NN scans (find_winner_knn tie order etc.): duplicate rows force exact ties
    cb = np.concatenate([ci.points[:40], ci.points[:40]], axis=0)
    for knn in (1, 2, 5):
        wi, wd, _ = ref.winners(cb, e2.points[:300], knn=knn, use_knn_fn=True)
        np.savez_compressed(os.path.join(TR, "knn_ties_%d.npz" % knn), index=wi.astype(np.int32),
                            diff=wd)
    wi, wd, _ = ref.winners(cb, e2.points[:300], knn=1, use_knn_fn=False)
    np.savez_compressed(os.path.join(TR, "euc_ties.npz"), index=wi.astype(np.int32), diff=wd)

    # scalars: schedules, lattice distances, RNG, shuffle
    its = np.array([0, 1, 2, 17, 999, 4999, 5000, 123456, 9999999, 16777217, 99999999], dtype=np.int64)
    lens = np.array([1, 5000, 10000, 10000000, 100000000], dtype=np.int64)
    alphas = np.array([[ref.alpha(t, int(i), int(l), 0.05) if i <= l else 0.0 for i in its]
                       for t in (1, 2) for l in lens], dtype=np.float32)
    md = np.array([[ref.mapdist(tp, bx, by, tx, ty) for tp in (3, 4)]
                   for bx in range(0, 9, 2) for by in range(0, 7) for tx in range(0, 9, 3)
                   for ty in range(0, 7)], dtype=np.float32)
    np.savez_compressed(os.path.join(TR, "scalars.npz"), its=its, lens=lens, alphas=alphas,
                        mapdist=md, rand123=ref.rand_seq(123, 200), perm7=ref.shuffle_perm(3840, 7),
                        perm_small=ref.shuffle_perm(10, 3), randinit_ex=ref.randinit(ex.points, 12, 8, 123))
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
