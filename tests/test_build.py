"""CPU-side checks of the product library: it builds for gfx950, exports every symbol
include/somhip.h declares, fails loudly without a GPU (no CPU fallback), and the exact
arithmetic chains contain no fused multiply-add."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    lib = os.path.join(ROOT, "som_lvq_pak_amd", "libsomhip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", ROOT, "lib"])
    return lib


def test_header_symbols_exported(built):
    from som_lvq_pak_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "somhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(somhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libsomhip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.somhip_version() == 1


def test_auto_batch_schedule_is_host_arithmetic(built):
    """somhip_som_auto_batch (the engine's own mini-batch boundaries, SOMHIP_BATCH_AUTO) is plain host arithmetic: it must
    answer without a GPU, cut a schedule into contiguous batches that end exactly at its length, follow its rule in
    (units, radius(t), alpha(t)) -- long batches while the rest of the run still forgets them, a short power of two after,
    batch 1 wherever the rule is not vouched for -- and refuse nonsense."""
    import ctypes as C
    from som_lvq_pak_amd import _lib
    lib = _lib.load()

    def ab(length, it, alpha=0.05, radius=128.0, units=65536, topol=3, neigh=1, alpha_type=1):
        a, b = C.c_int64(0), C.c_int64(0)
        p = _lib.SomParams(length, alpha, radius, alpha_type, 0, 0, -1, 0, 0, 0)
        rc = lib.somhip_som_auto_batch(C.byref(p), units, topol, neigh, it, C.byref(a), C.byref(b))
        return rc, a.value, b.value

    def cut(length, **kw):
        it, sizes = 0, []
        while it < length:
            rc, st, ln = ab(length, it, **kw)
            assert rc == 0 and st == it and 0 < ln <= 32768 and st + ln <= length
            assert ab(length, st + ln - 1, **kw)[1:] == (st, ln)    # every iteration of the batch names the same batch
            sizes.append(ln)
            it = st + ln
        assert it == length
        return sizes

    # configs[3]: 256 x 256 hexa bubble, alpha 0.05 linear, radius 128 -> 1, 10 M iterations
    sizes = cut(10_000_000)
    nlong = sizes.index(8192)
    assert sizes[:nlong] == [32768] * nlong and set(sizes[nlong:-1]) == {8192} and sizes[-1] <= 8192
    assert nlong == 259                                            # the long batches end where F = 64 is left of the run (84.9 %)
    # the rule follows its inputs: a larger rate or a gaussian neighbourhood forgets faster (long batches run on longer),
    # more units per sample's neighbourhood move less per batch (longer tail batches)
    assert cut(10_000_000, alpha=0.2).count(32768) > nlong
    assert cut(10_000_000, neigh=2).count(32768) > nlong
    assert cut(10_000_000, radius=32.0).count(32768) < nlong       # a small radius teaches less of the map per sample: forgets slower
    big = cut(40_000_000, units=262144, radius=256.0)
    assert big[0] == 32768 and big[-2] == 8192
    assert cut(10_000_000, alpha_type=2)[0] in (1, 32768)           # inverse_t: still a valid cut
    # not vouched for -> the reference's own schedule: small maps (configs[1]), short runs, runs too short for 64 long batches
    for length, kw in ((100000, dict(radius=10.0, units=1024)), (1, {}), (4095, {}), (32 * 32768, {}), (10_000_000, dict(units=16383)),
                       (10_000_000, dict(alpha=0.0)), (70 * 32768, {})):
        for it in (0, length // 2, length - 1):
            assert ab(length, it, **kw) == (0, it, 1), (length, kw)
    assert ab(1000, 1000)[0] != 0 and ab(0, 0)[0] != 0 and ab(1000, -1)[0] != 0 and ab(1000, 5, units=0)[0] != 0


def test_no_cpu_fallback(built):
    """without a GPU the engine must refuse, not silently compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from som_lvq_pak_amd import engine as E
    with pytest.raises(Exception, match="no HIP device|no CPU path|hip"):
        E.Engine(0)


def test_product_never_touches_oracle():
    """nothing under som_lvq_pak_amd/ (the product) may import or link oracle/"""
    pkg = os.path.join(ROOT, "som_lvq_pak_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hpp", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"(^|\s)(from|import)\s+oracle\b", txt), f
                assert "liboracle" not in txt and "oracle/_ref" not in txt and "ref_harness" not in txt, f


def test_exact_kernels_have_no_fma(built, tmp_path):
    """distance = fp32 sub, mul, add; update = sub, mul, add -- three roundings each
    (reference lvq_pak.c:70-71, 348-349).  A contracted v_fma/v_fmac in those kernels would
    change results, so look at the gfx950 ISA.  (Kernels that divide or call exp legitimately
    contain fma inside the division / exp expansions and are checked for parity on the GPU.)"""
    s = os.path.join(str(tmp_path), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "--cuda-device-only", "-S", "-o", s,
                           os.path.join(ROOT, "som_lvq_pak_amd", "csrc", "somhip.hip")])
    txt = open(s).read()
    bodies = dict(re.findall(r"^(_ZN6somhip\w+):.*?\n(.*?)s_endpgm", txt, flags=re.S | re.M))
    checked = 0
    for name, body in bodies.items():
        exact = ("k_scan_exact" in name or "k_scan_masked" in name
                 or re.search(r"k_som_update_runILi\d+ELi\d+ELb0", name) or "k_som_online_stepILb0" in name)
        if not exact:
            continue
        checked += 1
        bad = re.findall(r"\bv_(?:pk_)?(?:fma|fmac|mac|mad)_f32\b.*", body)
        assert not bad, (name, bad[:3])
        assert re.search(r"v_(pk_)?mul_f32", body) and re.search(r"v_(pk_)?add_f32", body)
    assert checked >= 6


def test_ring_kernel_keeps_its_asm_loads_in_place(built, tmp_path):
    """k_dist_mfma_bf16_l1r (kernels/prefilter_l1_ring.hpp) reads its MFMA fragments by ds_read_b128 in inline assembly and
    orders them with counted s_waitcnt lgkmcnt of its own: the compiler does not know those registers are in flight.  The
    kernel is only sound while the compiler neither spills nor copies such a register (a v_mov / scratch store issued
    before the data has arrived would move stale bytes): no scratch, no spill, 2 waves per SIMD, and no instruction other
    than an MFMA takes a ds_read destination as a source.  Its stage waits are the counted ones (vmcnt(4) per stage,
    lgkmcnt(4) + 4 x lgkmcnt(11)), and no stage wait drains (the drains belong to the epilogue and the exit)."""
    s = os.path.join(str(tmp_path), "k.s")
    res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", s, "-Rpass-analysis=kernel-resource-usage",
                          os.path.join(ROOT, "som_lvq_pak_amd", "csrc", "somhip.hip")], stderr=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    m = re.search(r"Function Name: _ZN6somhip20k_dist_mfma_bf16_l1r.*?LDS Size", res.stderr, flags=re.S)
    assert m, "no resource remark for the ring kernel"
    rem = m.group(0)
    assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", rem) and re.search(r"VGPRs Spill: 0\b", rem) and re.search(r"SGPRs Spill: 0\b", rem), rem
    assert re.search(r"Occupancy \[waves/SIMD\]: 2\b", rem), rem
    txt = open(s).read()
    body = re.search(r"^_ZN6somhip20k_dist_mfma_bf16_l1r\w+:.*?\n(.*?)\.Lfunc_end", txt, flags=re.S | re.M).group(1)
    lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]

    def vregs(text):
        out = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
            out.update(range(int(a), int(b) + 1))
        out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
        return out

    dst = set()
    for ln in lines:
        if ln.startswith("ds_read_b128"):
            dst |= vregs(ln.split(",")[0])
    assert len(dst) >= 64
    nmfma = 0
    for ln in lines:
        op = ln.split()[0]
        if op.startswith("v_mfma"):
            nmfma += 1
            continue
        if op.startswith(("ds_read", "s_")) or "," not in ln:
            continue
        if op.startswith(("v_mov", "v_accvgpr", "scratch_", "buffer_store", "v_swap", "v_permlane")):
            assert not (vregs(ln.split(",", 1)[1]) & dst), "a register loaded by ds_read is copied: " + ln
    assert nmfma % 32 == 0 and nmfma >= 64
    assert body.count("s_waitcnt vmcnt(4)") >= 2 and body.count("s_waitcnt lgkmcnt(4)") >= 2 and body.count("s_waitcnt lgkmcnt(11)") >= 8
    # between a stage's barrier and its last MFMA no drain: every vmcnt(0) / lgkmcnt(0) stands outside the MFMA blocks
    for blk in re.findall(r"s_waitcnt vmcnt\(4\).*?s_barrier(.*?)(?=s_waitcnt vmcnt\(4\)|s_waitcnt vmcnt\(0\))", body, flags=re.S):
        if "v_mfma" in blk:
            head = blk[:blk.rindex("v_mfma")]
            assert "lgkmcnt(0)" not in head and "vmcnt(0)" not in head


def test_scalar_update_kernel_reads_nothing_before_its_wait(built, tmp_path):
    """k_som_update_bubble_s / k_som_update_gauss_s issue their scalar loads from inline assembly, so the
    compiler's own s_waitcnt insertion does not cover them: the kernels rely on nothing touching a load's
    destination SGPRs between the s_load and the next `s_waitcnt lgkmcnt(0)`.  Check that on the ISA inside
    every basic block (a whole-CFG dataflow reports infeasible paths through the shared break-or-continue
    block of the unrolled loop), that every x load follows a wait in its own block, and that the arithmetic
    is packed sub/mul/add.  The functional proof is the bit-exact parity suite on the GPU."""
    s = os.path.join(str(tmp_path), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                           "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", s,
                           os.path.join(ROOT, "som_lvq_pak_amd", "csrc", "somhip.hip")])
    txt = open(s).read()
    bodies = dict(re.findall(r"^(_ZN6somhip2[01]k_som_update_(?:bubble|gauss)_[sh]\w+):.*?\n(.*?)\.Lfunc_end", txt, flags=re.S | re.M))
    assert len(bodies) >= 3 and any("gauss_s" in n for n in bodies) and any("gauss_h" in n for n in bodies) and any("bubble" in n for n in bodies)

    def sregs(text):
        out = set()
        for a, b, c in re.findall(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", text):
            out.update(range(int(a), int(b) + 1) if a else [int(c)])
        return out

    def check(name, body, across_blocks):
        """Forward dataflow over the kernel's basic blocks: pending = SGPRs with a scalar load in flight.
        across_blocks=False keeps the check inside each basic block (the gaussian kernel's shared
        break-or-continue block makes infeasible paths look like hazards; its parity tests run on the GPU)."""
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(".") or re.match(r"\.LBB\w+:", ln or "")]
        blocks, label_of, cur = [[]], {}, 0
        for ins in lines:
            m = re.match(r"(\.LBB\w+):", ins)
            if m:
                if blocks[-1]:
                    blocks.append([])
                label_of[m.group(1)] = len(blocks) - 1
                continue
            blocks[-1].append(ins)
            if ins.startswith(("s_branch", "s_cbranch", "s_endpgm")):
                blocks.append([])
        succ = []
        for i, b in enumerate(blocks):
            last = b[-1] if b else ""
            tgt = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", last)
            out = []
            if tgt:
                out.append(label_of[tgt.group(1)])
            if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(blocks):
                out.append(i + 1)
            succ.append(out)
        pend_in = [set() for _ in blocks]
        loads = 0

        def run(i, verify):
            nonlocal loads
            pending = set(pend_in[i])
            for ins in blocks[i]:
                if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                    pending.clear()
                    continue
                m = re.match(r"s_load_dword(?:x\d+)?\s+(s\[\d+:\d+\]|s\d+)\s*,(.*)", ins)
                if m:
                    if verify:
                        assert not (sregs(m.group(2)) & pending), (name, ins)
                        loads += 1
                    pending |= sregs(m.group(1))
                    continue
                if verify:
                    assert not (sregs(ins) & pending), (name, ins, sorted(pending))
            return pending

        work = list(range(len(blocks)))
        while work:
            i = work.pop()
            out = run(i, False)
            for j in succ[i] if across_blocks else []:
                if not out <= pend_in[j]:
                    pend_in[j] |= out
                    work.append(j)
        for i in range(len(blocks)):
            run(i, True)
        assert loads >= 8, name

    for name, body in bodies.items():
        if "bubble" in name:                               # (the gaussian rate's fp64 exp/sqrt expansions legitimately use fma)
            assert not re.findall(r"\bv_(?:pk_)?(?:fma|fmac|mac|mad)_f32\b.*", body), name
        assert re.search(r"v_(pk_)?mul_f32", body) and re.search(r"v_(pk_)?add_f32", body), name
        # no scalar-register spills at all: a spill may copy the destination of a load that is still in flight (a
        # 32-dims-per-wave gaussian variant did exactly that) and the per-block scan below would miss one in another block
        # (k_som_update_gauss_h: 32 dims per wave at 8 waves per SIMD; it hands sample indices and per-entry scalars across
        # with v_readlane and does keep loop-invariant pointers in a VGPR's lanes -- the scan below sees a v_writelane of a
        # register in flight like any other read; its entry loop itself must be free of spills of either kind)
        if "gauss_h" in name:
            loops = [blk for blk in re.split(r"\n\.LBB\w+:", body) if blk.count("s_load_dwordx16") >= 4]
            assert loops and all("v_writelane_b32" not in blk and "scratch_" not in blk.split("s_load_dwordx16", 1)[1].rsplit("s_load_dwordx16", 1)[0]
                                 for blk in loops), name
        else:
            assert "v_writelane_b32" not in body and "v_readlane_b32" not in body, name
        check(name, body, False)
        # phase structure: every x load (s_load_dwordx16 from the inline assembly) is issued after a wait in its own block
        for blk in re.split(r"\n\.LBB\w+:", body) if "gauss_h" not in name else []:   # (K4h opens a tile with a request)
            if "ASMSTART\n\ts_load_dwordx16" in blk:
                assert blk.index("s_waitcnt lgkmcnt(0)") < blk.index("ASMSTART\n\ts_load_dwordx16"), name


def test_gaussian_rate_short_form_gives_the_library_chains_float(tmp_path):
    """kernels/gauss_rate.hpp (the rate of gaussian_adapt, som_rout.c:539-542, without fp64 library calls), compiled for
    the host, against sqrt / division / exp as the reference writes them with glibc: every argument the short form
    decides gives the same float, its quotient is the correctly rounded one, and it decides nearly all of them
    (lattice distances of maps up to 350 x 400, radii over the whole float range, the schedule's 1 ... 128 densely)"""
    exe = str(tmp_path / "gauss_rate_check")
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "som_lvq_pak_amd", "csrc", "kernels"), "-x", "c",
           os.path.join(ROOT, "tests", "helpers", "gauss_rate_check.c"), "-o", exe, "-lm"]
    if " fma " in open("/proc/cpuinfo").read():
        cmd.insert(1, "-mfma")
    subprocess.check_call(cmd)
    for seed in (1, 2):
        p = subprocess.run([exe, "8000000", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        m = re.match(r"gauss_rate_check: (\d+) cases, (\d+) decided .* (\d+) wrong floats, (\d+) wrong quotients", p.stdout)
        assert m and int(m.group(3)) == 0 and int(m.group(4)) == 0, p.stdout
        assert int(m.group(2)) > 0.98 * int(m.group(1)), p.stdout
