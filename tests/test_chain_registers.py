"""CPU (cross-compile only): the persistent factorisation kernel (3pre_amd/csrc/pre3_cholp.hip) must keep its dependent chain free of scratch
traffic.  The chain's unrolled code takes every vector register a 768-thread workgroup may have; one more value living across it comes back
as a scratch reload in every pipeline step of the worker waves (seen during development: +30 % chain time).  The check reads the compiler's
own assembly: no scratch instruction between the first and the last matrix instruction that only the chain uses (round 6, the flag-driven
chain: v_mfma_f32_4x4x1 = the factor / z waves' lookahead, v_mfma_f32_16x16x4_f32 = the D workers, v_mfma_f32_32x32x2_f32 = the X workers), and
a bounded number of vector-register spills in the kernel as a whole (those sit around the out-of-line rescue stage's call, outside that region)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3pre_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_the_chain_of_k_cholp_has_no_scratch_traffic():
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-mllvm", "-pragma-unroll-threshold=200000"]      # the Makefile's
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for f in ("-O3", "-std=c++17", "-ffp-contract=on", "-pragma-unroll-threshold=200000"):
        assert f in mk, "the Makefile's flags changed: keep this test's in step (%s)" % f
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "cholp.s")
        r = subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-I", CSRC, "-Rpass-analysis=kernel-resource-usage", "-o", out,
                            os.path.join(CSRC, "pre3_cholp.hip")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(out).read().split("\n")
    start = [i for i, l in enumerate(asm) if l.startswith("_ZN4pre37k_cholp")]
    assert start, "k_cholp not found in the assembly"
    kern = asm[start[0]:]
    chain_ops = ("v_mfma_f32_32x32x2_f32", "v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_16x16x4_f32")
    for op in chain_ops:
        assert sum(op in l for l in kern) >= 40, "the chain's %s are gone?" % op
    mf = [i for i, l in enumerate(kern) if any(op in l for op in chain_ops)]
    assert len(mf) > 100, "the chain's f32 MFMAs are gone?"
    # basic blocks of the region: none that holds one of the chain's matrix instructions may touch scratch.  Between the roles (each wave's role is a
    # branch of its own, entered once per panel) the z wave parks the 16 accumulator registers it carries through the chain for the products and
    # reloads them at its role's exit: once per panel, in a block without arithmetic -- allowed, and bounded.
    blocks, cur = [], []
    for l in kern[mf[0]:mf[-1]]:
        if l.startswith(".LBB"):
            blocks.append(cur)
            cur = []
        cur.append(l.strip())
    blocks.append(cur)
    inside = [l for b in blocks if any(op in x for x in b for op in chain_ops) for l in b if "scratch_" in l]
    assert not inside, "scratch traffic inside the chain of k_cholp:\n" + "\n".join(inside[:8])
    between = [l for b in blocks for l in b if "scratch_" in l]
    assert len(between) <= 8, "scratch traffic between the chain's roles grew:\n" + "\n".join(between[:16])
    # (round 6) the strips' and the consumers' roles take the launch arguments by pointer: as a by-value struct they were 240 B of scratch stores per
    # thread at the call (39 MB of the LI launch's HBM writes, tools/pmc_ab_write.sh); and their calls must not carry the `tail` mark, or the roles
    # save ~75 callee-saved registers per thread at their entry
    for role in ("strip_body", "dd_body"):
        lab = [i for i, l in enumerate(asm) if re.match(r"^_ZN4pre3\d+%sE\w*:" % role, l)]
        assert lab, role
        assert "6CpArgs" not in asm[lab[0]], "%s takes CpArgs by value again: %s" % (role, asm[lab[0]])
        end = next(i for i in range(lab[0], len(asm)) if asm[i].startswith(".Lfunc_end"))
        n_st = sum("scratch_store" in l for l in asm[lab[0]:end])
        assert n_st <= 24, "%s stores %d registers to scratch (callee-saved convention back?)" % (role, n_st)
    call_copies = [l for l in kern[:mf[0]] if "scratch_store" in l and "Spill" not in l]
    assert len(call_copies) <= 32, "by-value argument copies in front of the roles' calls: %d stores" % len(call_copies)
    m = re.findall(r"Function Name: (\S+).*?VGPRs Spill: (\d+)", r.stderr, re.S)
    spills = {n: int(v) for n, v in m}
    k = [v for n, v in spills.items() if "k_cholp" in n]
    # (round 5: crit calls the out-of-line rescue stage, crit_tail, ONCE per launch between the LI update's last chain and the HI panel's; the
    #  values it keeps across that call are saved around it -- outside the chain, which the check above pins)
    # (round 6: + the 16 parked across the z wave's role, above; WRITE_SIZE of the launch measured unchanged, tools/pmc_ab_write.sh)
    assert k and k[0] <= 56, spills
